"""Generates tests/golden/reference_kat.json: known-answer vectors for the reference's own integer
and float logic in /root/reference/src/yolact.rs and src/scene.rs.

The reference has no tests, no fixtures and cannot be built here (Rust toolchain absent, SURVEY.md
§8c), so these vectors come from THIS script: a literal pure-Python emulation of the Rust
expressions (iterator chains, `as` casts, strict `>` on f32, `&`/`<<` precedence, wrapping usize),
written independently of oracle/orc_ref.c. It reproduces SURVEY.md Appendix A.1-A.6 verbatim and
adds seeded random cases. Run:  python tests/golden/make_reference_kat.py
"""
import json
import math
import os
import random
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
U32 = 0xFFFFFFFF
USIZE = (1 << 64) - 1


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def gated_argmax_chunk(chunk):
    """yolact.rs:108-118: `let mut max = 0.0; chunk.iter().take(4).map(|a| {*a > max && {max = *a; true}})`
    then the four-arm `match`."""
    mx = 0.0
    cls = []
    for a in chunk[:4]:
        hit = (a > mx)  # NaN compares false
        if hit:
            mx = a
        cls.append(hit)
    if cls == [False, True, False, False]:
        return 1
    if cls[0] is False and cls[2] is True and cls[3] is False:
        return 2
    if cls[0] is False and cls[3] is True:
        return 3
    return 0


def pack(cls, id_i8):
    """yolact.rs:127: `(*cls as u32) << 24 & (id as u32) << 16` — `<<` binds tighter than `&`;
    `id: i8 as u32` sign-extends."""
    idu = id_i8 & U32
    return ((cls << 24) & U32) & ((idu << 16) & U32)


def terrible_id(img, grid_w=28, budget=200000):
    """yolact.rs:52-88 with release-build wrapping usize arithmetic. Returns (diverged, ids)."""
    n = len(img)
    out = [-1] * n
    idv = -1
    pops = 0
    for px0, c in enumerate(img):
        if c == 3 and out[px0] == -1:
            idv += 1
            if idv > 127:
                idv -= 256
            st = [px0]
            while st:
                px = st.pop()
                pops += 1
                if pops > budget:
                    return True, None
                for q in ((px - 1) & USIZE, (px + 1) & USIZE, (px - grid_w) & USIZE, (px + grid_w) & USIZE):
                    if q < n and img[q] == 3:  # img.get(q) == Some(&3)
                        out[q] = idv
                        st.append(q)
    return False, out


def main():
    rnd = random.Random(20261004)
    kat = {}
    # A.1 gated argmax: the survey's table + random
    a1 = [[0.5, 9, 9, 9], [-1, 2, 2, 1], [-1, 1, 2, 3], [-1, 3, 2, 1], [-1, 1, 3, 2], [-1, -1, -1, -1],
          [0, 0, 0, 0], [-1, "nan", 1, 0.5]]
    vals = [-2.0, -0.0, 0.0, 0.25, 1.0, 1.0, 3.5, "nan", "inf", "-inf"]
    for _ in range(400):
        a1.append([rnd.choice(vals) for _ in range(4)])
    for _ in range(400):
        a1.append([f32(rnd.uniform(-3, 3)) for _ in range(4)])

    def tof(v):
        return float(v) if isinstance(v, str) else v
    kat["gated_argmax"] = [{"in": c, "cls": gated_argmax_chunk([tof(v) for v in c])} for c in a1]
    # A.2 packing
    kat["pack"] = [{"cls": c, "id": i, "u32": pack(c, i)} for c in range(4) for i in (-1, 0, 1, 5, 127, -128)]
    # A.4 pixel packing: u32 = r<<24|g<<16|b<<8 (scene.rs:86); unpack = to_be_bytes()[..3]
    px = [(18, 52, 86)] + [(rnd.randrange(256), rnd.randrange(256), rnd.randrange(256)) for _ in range(32)]
    kat["pixel"] = [{"rgb": list(p), "u32": (p[0] << 24) | (p[1] << 16) | (p[2] << 8)} for p in px]
    # A.5 dequantisation: scale * (((x as i32) - zero_point) as f32), f32 arithmetic
    dq = [(130, 128, 0.5), (0, 128, 0.0625)] + [(rnd.randrange(256), rnd.randrange(256), f32(rnd.uniform(0.001, 0.2))) for _ in range(64)]
    kat["dequant"] = [{"x": x, "zp": zp, "scale": s, "out": f32(f32(s) * f32(float(x - zp)))} for x, zp, s in dq]
    # A.6 flood fill: terminating and diverging grids (28x28, values 0..3)
    grids = []
    g = [0] * 784
    grids.append(("no_balls", list(g)))
    g2 = list(g); g2[5 * 28 + 7] = 3
    grids.append(("lone_ball", g2))
    g3 = list(g); g3[100] = 3; g3[101] = 3
    grids.append(("adjacent_pair", g3))
    g4 = list(g); g4[27] = 3; g4[28] = 3  # row-wrap neighbours in linear index space
    grids.append(("row_wrap_pair", g4))
    g5 = list(g); g5[0] = 3; g5[783] = 3; g5[400] = 3
    grids.append(("corners_isolated", g5))
    g6 = list(g)
    for k in range(0, 784, 3):
        g6[k] = 3  # every third cell: neighbours +-1 are not balls, +-28: 28 % 3 = 1 -> not balls
    grids.append(("sparse_many", g6))
    for t in range(12):
        gr = [rnd.choice([0, 0, 0, 1, 2]) for _ in range(784)]
        for _ in range(rnd.randrange(0, 6)):
            gr[rnd.randrange(784)] = 3
        grids.append((f"random_{t}", gr))
    ff = []
    for name, gr in grids:
        div, ids = terrible_id(gr)
        ff.append({"name": name, "classes": gr, "diverges": div, "ids": ids})
    kat["flood_fill"] = ff
    # A.10 consumer: low 16 bits (scene.rs:93)
    kat["consumer_low16"] = [{"u32": v, "u16": ((v << 16) & U32) >> 16} for v in (0, 0x01000000, 0x03000000, 0x12345678, U32)]
    with open(os.path.join(HERE, "reference_kat.json"), "w") as f:
        json.dump(kat, f, allow_nan=True)
    print("wrote reference_kat.json:", {k: len(v) for k, v in kat.items()})


if __name__ == "__main__":
    main()
