"""Builds profiles/<round>_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/pmc_run.py: per kernel symbol, HBM-side bytes per launch, corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: counters are in KiB (x1024), and on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced read stream (x2)."""
import csv, json, re, sys, collections

def symbol(name):
    m = re.search(r"conv_igemm_f16<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        a = [{"false": "0", "true": "1"}.get(x, x) for x in a]
        s = ",".join(a[:6])
        if len(a) >= 9 and a[8] == "16":
            s += ",mfma16"
        ml = "[ml]" if len(a) >= 10 and a[9] == "1" else ""
        tail = "[+1x1]" if len(a) >= 14 and a[13] == "1" else ""   # fused 1x1 tail (conv_igemm.hip: TAIL)
        if len(a) >= 15 and a[14] == "1":
            tail += "[3x3]"                                         # the streaming tile's 3x3 specialisation (K3)
        if len(a) >= 11 and a[10] == "1":                          # fp8 operands: the engine's label for this instantiation
            return f"conv_igemm_fp8<{','.join(a[:4])}>{ml}{tail}"
        return f"conv_igemm_f16<{s}>{ml}{tail}"
    m = re.search(r"bneck_chain_f16<([^>]*)>", name)   # <PL, TM, wave grids x4, NEXT, KT2>: the engine's label (bneck.hip: bneck_symbol)
    if m:
        a = [{"false": "0", "true": "1"}.get(x.strip(), x.strip()) for x in m.group(1).split(",")]
        return "bneck_chain_f16<%s,%s%s%s>" % (a[0], a[1], ",next" if len(a) > 6 and a[6] == "1" else "", ",dual" if len(a) > 7 and a[7] != "0" else "")
    m = re.search(r"(?:yh::|_ZN2yh\d+)([a-z_0-9]+)", name)
    return m.group(1) if m else name[:40]

def load(path, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                k = symbol(r["Kernel_Name"])
                acc[k] += float(r["Counter_Value"]); n[k] += 1
    return acc, n

fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    launches = max(nf[k], nw[k])
    if not launches or not k.startswith(("conv_igemm", "bneck_chain", "bneck_xn", "conv_rowpatch", "stem_pool", "splitk", "det_", "bilinear", "maxpool", "preprocess")):
        continue
    fb = fetch[k] * 1024 * 2 / max(nf[k], 1)
    wb = write[k] * 1024 / max(nw[k], 1)
    out[k] = dict(launches=launches, fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb), hbm_bytes_per_launch=round(fb + wb))
json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_run.py %s; FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes" % sys.argv[3],
               batch=int(sys.argv[3]), kernels=out), open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
