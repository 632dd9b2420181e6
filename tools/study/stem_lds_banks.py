"""LDS bank-conflict count of stem_pool_f16's access patterns, by the lane-group rules of
/opt/skills/guides/MI355X_MICROARCH.md (LDS table). Prints LDS-array cycles per wave-instruction (conflict-free: 4 for b128)."""
import itertools, random, sys
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 += [[l + 32 for l in g] for g in G128[:2]]
def cyc_b128(addr):   # addr[lane] byte address or None
    tot = 0
    for g in G128:
        banks = {}
        for l in g:
            if addr[l] is None: continue
            for d in range(4):
                banks.setdefault(((addr[l] >> 2) + d) % 64, set()).add((addr[l] >> 2) + d)
        tot += max([len(v) for v in banks.values()] or [1])
    return tot
def cyc_groups(addr, width, nbanks, groups):
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            if addr[l] is None: continue
            for d in range(max(1, width // 4)):
                banks.setdefault(((addr[l] >> 2) + d) % nbanks, set()).add((addr[l] >> 2) + d)
        tot += max([len(v) for v in banks.values()] or [1])
    return tot
ST = 17
def stem(PC, SS, quiet=False):
    NPX = ST * ST
    # B fragment reads: lane (l15, lh): patch + ((2 mi) PC + 2 mj + 2 lh) 8 + r PC 8
    tot = n = 0
    for mt in range((NPX + 15) // 16):
        for r in range(7):
            addr = []
            for lane in range(64):
                l15, lh = lane & 15, lane >> 4
                m = mt * 16 + l15
                mi, mj = (m // ST, m % ST) if m < NPX else (ST - 1, ST - 1)
                addr.append(((2 * mi) * PC + 2 * mj + 2 * lh) * 8 + r * PC * 8)
            tot += cyc_b128(addr); n += 1
    bread = tot / n
    # stage writes (ds_write_b64: 4 x 16 contiguous lanes, 32 banks)
    tot = n = 0
    for mt in range((NPX + 15) // 16):
        for ct in range(4):
            addr = []
            for lane in range(64):
                l15, lh = lane & 15, lane >> 4
                m = mt * 16 + l15
                addr.append(m * SS + (ct * 16 + 4 * lh) * 2 if m < NPX else None)
            tot += cyc_groups(addr, 8, 32, [list(range(16 * k, 16 * k + 16)) for k in range(4)]); n += 1
    swrite = tot / n
    # pool reads
    tot = n = 0
    for w in range(8):
        for d in range(9):
            addr = []
            for lane in range(64):
                wq = w * 64 + lane
                pp, cg = wq >> 3, wq & 7
                ly, lx = pp // 8, pp % 8
                addr.append(((2 * ly + d // 3) * ST + 2 * lx + d % 3) * SS + cg * 16)
            tot += cyc_b128(addr); n += 1
    pread = tot / n
    if not quiet: print(f"PC={PC} SS={SS}: B read {bread:.2f} cyc (free 4), stage write {swrite:.2f} (free 4), pool read {pread:.2f} (free 4)")
    per_tile = 8 * 5 * 7 * bread / 1 * (19 / 20) + 19 * 4 * swrite + 72 * pread
    return bread, swrite, pread
stem(40, 144)
for PC in (40, 42, 44, 46, 48, 50, 52, 56):
    for SS in (144, 136, 160, 192, 208, 272):
        stem(PC, SS)
# LUT: 16-bit reads at random entries
random.seed(1)
tot = 0
for _ in range(2000):
    addr = [2 * (256 * random.randrange(3) * 0 + random.randrange(256)) for _ in range(64)]
    tot += cyc_groups(addr, 2, 32, [list(range(32)), list(range(32, 64))])
print("LUT read (random bytes, one channel):", tot / 2000, "cyc (free 2)")

print("---- row/column groups: group g < 17 = stem row g, columns 0..15; groups 17, 18 = column 16 (rows 0..15; row 16 + padding)")
def pix(mt, l15):
    if mt < 17: return mt, l15
    if mt == 17: return l15, 16
    return (16, 16) if l15 == 0 else None
def stem2(PC, SS, poolmap):
    tot = n = 0
    for mt in range(19):
        for r in range(7):
            addr = []
            for lane in range(64):
                l15, lh = lane & 15, lane >> 4
                q = pix(mt, l15) or (16, 16)
                addr.append(((2 * q[0]) * PC + 2 * q[1] + 2 * lh) * 8 + r * PC * 8)
            tot += cyc_b128(addr); n += 1
    bread = tot / n
    tot = n = 0
    for mt in range(19):
        for ct in range(4):
            addr = []
            for lane in range(64):
                l15, lh = lane & 15, lane >> 4
                q = pix(mt, l15)
                addr.append((q[0] * ST + q[1]) * SS + (ct * 16 + 4 * lh) * 2 if q else None)
            tot += cyc_groups(addr, 8, 32, [list(range(16 * k, 16 * k + 16)) for k in range(4)]); n += 1
    swrite = tot / n
    tot = n = 0
    for w in range(8):
        for d in range(9):
            addr = []
            for lane in range(64):
                pp, cg = poolmap(w * 64 + lane)
                ly, lx = pp // 8, pp % 8
                addr.append(((2 * ly + d // 3) * ST + 2 * lx + d % 3) * SS + cg * 16)
            tot += cyc_b128(addr); n += 1
    return bread, swrite, tot / n
maps = {"px=wq>>3,cg=wq&7": lambda wq: (wq >> 3, wq & 7), "px=wq&63,cg=wq>>6": lambda wq: (wq & 63, wq >> 6),
        "px=(wq>>1)&63.., cg=2(wq>>7)+(wq&1)": lambda wq: ((wq >> 1) & 63, 2 * (wq >> 7) + (wq & 1)),
        "px=(wq>>2)&63, cg=4(wq>>8)+(wq&3)": lambda wq: ((wq >> 2) & 63, 4 * (wq >> 8) + (wq & 3))}
best = []
for PC in (40, 42, 44, 46):
    for SS in range(128, 300, 8):
        for name, f in maps.items():
            b, s, pr = stem2(PC, SS, f)
            cost = 8 * 5 * 7 * b / 5 * 19 / 8 / 1 + 0   # placeholder
            lds = 19 * 7 * 2 * b + 19 * 4 * s + 72 * pr     # per tile: B reads (two channel halves), stage writes, pool reads
            best.append((lds, PC, SS, name, b, s, pr))
best.sort()
for x in best[:12]: print("LDS cycles/tile %.0f  PC=%d SS=%d pool map %s: B read %.2f, stage write %.2f, pool read %.2f" % x)
b, s, pr = stem(40, 144, True); print("now: %.0f" % (19 * 7 * 2 * b + 19 * 4 * s + 72 * pr))
