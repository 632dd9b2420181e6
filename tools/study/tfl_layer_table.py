"""From a rocprofv3 --kernel-trace CSV of tfl_layer_run.py: one line per launch of the plan - kernel, workgroups, median duration and
median idle time in front of it over the invokes - and the totals.   usage: tfl_layer_table.py TRACE.csv LAUNCHES_PER_INVOKE"""
import csv, sys
import numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
k = [r for r in rows if "tfl_" in r["Kernel_Name"]]
nblk = len(k) // n
k = k[len(k) - nblk * n:]   # (whole invokes, counted from the end)
dur = np.array([[int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in k[b * n:(b + 1) * n]] for b in range(nblk)]) / 1e3
st = np.array([[int(r["Start_Timestamp"]) for r in k[b * n:(b + 1) * n]] for b in range(nblk)]) / 1e3
en = st + dur
gap = np.zeros_like(dur)
gap[:, 1:] = st[:, 1:] - en[:, :-1]
d, g = np.median(dur[20:], 0), np.median(gap[20:], 0)
def short(s):
    s = s.replace("(anonymous namespace)::", "").replace("void ", "")
    return s.split("(")[0]
for i in range(n):
    r = k[i]
    wg = [int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ"]
    print(f"{i:3d} {short(r['Kernel_Name']):28s} wg {wg[0]:5d} x {wg[1]:3d} x {wg[2]:2d}  {d[i]:7.2f} us  idle before {g[i]:5.2f}")
print(f"{nblk - 20} invokes: kernels {d.sum():.1f} us, idle {g.sum():.1f} us, span {np.median(en[20:, -1] - st[20:, 0]):.1f} us")
by = {}
for i in range(n):
    a = by.setdefault(short(k[i]["Kernel_Name"]), [0, 0.0]); a[0] += 1; a[1] += d[i]
for name, (c, t) in sorted(by.items(), key=lambda x: -x[1][1]):
    print(f"  {name:28s} {c:3d} launches {t:7.1f} us  ({t / c:.2f} each)")
