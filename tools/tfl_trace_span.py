"""From a rocprofv3 --kernel-trace CSV of tools/time_tflite_fuse.py: per invoke, the device span (first kernel start -> last kernel end),
the sum of kernel durations and the idle time between kernels - is the invoke bound by the host's launch rate or by the kernels?"""
import csv, sys
import numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n_per = int(sys.argv[2])
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "tfl_" in r["Kernel_Name"]]
# invokes = runs of n_per consecutive tfl kernels starting at the first conv of the plan
spans, sums, gaps = [], [], []
for i in range(0, len(k) - n_per + 1, n_per):
    blk = k[i:i + n_per]
    spans.append((blk[-1][1] - blk[0][0]) / 1e3)
    sums.append(sum(e - s for s, e, _ in blk) / 1e3)
    gaps.append(sum(max(0, blk[j + 1][0] - blk[j][1]) for j in range(n_per - 1)) / 1e3)
print(f"{len(spans)} blocks of {n_per} kernels: device span median {np.median(spans):.1f} us, sum of kernel durations {np.median(sums):.1f} us, idle between kernels {np.median(gaps):.1f} us")
