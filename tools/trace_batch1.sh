# rocprofv3 kernel trace of graph-replayed batch-1 steps: kernel durations vs gaps between kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tr1 -o run --output-format csv -- python3 $R/tools/time_steps.py 1 > $R/gpurun_out/tr1.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/tr1/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-92*20:]   # the last ~20 steps
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gaps_s = sorted(gaps)
print("kernels", len(rows), "busy us", busy/1e3, "span us", span/1e3, "busy frac", busy/span)
print("gap median ns", gaps_s[len(gaps)//2], "p90", gaps_s[int(len(gaps)*0.9)], "mean", sum(gaps)/len(gaps))
import collections
d = collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"][:50]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])): print(f"{k:52s} n={len(v):5d} avg {sum(v)/len(v)/1e3:7.2f} us total {sum(v)/1e3:9.1f}")
PY
