#!/bin/bash
# rocprofv3 kernel trace of graph-replayed steps at one batch size: the timeline of the LAST step (start offset, duration,
# idle time in front of each kernel, how many kernels run at that moment). Usage (inside gpurun): bash tools/trace_step.sh [batch]
B=${1:-1}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/trs
rocprofv3 --kernel-trace -d $R/gpurun_out/trs -o run --output-format csv -- python3 $R/tools/time_steps.py $B > $R/gpurun_out/trs.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, re
f = glob.glob("$R/gpurun_out/trs/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "stem_pool" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]          # the last complete step
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
def short(n):
    m = re.search(r"conv_igemm_f16<([^>]*)>", n)
    if m:
        x = [s.strip() for s in m.group(1).split(",")]
        return "conv<" + ",".join(x[:4]) + ",st" + x[5] + (",sk" if x[7] == "true" else "") + (",ml" if x[9] == "true" else "") + (",dual" if len(x) > 12 and x[12] == "true" else "") + ">"
    m = re.search(r"(?:yh::|_ZN2yh\d+)([a-z_0-9]+)", n)
    return m.group(1) if m else n[:30]
end_prev = t0
print(f"batch $B: {len(step)} kernels, step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
busy_union, cur_end = 0, t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    idle = max(0, s - cur_end)
    conc = sum(1 for q in step if int(q["Start_Timestamp"]) <= s < int(q["End_Timestamp"]))
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  idle {idle / 1e3:5.1f}  x{conc}  {short(r['Kernel_Name'])}")
    if e > cur_end:
        busy_union += e - max(s, cur_end); cur_end = e
print(f"device busy (union of kernel intervals) {busy_union / 1e3:.1f} us of {(cur_end - t0) / 1e3:.1f} us")
PY
