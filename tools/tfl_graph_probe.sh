#!/bin/bash
# VERDICT r1 item 5 - what makes rocprofv3 --kernel-trace crash on a replayed graph, and does a second captured branch
# avoid it? Each run is its own process with its own timeout; logs in gpurun_out/tflgraph/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/tflgraph
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, then the program and its arguments
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/p_$name -o run --output-format csv -- "$@" > $OUT/$name.log 2>&1
  echo "exit=$?" >> $OUT/$name.log
  echo "== $name: $(grep -E 'graph replays|invoke 224|classify 640|exit=' $OUT/$name.log | tr '\n' ' ')"
  if [ "$name" = "E_tflite_graph_two_branches" ]; then cp "$(find $OUT/p_$name -name '*kernel_stats.csv' | sort | tail -1)" $OUT/r02_tflite_rocprofv3_kernel_stats.csv 2>/dev/null; fi
  rm -rf $OUT/p_$name
}
run E_tflite_graph_two_branches python3 $R/tools/time_tflite.py --graph 2 --invokes 300
run F_engine_invoke_two_branches python3 $R/tools/graph_probe.py invoke 300
run G_tflite_graph_single_branch python3 $R/tools/time_tflite.py --graph 1 --invokes 50
echo "== no profiler:"
python3 $R/tools/time_tflite.py --graph 2 --invokes 300 | tail -2
python3 $R/tools/time_tflite.py --graph 1 --invokes 300 | tail -2
python3 $R/tools/time_tflite.py --graph 0 --invokes 300 | tail -2
echo done
