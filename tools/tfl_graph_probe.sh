#!/bin/bash
# Graph replay under rocprofv3 --kernel-trace (profiles/r02_graph_replay_under_rocprofv3.md): every capture
# this library makes must replay under the profiler - forked captures as they are, the others through the one-node second
# branch they are captured with. Each run is its own process with its own timeout.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/tflgraph
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, then the program and its arguments
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/p_$name -o run --output-format csv -- "$@" > $OUT/$name.log 2>&1
  echo "exit=$?" >> $OUT/$name.log
  echo "== $name: $(grep -E 'graph replays|invoke 224|classify 640|ms/step|exit=' $OUT/$name.log | tr '\n' ' ')"
  rm -rf $OUT/p_$name
}
run H_engine_invoke_batch2 python3 $R/tools/graph_probe.py invoke 300
run I_engine_evaluate_batch2 python3 $R/tools/graph_probe.py evaluate 300
run I2_engine_evaluate_batch1_tail_on_main_stream python3 $R/tools/time_steps.py 1 tailfork=0
run J_time_steps_batch1_and_4 python3 $R/tools/time_steps.py 1 4
run K_tflite_graph1 python3 $R/tools/time_tflite.py --graph 1 --invokes 100
echo done
