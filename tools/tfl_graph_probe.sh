#!/bin/bash
# VERDICT r1 item 5 - what makes rocprofv3 --kernel-trace crash on the replayed TFLite plan?
# Discriminating runs, each its own process with its own timeout; logs in gpurun_out/tflgraph/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/tflgraph
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, then the program and its arguments
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/p_$name -o run --output-format csv -- "$@" > $OUT/$name.log 2>&1
  echo "exit=$?" >> $OUT/$name.log
  echo "== $name: $(grep -E 'graph replays|invoke 224|exit=' $OUT/$name.log | tr '\n' ' ')"
  rm -rf $OUT/p_$name
}
run A_engine_invoke_single_branch python3 $R/tools/graph_probe.py invoke 300
run B_engine_evaluate_forked python3 $R/tools/graph_probe.py evaluate 300
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run C_tflite_graph_no_packet_capture python3 $R/tools/time_tflite.py --graph 1 --invokes 300
run D_engine_invoke_no_packet_capture python3 $R/tools/graph_probe.py invoke 300
unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
echo done
