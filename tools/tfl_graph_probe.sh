#!/bin/bash
# the TFLite plan as a replayed hipGraph, without and with rocprofv3 --kernel-trace.
# Each step is bounded by its own timeout; logs go to gpurun_out/tflgraph/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/tflgraph
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[1] graph replay, no profiler, 600 invokes"
timeout -k 10 240 python3 $R/tools/time_tflite.py --graph 1 --invokes 600 > $OUT/plain_graph1.log 2>&1; echo "exit=$?" >> $OUT/plain_graph1.log
tail -3 $OUT/plain_graph1.log
echo "[2] eager under rocprofv3 --kernel-trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_eager -o run --output-format csv -- python3 $R/tools/time_tflite.py --graph 0 --invokes 300 > $OUT/prof_graph0.log 2>&1; echo "exit=$?" >> $OUT/prof_graph0.log
tail -3 $OUT/prof_graph0.log
echo "[3] graph replay under rocprofv3 --kernel-trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_graph -o run --output-format csv -- python3 $R/tools/time_tflite.py --graph 1 --invokes 300 > $OUT/prof_graph1.log 2>&1; echo "exit=$?" >> $OUT/prof_graph1.log
tail -25 $OUT/prof_graph1.log
find $OUT -name "*kernel_stats.csv" | head; rm -rf $OUT/prof_eager/*/*kernel_trace.csv $OUT/prof_graph/*/*kernel_trace.csv 2>/dev/null; find $OUT -name "*.csv" -size +2M -delete
echo done
