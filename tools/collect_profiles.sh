#!/bin/bash
# Regenerates every artifact under profiles/ for one round tag (default r01) on the GPU box:
#   bench line, rocprofv3 kernel stats of the same bench command, per-launch hipEvent tables,
#   PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) -> traffic JSON, SQ counters of the conv kernels.
# Usage (inside gpurun): bash tools/collect_profiles.sh r01     -> writes gpurun_out/profiles_r01/
set -e -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[1/6] PMC FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmcF -o run --output-format csv -- python3 $R/tools/pmc_run.py 64 > $OUT/pmcF.log 2>&1
echo "[2/6] PMC WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmcW -o run --output-format csv -- python3 $R/tools/pmc_run.py 64 > $OUT/pmcW.log 2>&1
F=$(find $OUT/pmcF -name "*counter_collection.csv" | sort | tail -1)
W=$(find $OUT/pmcW -name "*counter_collection.csv" | sort | tail -1)
python3 $R/tools/make_traffic.py $F $W 64 $OUT/${TAG}_traffic.json > /dev/null
cp $OUT/${TAG}_traffic.json $R/profiles/${TAG}_traffic.json   # bench.py reads roofline.traffic from profiles/
echo "[3/6] bench"; python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
echo "[4/6] rocprofv3 kernel stats of the bench command"
# (the batch-64 leg alone: the kernel averages of this CSV are then the headline step's own, as the bench line's roofline is)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-batch1 --no-configs4 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp "$(find $OUT/stats -name "*kernel_stats.csv" | sort | tail -1)" $OUT/${TAG}_rocprofv3_kernel_stats.csv
# (configs[4]'s share under the profiler too: its layer-3 launches are the eight-wave bneck_xn_f16)
rocprofv3 --kernel-trace --stats -d $OUT/stats4 -o run --output-format csv -- python3 $R/bench.py --backbone 101 --size 700 --precision fp8 --batch 8 --no-cpu-baseline --no-batch1 --no-configs4 > $OUT/bench4_under_rocprof.json 2> $OUT/stats4.log
cp "$(find $OUT/stats4 -name "*kernel_stats.csv" | sort | tail -1)" $OUT/${TAG}_rocprofv3_kernel_stats_configs4.csv
echo "[5/6] per-launch tables, the configs[4] lines, the reference-path timings"
python3 $R/tools/profile_table.py --batch 64 > $OUT/${TAG}_hipevent_per_launch_batch64.txt
python3 $R/tools/profile_table.py --batch 1 > $OUT/${TAG}_hipevent_per_launch_batch1.txt
python3 $R/tools/profile_table.py --backbone 101 --size 700 --batch 64 --precision fp8 > $OUT/${TAG}_hipevent_per_launch_yolact700_r101_fp8_batch64.txt
python3 $R/bench.py --backbone 101 --size 700 --precision fp8 --batch 64 --no-batch1 > $OUT/${TAG}_bench_yolact700_r101_fp8_batch64.json 2> $OUT/bench_fp8.err
python3 $R/bench.py --backbone 101 --size 700 --batch 64 --no-batch1 --no-configs4 --no-cpu-baseline > $OUT/${TAG}_bench_yolact700_r101_f16_batch64.json 2> $OUT/bench_r101.err
python3 $R/bench.py --backbone 101 --size 700 --precision fp8 --batch 8 --no-batch1 --no-cpu-baseline > $OUT/${TAG}_bench_yolact700_r101_fp8_batch8.json 2>> $OUT/bench_fp8.err
{ python3 $R/tools/time_steps.py 1 2 4 8 16 32 64; python3 $R/tools/time_classify.py; python3 $R/tools/time_tflite.py; python3 $R/tools/time_scene.py; python3 $R/tools/time_h2d.py; } > $OUT/${TAG}_timings.txt 2>&1
python3 $R/tools/ab_chain.py 64 0 1 65 > $OUT/${TAG}_ab_chain.txt 2>&1
# round 4: the .tflite executor under the profiler (fused plan and one launch per operator), its fused / unfused / graph timings,
# the cost model of one int8-MFMA conv launch
for f in 1 0; do
  rocprofv3 --kernel-trace --stats -d $OUT/tfl$f -o run --output-format csv -- python3 $R/tools/time_tflite_fuse.py fuse=$f,graph=0,group=$f > $OUT/tfl$f.log 2>&1
done
cp "$(find $OUT/tfl1 -name "*kernel_stats.csv" | sort | tail -1)" $OUT/${TAG}_tflite_rocprofv3_kernel_stats.csv
cp "$(find $OUT/tfl0 -name "*kernel_stats.csv" | sort | tail -1)" $OUT/${TAG}_tflite_unfused_rocprofv3_kernel_stats.csv
python3 $R/tools/tfl_trace_span.py "$(find $OUT/tfl1 -name "*kernel_trace.csv" | sort | tail -1)" 58 > $OUT/${TAG}_tflite_device_span.txt 2>&1 || true
python3 $R/tools/tfl_trace_span.py "$(find $OUT/tfl0 -name "*kernel_trace.csv" | sort | tail -1)" 132 >> $OUT/${TAG}_tflite_device_span.txt 2>&1 || true
{ python3 $R/tools/time_tflite_fuse.py; python3 $R/tools/study/tfl_host_vs_device.py; python3 $R/tools/study/tfl_conv_steps.py 3; python3 $R/tools/study/tfl_conv_steps.py 2; } > $OUT/${TAG}_tflite_timings.txt 2>&1
# ... the per-launch timeline of the plan (batch-2 invokes) with the register-fed int8 convolutions (tfl_dot = 3, default) and the LDS tiles (2)
for dot in 3 2; do
  rocprofv3 --kernel-trace -d $OUT/tl$dot -o run --output-format csv -- python3 $R/tools/study/tfl_layer_run.py tfl_dot=$dot,tfl_group=$((dot == 3)) > $OUT/tl$dot.log 2>&1
  python3 $R/tools/study/tfl_layer_table.py "$(find $OUT/tl$dot -name "*kernel_trace.csv" | sort | tail -1)" "$(grep LAUNCHES $OUT/tl$dot.log | cut -d' ' -f2)" > $OUT/${TAG}_tflite_layer_table_dot$dot.txt 2>&1 || true
done
for m in "" "--fp8-per-tensor"; do python3 $R/bench.py --no-batch1 --no-tflite --cpu-budget 2 --steps 5 $m > $OUT/${TAG}_bench_fp8_scales${m}.json 2>> $OUT/bench_fp8.err; done
python3 $R/tools/fault_audit.py > $OUT/${TAG}_fault_audit_dump.txt 2>&1
python3 $R/bench.py --gpus 2 --single-process --batch 32 --steps 10 --warmup 3 > $OUT/${TAG}_bench_single_process_2members.json 2> $OUT/bench_sp.err
echo "[6/6] SQ counters"
{
  echo "# rocprofv3 --pmc passes over tools/pmc_run.py 16 (batch 16, two eager forwards); sums over all dispatches of each kernel"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM"; do
    tag=$(echo $set | md5sum | cut -c1-6)
    rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$tag -o run --output-format csv -- python3 $R/tools/pmc_run.py 16 > $OUT/pmc_$tag.log 2>&1
    python3 $R/tools/pmc_summary.py "$(find $OUT/pmc_$tag -name "*counter_collection.csv" | sort | tail -1)"
  done
} > $OUT/${TAG}_pmc_kernels.txt
rm -rf $OUT/stats $OUT/stats4 $OUT/pmcF $OUT/pmcW $OUT/tfl0 $OUT/tfl1 $OUT/tl3 $OUT/tl2; find $OUT -maxdepth 1 -type d -name 'pmc_*' -exec rm -rf {} +
# Round 5: the GPU test run of THIS tree is part of the collection, green or not (tests/conftest.py arms faulthandler:
# a crash leaves every thread's Python stack in it), and every log of the collection that is not green is kept under a name that says so -
# round 4's host segfault was lost because only the green re-run's log survived. Copy ${TAG}_* (all of them) into profiles/.
( cd $R && python3 -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "exit code $?" >> $OUT/${TAG}_pytest_gpu.txt ) || true
for f in $OUT/*.log $OUT/*.err $OUT/${TAG}_pytest_gpu.txt; do
  [ -f "$f" ] || continue
  if grep -q -E "Traceback|Segmentation fault|Fatal Python error|Memory access fault|core dumped|Aborted|FAILED|exit code [1-9]" "$f"; then
    cp "$f" "$OUT/${TAG}_NONGREEN_$(basename "$f").txt"
  fi
done
echo done
