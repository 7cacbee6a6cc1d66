#!/bin/bash
# Round evidence on the GPU box, most valuable first, every step bounded (a step that hangs is killed with its whole process group
# and the next one starts on a clean device): GPU test log, the driver's bench line, rocprofv3 kernel stats, PMC passes (each in its
# own run, --kernel-trace only).  Usage: bash tools/collect_evidence.sh <tag> [budget seconds]; results under gpurun_out/<tag>/
# (only summaries: gpurun merges at most 64 MiB back) -- copy them into profiles/.
# PMC passes on the RESIDENT GRID come last: rocprofv3's counter collection next to a kernel that never ends hung once in three runs
# (round 5; the pass is killed after 100 s then, the launch-path passes of the same window body are the fallback).
set -u
TAG=${1:-rXX}
BUDGET=${2:-700}
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
left() { echo $(( BUDGET - ($(date +%s) - T0) )); }
# run_bounded <seconds> <log> <command...>: own session; the whole process group is killed when the time is up
run_bounded() {
  local t=$1 log=$2; shift 2
  if [ $(left) -lt $(( t / 2 )) ]; then echo "[skip: budget] $*" | tee -a $OUT/skipped.txt; return 99; fi
  setsid "$@" > $log 2>&1 &
  local p=$!
  ( sleep $t; kill -KILL -- -$p 2>/dev/null ) &
  local w=$!
  wait $p; local rc=$?
  kill $w 2>/dev/null; wait $w 2>/dev/null
  echo "[$(left) s left] rc=$rc $(echo "$*" | cut -c1-140)"
  return $rc
}
trim() { find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_agent_info.csv" -delete; }
NOX="--no-cpu-baseline --no-secondary --no-parity"
DRV="python bench.py --gpus 1 --steps 20 --warmup 5"
ONE="python bench.py --steps 3 --warmup 1 --streams 1 --pipeline 0 $NOX"
C4="python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 3 --warmup 1 $NOX"
pmc() {  # pmc <name> <seconds> "<counters>" <command...>   -> $OUT/<name>.txt (per-kernel table)
  local name=$1 t=$2 ctr=$3; shift 3
  run_bounded $t $OUT/$name.log rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$name -o bench -- "$@" || return 1
  python tools/pmc_summary.py table $OUT/$name/bench_counter_collection.csv $OUT/$name.txt "$*"
}
sec0() {
# 0. what the FP64 roofline is priced against, measured on this box (tools/probes/fp64_peak_probe.hip)
run_bounded 120 $OUT/fp64_peak_probe.txt bash -c "hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_peak_probe tools/probes/fp64_peak_probe.hip && /tmp/fp64_peak_probe"
}
sec1() {
# 1. tests + the bench lines
run_bounded 300 $OUT/pytest_gpu.log python -m pytest tests -m gpu -q
run_bounded 400 $OUT/bench_driver_command.err bash -c "$DRV > $OUT/bench_driver_command.json"
run_bounded 150 $OUT/bench_track.err bash -c "python bench.py --track $NOX > $OUT/bench_track.json"
run_bounded 150 $OUT/bench_config4.err bash -c "python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 6 --warmup 1 $NOX > $OUT/bench_config4.json"
}
sec2() {
# 2. kernel stats (no counters): driver command, launch path, one sequence, config 4, tracking rows
# (HAVE=1: the driver-command / launch-path / one-sequence stats and the single-window FETCH / WRITE / MFMA passes exist already)
[ ${HAVE:-0} = 1 ] || run_bounded 200 $OUT/prof_default.log rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -o bench -- $DRV $NOX
[ ${HAVE:-0} = 1 ] || run_bounded 200 $OUT/prof_default_launches.log env MVO_BA_SERVICE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default_launches -o bench -- $DRV $NOX
[ ${HAVE:-0} = 1 ] || run_bounded 150 $OUT/prof_streams1.log rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_streams1 -o bench -- python bench.py --steps 6 --warmup 1 --streams 1 --pipeline 0 $NOX
run_bounded 150 $OUT/prof_config4.log rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_config4 -o bench -- $C4
run_bounded 150 $OUT/prof_track1.log rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_track1 -o bench -- python bench.py --track --steps 6 --warmup 1 --streams 1 --pipeline 0 $NOX
trim
}
sec3() {
# 3. counters, kernels that end: the single-window launch (latency cut), the throughput cut on the launch path under the headline load
# (the same window body and flavour as the resident grid's), the frame kernels, config 4
[ ${HAVE:-0} = 1 ] || pmc pmc_mfma 120 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $ONE
pmc pmc_lds 120 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" $ONE
[ ${HAVE:-0} = 1 ] || run_bounded 120 $OUT/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $ONE
[ ${HAVE:-0} = 1 ] || run_bounded 120 $OUT/pmc_write.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $ONE
[ ${HAVE:-0} = 1 ] || python tools/pmc_summary.py fetch_write $OUT/pmc_fetch/bench_counter_collection.csv $OUT/pmc_write/bench_counter_collection.csv $OUT/pmc_streams1_fetch_write_size.csv "$ONE"
LP="env MVO_BA_SERVICE=0 python bench.py --steps 10 --warmup 2 $NOX"
pmc pmc_launch_path_mfma 150 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $LP
pmc pmc_launch_path_lds 150 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" $LP
run_bounded 150 $OUT/frame_kernels.log bash tools/pmc_extract.sh $TAG/frame_kernels
trim
}
sec4() {
# 4. counters on the resident grid itself (MVO_BA_SERVICE=2, no warm-up: the windows of the run = the windows the grid solved)
DEF="env MVO_BA_SERVICE=2 python bench.py --steps 10 --warmup 0 $NOX"
if run_bounded 100 $OUT/pmc_mfma_grid.log rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma_grid -o bench -- $DEF; then
  python tools/pmc_summary.py grid $OUT/pmc_mfma_grid/bench_counter_collection.csv $OUT/pmc_mfma_grid.log $OUT/pmc_grid_mfma_busy.txt "$DEF"
fi
pmc pmc_grid_lds 100 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" $DEF
if run_bounded 100 $OUT/pmc_fetch_default.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_default -o bench -- $DEF && \
   run_bounded 100 $OUT/pmc_write_default.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_default -o bench -- $DEF; then
  python tools/pmc_summary.py fetch_write $OUT/pmc_fetch_default/bench_counter_collection.csv $OUT/pmc_write_default/bench_counter_collection.csv $OUT/pmc_fetch_write_size_per_kernel.csv "$DEF" $((32 * 10 * 10))
fi
}
sec5() {
# 5. config 4 counters
run_bounded 150 $OUT/pmc_fetch_config4.log rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_config4 -o bench -- $C4
run_bounded 150 $OUT/pmc_write_config4.log rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_config4 -o bench -- $C4
python tools/pmc_summary.py fetch_write $OUT/pmc_fetch_config4/bench_counter_collection.csv $OUT/pmc_write_config4/bench_counter_collection.csv $OUT/config4_pmc_fetch_write_size_per_kernel.csv "$C4"
pmc config4_pmc_mfma_busy 150 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $C4
trim
}
sec6() {
run_bounded 600 $OUT/bench_default.err bash -c "python bench.py > $OUT/bench_default.json"
}
for sct in ${SECTIONS:-0 1 2 3 4 5 6}; do sec$sct; done
trim
du -sh $OUT; ls $OUT
tail -3 $OUT/pytest_gpu.log
for f in $OUT/pmc_grid_mfma_busy.txt $OUT/pmc_fetch_write_size_per_kernel.csv $OUT/pmc_launch_path_mfma.txt; do [ -f $f ] && cat $f; done
