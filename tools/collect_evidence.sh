#!/bin/bash
# (a step = 10 frames per shard since round 4: the step counts below are a tenth of round 3's)
# Round evidence on the GPU box: rocprofv3 kernel stats, PMC passes (each in its own run, --kernel-trace only), bench JSON
# lines.  Usage: bash tools/collect_evidence.sh <round tag, e.g. r02>; results under gpurun_out/<tag>/ (copy the
# summaries into profiles/).
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
ONE="python bench.py --steps 3 --warmup 1 --streams 1 --pipeline 0 --no-cpu-baseline --no-secondary --no-parity"
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -o bench -- $DRV > $OUT/prof_default.log 2>&1
# QUICK=1: only the passes whose numbers moved since the last full collection (kernel stats, config 4, the bench lines); the PMC
# passes of the 5-keyframe configurations are skipped
Q=${QUICK:-0}
# the same with the launch path (no resident grid): per-launch k_ba_lm durations of the throughput cut
[ $Q = 1 ] || MVO_BA_SERVICE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default_launches -o bench -- $DRV > $OUT/prof_default_launches.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_streams1 -o bench -- python bench.py --steps 6 --warmup 1 --streams 1 --pipeline 0 --no-cpu-baseline --no-secondary --no-parity > $OUT/prof_streams1.log 2>&1
[ $Q = 1 ] || timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $ONE > $OUT/pmc_fetch.log 2>&1
[ $Q = 1 ] || timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $ONE > $OUT/pmc_write.log 2>&1
[ $Q = 1 ] || timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -o bench -- $ONE > $OUT/pmc_mfma.log 2>&1
[ $Q = 1 ] || python tools/pmc_summary.py fetch_write $OUT/pmc_fetch/bench_counter_collection.csv $OUT/pmc_write/bench_counter_collection.csv $OUT/pmc_streams1_fetch_write_size.csv "$ONE"
# The RESIDENT GRID (the kernel bench.py's roofline block names): the same passes on the headline workload with every window on
# the grid from the first frame (MVO_BA_SERVICE=2, no warm-up: the windows the grid solved = the windows of the run, printed in
# the JSON line of each pass): FETCH / WRITE per window, matrix-core busy cycles per SIMD-cycle of the CUs its windows occupied
# (resident cycles of the windows x workgroups x 4 SIMDs), LDS conflicts
DEF="env MVO_BA_SERVICE=2 python bench.py --steps 10 --warmup 0 --no-cpu-baseline --no-secondary --no-parity"
[ $Q = 1 ] || timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_default -o bench -- $DEF > $OUT/pmc_fetch_default.log 2>&1
[ $Q = 1 ] || timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_default -o bench -- $DEF > $OUT/pmc_write_default.log 2>&1
[ $Q = 1 ] || python tools/pmc_summary.py fetch_write $OUT/pmc_fetch_default/bench_counter_collection.csv $OUT/pmc_write_default/bench_counter_collection.csv $OUT/pmc_fetch_write_size_per_kernel.csv "$DEF" $((32 * 10 * 10))
[ $Q = 1 ] || timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma_grid -o bench -- $DEF > $OUT/pmc_mfma_grid.log 2>&1
[ $Q = 1 ] || python tools/pmc_summary.py grid $OUT/pmc_mfma_grid/bench_counter_collection.csv $OUT/pmc_mfma_grid.log $OUT/pmc_grid_mfma_busy.txt "$DEF"
[ $Q = 1 ] || timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds_grid -o bench -- $DEF > $OUT/pmc_lds_grid.log 2>&1
[ $Q = 1 ] || python tools/pmc_summary.py table $OUT/pmc_lds_grid/bench_counter_collection.csv $OUT/pmc_grid_lds.txt "$DEF"
[ $Q = 1 ] || python tools/pmc_summary.py table $OUT/pmc_mfma/bench_counter_collection.csv $OUT/pmc_mfma_busy.txt "$ONE"
[ $Q = 1 ] || timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds -o bench -- $ONE > $OUT/pmc_lds.log 2>&1
[ $Q = 1 ] || python tools/pmc_summary.py table $OUT/pmc_lds/bench_counter_collection.csv $OUT/pmc_lds.txt "$ONE"
# BASELINE configs[3] (S1242 / 4000 kp / BA10): the same passes on that config
C4="python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-parity"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_config4 -o bench -- $C4 > $OUT/prof_config4.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_config4 -o bench -- $C4 > $OUT/pmc_fetch_config4.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_config4 -o bench -- $C4 > $OUT/pmc_write_config4.log 2>&1
python tools/pmc_summary.py fetch_write $OUT/pmc_fetch_config4/bench_counter_collection.csv $OUT/pmc_write_config4/bench_counter_collection.csv $OUT/config4_pmc_fetch_write_size_per_kernel.csv "$C4"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma_config4 -o bench -- $C4 > $OUT/pmc_mfma_config4.log 2>&1
python tools/pmc_summary.py table $OUT/pmc_mfma_config4/bench_counter_collection.csv $OUT/config4_pmc_mfma_busy.txt "$C4"
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
timeout 200 python bench.py --track --no-cpu-baseline --no-secondary > $OUT/bench_track.json 2> $OUT/bench_track.err
# tracking rows: clean per-kernel durations (one shard, serial loop)
[ $Q = 1 ] || timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_track1 -o bench -- python bench.py --track --steps 6 --warmup 1 --streams 1 --pipeline 0 --no-cpu-baseline --no-secondary > $OUT/prof_track1.log 2>&1
timeout 200 python bench.py --width 1242 --height 375 --max-kp 4000 --ba-poses 10 --ba-points 4000 --streams 8 --steps 6 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_config4.json 2> $OUT/bench_config4.err
# the frame kernels by SQ counters (instructions, wave cycles, where the waves wait), headline mode, no solver on the device
bash tools/pmc_extract.sh $TAG/frame_kernels > $OUT/frame_kernels.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
# gpurun merges at most 64 MiB back: the raw per-dispatch tables (kernel traces / counter collections of ~40 k dispatches per
# pass) are summarised above -- only the summaries, the *_kernel_stats.csv and the logs travel
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_counter_collection.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
find $OUT -name ".shipped.so" -delete
du -sh $OUT
ls -la $OUT | head -30
cat $OUT/pmc_fetch_write_size_per_kernel.csv $OUT/pmc_streams1_fetch_write_size.csv $OUT/pmc_mfma_busy.txt $OUT/config4_pmc_fetch_write_size_per_kernel.csv $OUT/config4_pmc_mfma_busy.txt 2>/dev/null
for d in prof_default prof_default_launches prof_streams1 prof_config4 prof_track1; do f=$(find $OUT/$d -name "*kernel_stats.csv" 2>/dev/null | head -1); echo "== $d"; [ -n "$f" ] && cut -c1-160 "$f" | head -8; done < /dev/null
tail -c 600 $OUT/bench_default.json < /dev/null
