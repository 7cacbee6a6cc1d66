#!/bin/bash
# SQ counters of the frame kernels (extraction + matching), one sequence shard in the MODE of the headline shards (throughput:
# k_blur + k_brief_sample), no solver on the device: instructions, wave cycles and where the waves wait, per kernel.
#   bash tools/pmc_extract.sh <tag> [LIB=<alt libmvo_hip.so>]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:?tag}; ALT=${2#LIB=}
O=gpurun_out/$TAG; mkdir -p $O
SO=monocular-visual-odometry_amd/csrc/libmvo_hip.so
if [ -n "$ALT" ]; then cp $SO $O/.shipped.so && cp "$ALT" $SO; fi
CMD="python bench.py --streams 1 --steps 4 --warmup 1 --pipeline 0 --ba-mode none --ba-cut throughput --no-cpu-baseline --no-secondary --no-parity"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o bench -- $CMD > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o bench -- $CMD > $O/pmc_sq2.log 2>&1
if [ -n "$ALT" ]; then cp $O/.shipped.so $SO; fi
for d in pmc_sq pmc_sq2; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py table $f $O/$d.txt "$CMD" && cat $O/$d.txt; done
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_counter_collection.csv" -delete; find $O -name ".shipped.so" -delete
tail -3 $O/pmc_sq.log
