#!/bin/bash
# rocprofv3 passes for the sweep at c3 (run on the GPU box from the repo root): kernel trace + stats, then PMC groups in separate passes.
# usage: tools/prof_sweep.sh <tag> [extra bench.py args]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
CMD="python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.log
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -- $CMD > /dev/null 2> $OUT/pmc_$name.log
done
python3 tools/pmc_summary.py c3 $TAG $OUT/pmc_summary.json $OUT/pmc_* $OUT/trace > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*.csv" -size +2M -delete
