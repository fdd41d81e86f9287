#!/bin/bash
# rocprofv3 --kernel-trace --stats of the stages besides the sweep (run on the GPU box from the repo root)
set -u
OUT=gpurun_out/prof_stages
mkdir -p $OUT
export TMPDIR=/tmp
for job in raster_scaling time_stages time_filter time_flow; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$job -- python3 tools/$job.py > $OUT/$job.out 2> $OUT/$job.log
  find $OUT/$job -name "*kernel_stats.csv" -exec cp {} $OUT/${job}_kernel_stats.csv \;
  rm -rf $OUT/$job
done
python3 tools/time_filter.py > $OUT/time_filter_unprofiled.out 2>&1
python3 tools/time_stages.py > $OUT/time_stages_unprofiled.out 2>&1
