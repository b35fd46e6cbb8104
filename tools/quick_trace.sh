#!/bin/bash
# kernel trace of a few sampling steps + the per-launch timeline of the last step (tools/step_timeline.py)
#   tools/quick_trace.sh <tag> [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python bench.py --steps 8 --warmup 2 --repeats 1 --no-profile --cpu-budget 0 "$@" > $OUT/trace.log 2>&1
python tools/step_timeline.py "$OUT/t/*/*_kernel_trace.csv" > $OUT/timeline.txt 2>&1
tail -40 $OUT/timeline.txt
