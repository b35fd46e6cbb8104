#!/bin/bash
# Same-box A/B of several builds of the library: tools/ab_libs.sh libA.so libB.so ... (paths relative to the repo root;
# "cur" = the in-tree build).  REPS (default 2) rounds, bench flags through BENCH_FLAGS.
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "$@"; do
    if [ $lib = cur ]; then unset CM_LIB_PATH; else export CM_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
    v=$(python bench.py --steps ${STEPS:-100} --warmup 10 --cpu-budget 0 --no-profile --no-secondary $BENCH_FLAGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "[$lib] $v"
  done
done
