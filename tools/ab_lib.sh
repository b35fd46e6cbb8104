#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by 1-2 %): the current one and crowdmod-ddpm-4d_amd/libcrowdmod_hip_prev.so
# (built from an earlier commit, e.g. in a git worktree).   usage: tools/ab_lib.sh [bench.py flags]
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for which in prev cur; do
    if [ $which = prev ]; then export CM_LIB_PATH=$GRAFT_REPO_ROOT/crowdmod-ddpm-4d_amd/libcrowdmod_hip_prev.so; else unset CM_LIB_PATH; fi
    v=$(python bench.py --steps ${STEPS:-100} --warmup 10 --cpu-budget 0 --no-profile "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "[$which] $v"
  done
done
