#!/bin/bash
# A/B of environment-selected kernel configurations within ONE gpurun call (boxes differ by a few %).
# usage: tools/ab_env.sh "VAR=1 VAR2=3" "VAR=2" ...   ("" = defaults)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  v=$(env CM_DIAG=1 $cfg timeout -k 10 200 python bench.py --steps ${STEPS:-100} --warmup 10 --cpu-budget 0 --no-profile --no-secondary 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "[$cfg] $v"
done
