#!/bin/bash
# Same-box timing of single conv layers under several library builds:
#   tools/ab_layers.sh "<label> <label> ..." cur ab_libs/lib_x.so ...
cd $GRAFT_REPO_ROOT
keys=$1; shift
for lib in "$@"; do
  if [ $lib = cur ]; then unset CM_LIB_PATH; else export CM_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
  for k in $keys; do
    echo "[$lib] $(python tools/time_tiles.py $k:0:0:0:0 2>&1 | tail -1 | cut -c1-90)"
  done
done
