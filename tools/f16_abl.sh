cd $GRAFT_REPO_ROOT
export TT_F16=1 TT_H=24 TT_W=72 TT_B=32 CM_DIAG=1
for d in 0 2 4 128 130 6; do
  echo "== dbg $d"; CM_CONV_DBG=$d python tools/time_tiles.py "conv_1:0:0:0:0" "conv_2:0:0:0:0" 2>&1 | grep -v rejected | awk '{print $1, $5, $6}' | head -30
done
