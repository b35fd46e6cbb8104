cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "forward or unet or loop or train" > gpurun_out/t_ups.log 2>&1; tail -3 gpurun_out/t_ups.log
export CM_DIAG=1
for cfg in "CM_NO_UPS=1" "" "CM_UPS_OCC=3" "CM_UPS_OCC=4" "CM_UPS_TILE=4,6,6,0 CM_UPS_OCC=3" "CM_UPS_TILE=4,6,6,0 CM_UPS_OCC=2" "CM_UPS_NB=2"; do
  echo "== $cfg"
  env $cfg python tools/time_tiles.py "upsample:0:0:0:0" 2>&1 | grep -v amdgpu.ids | awk '{print $1, $5, $6}'
  env $cfg python bench.py --steps 100 --warmup 10 --cpu-budget 0 --no-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
