cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "f16" > gpurun_out/t_f16.log 2>&1; tail -3 gpurun_out/t_f16.log
export CM_DIAG=1
for rep in 1 2; do
for cfg in "CM_NO_UPS_F16=1" "X=1"; do
  v=$(env $cfg python bench.py --dtype f16 --config config/ATC_synthetic.yml --batch 32 --steps 100 --warmup 10 --cpu-budget 0 --no-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "[f16 24x72 $cfg] $v"
done; done
