set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd $R
for mode in v1 v2; do
  if [ $mode = v1 ]; then export CM_NO_CONV2=1; else unset CM_NO_CONV2; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/${mode}_trace -- python bench.py --steps 4 --warmup 1 --no-profile --cpu-budget 0 > gpurun_out/pmc/${mode}_trace.log 2>&1 || true
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc/${mode}_pmc1 -- python bench.py --steps 2 --warmup 0 --no-profile --cpu-budget 0 > gpurun_out/pmc/${mode}_pmc1.log 2>&1 || true
done
ls -R gpurun_out/pmc | head -40
