#!/bin/bash
# Round profile of the headline bench command (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel time (committed under profiles/)
#   2. rocprofv3 --pmc FETCH_SIZE          -> HBM read traffic   (own pass, no trace domains)
#   3. rocprofv3 --pmc WRITE_SIZE          -> HBM write traffic  (own pass)
#   4. rocprofv3 --pmc SQ_* (MFMA busy)    -> matrix-core utilisation
# The program after `--` is python itself (no env/bash hop: see the pool's exec rule).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${1:-r1}
mkdir -p $OUT
cd $R
STEPS=${STEPS:-10}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps $STEPS --warmup 2 --no-profile --cpu-budget 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 2 --warmup 0 --no-profile --cpu-budget 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 2 --warmup 0 --no-profile --cpu-budget 0 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python bench.py --steps 2 --warmup 0 --no-profile --cpu-budget 0 > $OUT/sq.log 2>&1
python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/trace.log; ls $OUT/*/*/ | head -20
