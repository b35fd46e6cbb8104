#!/bin/bash
# Round profile of the headline bench command (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel time (committed under profiles/): once as ONE batch lane (--lanes 1:
#      one launch per layer, the configuration the HIP-event roofline in bench.py measures) and once as the default two lanes
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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --lanes 1 --steps $STEPS --warmup 2 --repeats 1 --no-profile --no-secondary --cpu-budget 0 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes2 -- python bench.py --steps $STEPS --warmup 2 --repeats 1 --no-profile --no-secondary --cpu-budget 0 > $OUT/trace_lanes2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --lanes 1 --steps 2 --warmup 0 --repeats 1 --no-profile --no-secondary --cpu-budget 0 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --lanes 1 --steps 2 --warmup 0 --repeats 1 --no-profile --no-secondary --cpu-budget 0 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python bench.py --lanes 1 --steps 2 --warmup 0 --repeats 1 --no-profile --no-secondary --cpu-budget 0 > $OUT/sq.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --lanes 1 --no-secondary --cpu-budget 0 > $OUT/bench_lanes1.json 2>> $OUT/bench.err
python bench.py --mode train > $OUT/bench_train.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python bench.py --mode train --steps 5 --warmup 2 --repeats 1 > $OUT/trace_train.log 2>&1
python bench.py --dtype f16 --config config/ATC_synthetic.yml --batch 32 --cpu-budget 0 > $OUT/bench_f16_24x72.json 2>> $OUT/bench.err
python bench.py --config config/ATC_synthetic.yml --batch 32 --cpu-budget 0 > $OUT/bench_f32_24x72.json 2>> $OUT/bench.err
python bench.py --config config/HERMES-CR-120.yml --channels 3 --cpu-budget 0 > $OUT/bench_f32_cr120.json 2>> $OUT/bench.err
python bench.py --batch 2 --cpu-budget 0 > $OUT/bench_b2.json 2>> $OUT/bench.err
python bench.py --dtype f32r --no-secondary --cpu-budget 0 > $OUT/bench_f32r.json 2>> $OUT/bench.err
python bench.py --dtype f32x --no-secondary --cpu-budget 0 > $OUT/bench_f32x.json 2>> $OUT/bench.err
# forward error against the reference's own outputs (tests/golden/fwd.npz): default plan (h2), six-term form (CM_NO_H2), relaxed plan
{ echo "# default plan (h2 where the input is bounded)"; python tools/experiments/fwd_err.py; echo "# CM_DIAG=1 CM_NO_H2=1 (six-term bf16 everywhere: round 3's arithmetic)"; CM_DIAG=1 CM_NO_H2=1 python tools/experiments/fwd_err.py; echo "# relaxed plan (f32r)"; python tools/experiments/fwd_err.py f32r; } > $OUT/fwd_err.txt 2>/dev/null
# same-box A/B of the default plan against round 3's arithmetic
{ bash tools/ab_env.sh "" "CM_NO_H2=1" "" "CM_NO_H2=1"; } > $OUT/ab_h2.txt 2>&1
python tools/step_timeline.py "$OUT/trace/*/*_kernel_trace.csv" > $OUT/step_timeline.txt 2>&1
python tools/train_timeline.py "$OUT/trace_train/*/*_kernel_trace.csv" > $OUT/train_step_timeline.txt 2>&1
tail -1 $OUT/trace.log; ls $OUT/*/*/ | head -20
