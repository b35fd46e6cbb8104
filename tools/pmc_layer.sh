#!/bin/bash
# SQ counters of ONE conv layer launched back to back (tools/time_tiles.py "<label>:0:0:0:0" keeps the op's own tile):
#   tools/pmc_layer.sh <label substring> <out tag>
# Three separate --pmc passes (8 SQ slots each); no trace domains next to --pmc (pool rule).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KEY=$1; OUT=$R/gpurun_out/pmc_$2
mkdir -p $OUT; cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python tools/time_tiles.py $KEY:0:0:0:0 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/p2 -- python tools/time_tiles.py $KEY:0:0:0:0 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM --output-format csv -d $OUT/p3 -- python tools/time_tiles.py $KEY:0:0:0:0 > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if not any(s in k for s in ("conv_mfma", "conv_wino", "conv_first", "smalln", "attn_head", "wgrad")): continue
    n = len(next(iter(c.values())))
    if n < 10: continue
    print(k, "dispatches", n)
    for name in sorted(c): print("   %-28s %14.0f per dispatch" % (name, sum(c[name]) / len(c[name])))
PY
