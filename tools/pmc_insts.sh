#!/bin/bash
# Dynamic instruction mix per kernel (own --pmc pass, no trace domains): instructions per wave by type.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_insts
mkdir -p $OUT
cd $R
CM_LANES=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $OUT/a -- python bench.py --lanes 1 --no-secondary --steps 2 --warmup 0 --no-profile --cpu-budget 0 > $OUT/a.log 2>&1
python - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/prof_insts/a/*/*_counter_collection.csv"))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("cm::", "").replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
print("%-34s %6s %9s | per wave: %7s %7s %7s %6s %6s %6s %6s" % ("kernel", "calls", "waves", "VALU", "SALU", "MFMA", "LDS", "VMrd", "VMwr", "SMEM"))
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_MFMA"]):
    w = max(1.0, c["SQ_WAVES"])
    print("%-34s %6d %9d | %16.0f %7.0f %7.0f %6.0f %6.0f %6.0f %6.0f" % (k[:34], n[k], w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w,
          c["SQ_INSTS_MFMA"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w, c["SQ_INSTS_VMEM_WR"] / w, c["SQ_INSTS_SMEM"] / w))
PY
