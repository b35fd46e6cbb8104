#!/bin/bash
# FETCH_SIZE per 3x3x3 conv launch with and without the XCD-aware tile order (own --pmc passes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for tag in remap noremap; do
  mkdir -p gpurun_out/prof_fetch_$tag
  if [ $tag = noremap ]; then export CM_DIAG=1 CM_CONV_DBG=4096; else unset CM_CONV_DBG; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch_$tag -- python bench.py --lanes 1 --no-secondary --steps 2 --warmup 0 --no-profile --cpu-budget 0 > gpurun_out/prof_fetch_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os, re
for tag in ("remap", "noremap"):
    f = max(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], f"gpurun_out/prof_fetch_{tag}/*/*_counter_collection.csv")), key=os.path.getmtime)
    tot = n = 0
    for r in csv.DictReader(open(f)):
        m = re.search(r"conv_mfma_kernel<\d+, \d+, (\d+)", r["Kernel_Name"])
        if m and m.group(1) in ("27", "127", "8"):
            tot += float(r["Counter_Value"]); n += 1
    print(tag, "conv launches", n, "FETCH_SIZE KB/launch %.0f -> HBM read %.1f MB/launch (x2 corrected)" % (tot / n, 2 * tot / n / 1024))
PY
