"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/{trace,fetch,write,sq})
into the committed summaries under profiles/:
    <round>_kernel_stats.csv   (copy of the --stats table)
    <round>_pmc_summary.csv    per kernel: FETCH/WRITE KB per launch, corrected HBM bytes, MFMA busy,
                               VALU per MFMA, LDS bank-conflict share
    hbm_traffic.json           HBM bytes per launch of the 3x3x3 conv class (quoted by bench.py)
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128-B request
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section); counters come from separate passes."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")


def counters(sub):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


fetch, write, sq = counters("fetch"), counters("write"), counters("sq")
stats = glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"))
rows = []
conv_bytes, conv_n = 0.0, 0
for k in sorted(fetch, key=lambda k: -sum(fetch[k]["FETCH_SIZE"])):
    if "cm::" not in k:
        continue
    n = len(fetch[k]["FETCH_SIZE"])
    fk = sum(fetch[k]["FETCH_SIZE"]) / n
    wk = sum(write[k]["WRITE_SIZE"]) / max(1, len(write[k]["WRITE_SIZE"]))
    hbm = (2 * fk + wk) * 1024
    s = sq.get(k, {})

    def tot(name):
        return sum(s.get(name, [0.0]))
    mfma_busy = tot("SQ_VALU_MFMA_BUSY_CYCLES") / max(1.0, tot("GRBM_GUI_ACTIVE") / 8 * 1024) if s else float("nan")
    valu_per_mfma = tot("SQ_INSTS_VALU") / tot("SQ_INSTS_MFMA") if s and tot("SQ_INSTS_MFMA") > 0 else float("nan")
    lds_conf = tot("SQ_LDS_BANK_CONFLICT") / max(1.0, tot("SQ_LDS_IDX_ACTIVE")) if s else float("nan")
    rows.append((k, n, fk, wk, hbm, mfma_busy, valu_per_mfma, lds_conf))
    m3 = re.search(r"conv_mfma_kernel<\d+, \d+, (\d+)", k)
    if (m3 and m3.group(1) in ("27", "127", "8", "327", "427", "308")) or any(t in k for t in ("conv_smalln", "conv_wino", "conv_qr", "conv_f16d", "conv_ups", "conv_first_kernel")):
        conv_bytes += hbm * n
        conv_n += n
with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.csv"), "w") as fo:
    fo.write("kernel,launches_in_2_steps,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_corrected,"
             "mfma_busy_frac,valu_per_mfma,lds_bank_conflict_share\n")
    for r in rows:
        fo.write('"%s",%d,%.1f,%.1f,%.0f,%.3f,%.2f,%.3f\n' % r)
sys.path.insert(0, ROOT)
from bench import csrc_sha16  # noqa: E402  (fingerprint of the kernel sources this measurement belongs to)
json.dump({
    "kernel_class": "conv_wino[_p]_kernel + conv_qr[2]_kernel + conv_ups_kernel + conv_mfma_kernel<*,*,27|127|327|427|8|308> + conv_first_kernel + conv_smalln_kernel (all 3x3x3 conv launches)",
    "csrc_sha16": csrc_sha16(),
    "hbm_bytes_per_launch": conv_bytes / max(1, conv_n),
    "launches_counted": conv_n,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); bytes = "
              "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
    "source": f"profiles/{rnd}_pmc_summary.csv",
}, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.csv")).read())
print(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")).read())
