"""Per-kernel totals of the last training step of a rocprofv3 --kernel-trace CSV (steps delimited by adam_kernel)."""
import collections
import csv
import glob
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_train/*/*_kernel_trace.csv"
rows = list(csv.DictReader(open(sorted(glob.glob(pat))[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
tot = collections.defaultdict(lambda: [0, 0.0])
verbose = len(sys.argv) > 2
for r in rows[a + 1:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("cm::", "").replace("void ", "").split("(")[0][:50]
    tot[n][0] += 1
    tot[n][1] += (e - s) / 1e3
    if verbose and sys.argv[2] in n:
        print("%9.1f dur %7.1f %-40s grid %sx%sx%s" % ((s - t0) / 1e3, (e - s) / 1e3, n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"]))
print("step span %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a - 1))
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %4d  %8.1f us  avg %7.1f" % (k, n, us, us / n))
