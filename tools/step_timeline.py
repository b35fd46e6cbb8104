"""Print the kernel timeline of the last sampling step of a rocprofv3 --kernel-trace CSV and the
per-kernel-name totals of that step."""
import collections
import csv
import glob
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_seq/*/*_kernel_trace.csv"
f = sorted(glob.glob(pat))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sampler_step" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
tot = collections.defaultdict(lambda: [0, 0.0])
quiet = len(sys.argv) > 2
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("cm::", "").replace("void ", "").split("(")[0][:40]
    tot[name][0] += 1
    tot[name][1] += (e - s) / 1e3
    if not quiet:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {name:42s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
print("step span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} {n:4d} launches {us:9.1f} us")
