import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last chain: find the last 3 interpolation kernels
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("interpolation_kernel") or "interpolation_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
prev_end = None; tot_k = 0; tot_gap = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-60s dur %7.1f us  gap before %6.1f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, gap))
    tot_k += (e - s) / 1e3; tot_gap += max(gap, 0); prev_end = max(e, prev_end or 0)
print("kernels", len(rows[a:b]), "kernel time", round(tot_k, 1), "gaps", round(tot_gap, 1), "span", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
