"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: total, and the largest gaps with the kernels on either side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); end = t0; gaps = []
busy = 0
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end and i:
        gaps.append((s - end, rows[i - 1]["Kernel_Name"][:60], r["Kernel_Name"][:60], (s - t0) / 1e6))
    busy += max(0, e - max(s, end)); end = max(end, e)
tot = (end - t0) / 1e6
print("span %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps" % (tot, busy / 1e6, sum(g[0] for g in gaps) / 1e6, len(gaps)))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
big = [g for g in gaps if g[0] > thr * 1e3]
print("gaps > %.0f us: %d, %.2f ms" % (thr, len(big), sum(g[0] for g in big) / 1e6))
for g in sorted(big, reverse=True)[:25]:
    print("  %8.1f us at %8.2f ms   %s  ->  %s" % (g[0] / 1e3, g[3], g[1], g[2]))
