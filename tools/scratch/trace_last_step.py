"""The kernels of the LAST training step in a rocprofv3 --kernel-trace CSV (from the last k_adam group backwards to the one before), in launch order, with durations:
names shortened, consecutive repeats folded.  usage: trace_last_step.py <kernel_trace.csv> [marker kernel substring, default k_huber|k_lt_loss]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "k_adam"
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
# steps end with a burst of k_adam launches: split into bursts
bursts = []
for i in idx:
    if bursts and i - bursts[-1][-1] <= 3: bursts[-1].append(i)
    else: bursts.append([i])
if len(bursts) < 2: sys.exit("fewer than two steps in the trace")
a, b = bursts[-2][-1] + 1, bursts[-1][-1] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
print("last step: %d launches, span %.2f ms, busy %.2f ms" % (len(step), (t1 - t0) / 1e6, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6))
out = []
for r in step:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nrf::", "")[:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    if out and out[-1][0] == n: out[-1][1] += d; out[-1][2] += 1
    else: out.append([n, d, 1, g])
for n, d, c, g in out:
    print("  %9.1f us  x%-3d grid %-10d %s" % (d, c, g, n))
