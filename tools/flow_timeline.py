"""Timeline of one mvs_flow call from a rocprofv3 --kernel-trace csv: per kernel start / end relative to the call's first kernel, the
queue it ran on, the gap to the previous kernel of the same queue.  usage: python3 tools/flow_timeline.py <kernel_trace.csv> [call index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by u8_to_f32_kernel pairs: split at every first u8_to_f32 after a pack_flow4
calls, cur = [], []
for r in rows:
    if "u8_to_f32_kernel" in r["Kernel_Name"] and cur and "pack_flow4" in cur[-1]["Kernel_Name"]:
        calls.append(cur); cur = []
    cur.append(r)
calls.append(cur)
idx = int(sys.argv[2]) if len(sys.argv) > 2 else len(calls) // 2
c = calls[idx]
t0 = int(c[0]["Start_Timestamp"])
last_end = {}
busy = 0
for r in c:
    s, e, q = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", "?")
    gap = s - last_end.get(q, s)
    last_end[q] = e
    busy += e - s
    name = r["Kernel_Name"].split("(")[0].replace("void mvs::", "").replace("mvs::", "")[:40]
    print("%9.2f %9.2f  dur %6.2f  gap %6.2f  q%-3s %s  grid %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, gap / 1e3, q, name, r.get("Grid_Size", "")))
print("kernels %d, span %.1f us, sum of durations %.1f us" % (len(c), (max(int(r["End_Timestamp"]) for r in c) - t0) / 1e3, busy / 1e3))
