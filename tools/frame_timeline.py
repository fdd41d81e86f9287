"""Timeline of ONE mvs_process_frame call from a rocprofv3 --kernel-trace csv: kernels in start order with queue, duration and the gap to the previous kernel
on the same queue; then the busy time per kernel name.  usage: python3 tools/frame_timeline.py <kernel_trace.csv> [call index]   (calls are split after compact_scatter)"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# a call ends with triangulatePixels' compaction: split after every compact_scatter
calls, cur = [], []
for r in rows:
    cur.append(r)
    if "compact_scatter" in r["Kernel_Name"]:
        calls.append(cur); cur = []
if cur:
    calls.append(cur)
c = calls[int(sys.argv[2]) if len(sys.argv) > 2 else len(calls) // 2]
t0 = int(c[0]["Start_Timestamp"])
last, busy = {}, collections.Counter()
for r in c:
    s, e, q = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", "?")
    name = r["Kernel_Name"].split("(")[0].replace("void mvs::", "").replace("mvs::", "")[:44]
    print("%9.2f %9.2f  dur %7.2f  gap %7.2f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, (s - last.get(q, s)) / 1e3, q, name))
    last[q] = e
    busy[name] += e - s
print("kernels %d, span %.1f us, sum of durations %.1f us" % (len(c), (max(int(r["End_Timestamp"]) for r in c) - t0) / 1e3, sum(busy.values()) / 1e3))
for name, ns in busy.most_common(14):
    print("   %-46s %8.1f us" % (name, ns / 1e3))
