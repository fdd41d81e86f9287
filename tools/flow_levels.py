"""per-level average duration of the Farneback iteration kernels from a rocprofv3 kernel trace (csv)"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "farneback_iteration" in n:
        agg[(n.split("(")[0][-32:], int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), "launches, average %.1f us" % (sum(v) / len(v) / 1e3))
