"""kernel_trace.csv -> the timeline of one late step (a step ends with the adam kernel): start offset, duration, gap before each kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
for which in (-3, -2):
    a, b = ends[which - 1], ends[which]
    seg = rows[a + 1:b + 1]
    t0 = int(rows[a]["End_Timestamp"])
    prev = t0
    print(f"--- step ending at dispatch {b}: span {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{(s - t0) / 1e3:7.1f}  dur {(e - s) / 1e3:6.1f}  gap {(s - prev) / 1e3:5.1f}  {r['Kernel_Name'].replace('void ', '').replace('eq::', '')[:44]}")
        prev = e
