"""kernel_trace.csv -> per-step duration and gaps of the last steps (a step ends with the adam kernel)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
ends = ends[-int(sys.argv[2]):] if len(sys.argv) > 2 else ends[-40:]
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows[a + 1:b + 1]
    t0 = int(rows[a]["End_Timestamp"])
    first_gap = (int(seg[0]["Start_Timestamp"]) - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - t0) / 1e3
    gaps = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3 for i in range(len(seg) - 1)]
    big = [(round(g, 1), seg[i]["Kernel_Name"][:30], seg[i + 1]["Kernel_Name"][:30]) for i, g in enumerate(gaps) if g > 8]
    print(f"step span {span:7.1f} us  kernels {len(seg):3d} busy {busy:7.1f}  gap-before-first {first_gap:6.1f}  max inner gap {max(gaps) if gaps else 0:6.1f}  {big[:3]}")

spans = [(int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 for a, b in zip(ends[:-1], ends[1:])]
spans_sorted = sorted(spans)
print("steps", len(spans), "median", spans_sorted[len(spans) // 2], "mean", sum(spans) / len(spans), "max", spans_sorted[-5:])
print("outliers (index, span):", [(i, round(x, 1)) for i, x in enumerate(spans) if x > 1.15 * spans_sorted[len(spans) // 2]][:40])
