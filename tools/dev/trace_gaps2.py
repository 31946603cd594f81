"""developer: from a rocprofv3 kernel trace of bench.py (tools/kstat.sh) -- the idle gap between consecutive step graphs on the main queue
(end of adam_guarded -> start of the next step's first kernel), the step period, and how long the main-queue kernels take when a
build-stream kernel overlaps them.   python tools/dev/trace_gaps2.py gpurun_out/kstat_<tag>/k_kernel_trace.csv"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [r for r in rows if "adam_guarded" in r["Kernel_Name"]]
mainq = adam[len(adam) // 2]["Queue_Id"]
main = [r for r in rows if r["Queue_Id"] == mainq]
gaps, periods = [], []
for a, b in zip(main, main[1:]):
    if "adam_guarded" in a["Kernel_Name"]:
        gaps.append((b["s"] - a["e"]) / 1e3)
ad = [r for r in main if "adam_guarded" in r["Kernel_Name"]]
periods = [(b["e"] - a["e"]) / 1e3 for a, b in zip(ad, ad[1:])]
inner = []
for a, b in zip(main, main[1:]):
    if "adam_guarded" not in a["Kernel_Name"]:
        inner.append((b["s"] - a["e"]) / 1e3)
mid = lambda v: statistics.median(v) if v else float("nan")
print(f"main queue {mainq}: {len(ad)} steps; gap adam -> next step's first kernel: median {mid(gaps):.1f} us (p10 {sorted(gaps)[len(gaps)//10]:.1f}, p90 {sorted(gaps)[9*len(gaps)//10]:.1f}); "
      f"gap between kernels inside a step: median {mid(inner):.2f} us; step period median {mid(periods):.1f} us")
side = [r for r in rows if r["Queue_Id"] != mainq and any(t in r["Kernel_Name"] for t in ("knn", "csr_"))]
if side:
    print(f"build queue: {len(side)} kernels, median durations:", {k: round(mid([(r['e'] - r['s']) / 1e3 for r in side if k in r['Kernel_Name']]), 1) for k in ("knn", "csr_scan", "csr_sort")})

if "--timeline" in sys.argv:
    # two consecutive steps in the middle of the run, both queues: start offset (us), duration (us), queue, kernel
    k = len(ad) // 2
    lo, hi = ad[k - 1]["e"], ad[k + 1]["e"]
    print(f"\ntimeline of two consecutive steps (queue {mainq} = the step graphs, the other queue = the graph builds running ahead):")
    for r in rows:
        if lo <= r["s"] <= hi:
            print(f"{(r['s'] - lo) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:7.1f}  q{r['Queue_Id']}  {r['Kernel_Name'].replace('void ', '').replace('eq::', '')[:80]}")
