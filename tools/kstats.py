#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel calls, average µs, share, µs per step."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>9s} {'us/step':>9s} {'pct':>6s}")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e3/steps:9.1f} {float(r['Percentage']):6.2f}")
print(f"total GPU kernel time {tot/1e6:.2f} ms = {tot/1e3/steps:.1f} us/step over {len(rows)} kernels, {sum(int(r['Calls']) for r in rows)} launches ({sum(int(r['Calls']) for r in rows)/steps:.0f}/step)")
