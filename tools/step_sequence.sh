# Launch sequence (start offset us, duration us, kernel) of the last full training step of a short bench run under rocprofv3.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-workloads --min-seconds 0 --steps 3 --warmup 2 > /tmp/seq.log 2>&1
f=$(find /tmp/seq -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of adam_kernel marks the end of a step; print the last full step
ends = [i for i, n in enumerate(names) if "adam_" in n]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:8.1f} {((e-s)/1000):6.1f}  {r['Kernel_Name'][:90]}")
PY
