cd /tmp && export TMPDIR=/tmp
for m in "" "--morton"; do
  for w in "--workload c3" "--workload c4"; do
    rm -rf /tmp/pm
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-workloads --min-seconds 0 --steps 30 --warmup 5 $w $m > /tmp/pm.log 2>&1
    echo "== $w $m"; tail -1 /tmp/pm.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
    python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/pm 30 12 | cut -c1-60,70-
  done
done
