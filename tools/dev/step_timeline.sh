# dev: raw kernel timeline of one replayed step (gaps between kernels included)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $ROOT/bench.py --no-cpu-baseline --no-workloads --no-fixed-pair --min-seconds 0 --steps 60 --warmup 10 "$@" > /tmp/tl.log 2>&1
python3 $ROOT/tools/dev/step_timeline.py "$(find /tmp/tl -name '*kernel_trace.csv' | head -1)"
