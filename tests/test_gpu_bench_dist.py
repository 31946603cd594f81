"""bench.py's N>1 path, exercised on ONE GPU as child processes:

* ``BENCH_FORCE_DIST=1``: the real EGNN + pose head re-homed into the joint bucket, an RCCL process group (1 rank), the step as
  two hipGraphs with the eager all-reduce between them, and as ONE graph with the collective captured (``--graph-collectives``);
  the gradient of one step must be bit-identical to the non-distributed path (a 1-rank mean all-reduce is the identity);
* ``--gpus N`` without a launcher environment starts N ranks itself -- on a 1-GPU box it must refuse loudly, not measure 1 GPU;
* ``--global-batch`` shards with ``dp.shard_range``.
Every run is a fresh child process (bench.py initialises the GPU and a process group of its own)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SMALL = ["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--min-seconds", "0", "--no-workloads", "--n", "512", "--k", "12",
         "--layers", "2"]


def _bench(extra, env_extra=None, expect_rc=0):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == expect_rc, f"rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    if expect_rc != 0:
        return None, p
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return json.loads(lines[-1]), p          # the JSON line is the LAST line of stdout (RCCL's banner is flushed before it)


@pytest.fixture(scope="module")
def plain():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    line, _ = _bench([], {"BENCH_GRAD_HASH": "1"})
    assert "multi_gpu" not in line and line["n_gpus"] == 1 and line["config"]["hipgraph"] is True
    return line


@pytest.fixture(scope="module")
def plain_fixed():
    line, _ = _bench(["--fixed-pair-replay"], {"BENCH_GRAD_HASH": "1"})
    assert "multi_gpu" not in line and line["config"]["hipgraph"] is True and line["config"]["graph_mode"] == "one graph"
    return line


@pytest.mark.parametrize("flags,mode", [([], "one graph (all-reduce captured)"), (["--split-graph"], "two graphs + eager all-reduce")])
def test_forced_dist_step_on_rccl_matches_the_plain_step(plain, flags, mode):
    """The package's loop (train.GraphedStep: all-reduce captured inside the step graph by default, --split-graph = two graphs around
    an eager all-reduce) and the harness-captured replay of rounds 1-3 (--fixed-pair-replay), each on a 1-rank RCCL group."""
    line, _ = _bench(flags, {"BENCH_FORCE_DIST": "1", "BENCH_GRAD_HASH": "1", "MASTER_PORT": "29611"})
    mg = line["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["world_size"] == 1 and mg["distinct_devices"] == 1 and "ranks_seen" not in mg
    assert mg["allreduce_bytes"] > 100_000 and mg["allreduce_us_isolated"] > 0
    assert line["config"]["graph_mode"] == mode and line["config"]["hipgraph"] is True
    assert line["config"]["params_finite"] is True and line["value"] > 0
    assert line["grad_hash"] == plain["grad_hash"], "the RCCL path changed the step's gradients"
    assert plain["grad_hash"]["abs_sum"] > 0


def test_gpus_flag_starts_ranks_itself_and_refuses_missing_devices():
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1)] + SMALL, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 2 and "exposes" in p.stderr and not p.stdout.strip()
    if n >= 2:                      # a multi-GPU box: the self-launched job must really span two devices
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-4000:]
        line = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
        assert line["n_gpus"] == 2 and line["multi_gpu"]["distinct_devices"] == 2


def test_global_batch_is_sharded_with_shard_range():
    line, _ = _bench(["--global-batch", "3"])
    assert line["scaling"] == "strong" and line["config"]["pairs_per_gpu"] == 3 and line["config"]["global_batch"] == 3
    assert abs(line["value"] - 3 * 1e3 / line["ms_per_step"]) < 1e-2 * line["value"]


def test_multi_rank_default_line_carries_the_scaling_workloads():
    """What the driver's `bench.py --gpus N` run must bring back, checked on the forced-dist path with one
    rank: 16 pairs per GPU (weak), the KITTI batch of 64 sharded over the ranks (strong), the eval sweep pair -- each with its
    roofline, the training ones with the isolated all-reduce time -- and the compact copy inside `roofline`."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_FORCE_DIST="1", MASTER_PORT="29617")
    import tempfile
    rec = os.path.join(tempfile.mkdtemp(), "full.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--min-seconds", "0.2", "--workload-seconds", "0.1",
                        "--steps", "50", "--warmup", "5", "--full-record", rec], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=1200)
    assert p.returncode == 0, p.stderr[-4000:]
    last = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]
    assert len(last) < 6144, len(last)                      # the driver's record must be able to hold the line (BENCH_r05: 32.7 KB -> parsed null)
    line = json.loads(last)
    assert line["config"]["graph_mode"] == "one graph (all-reduce captured)" and line["multi_gpu"]["allreduce_us_isolated"] > 0
    assert line["full_record"] == rec and line["workload_names"] == ["c3_b16", "c4_global64", "c5_eval"]
    full = json.load(open(rec))
    assert len(full["multi_gpu"]["ranks_seen"]) == 1
    w = full["workloads"]
    assert set(w) == {"c3_b16", "c4_global64", "c5_eval"}, w.keys()
    assert w["c3_b16"]["scaling"] == "weak" and w["c3_b16"]["config"]["pairs_per_gpu"] == 16 and w["c3_b16"]["allreduce_us_isolated"] > 0
    assert w["c4_global64"]["scaling"] == "strong" and w["c4_global64"]["config"]["global_batch"] == 64 and w["c4_global64"]["allreduce_us_isolated"] > 0
    assert w["c5_eval"]["config"]["points"] == 8192 and w["c5_eval"]["value"] > 0
    for name, e in w.items():
        assert e["roofline"]["frac"] > 0 and e["config"]["hipgraph"] is True, name
        assert line["roofline"][f"{name}_frac"] == e["roofline"]["frac"] and line["roofline"][f"{name}_value"] == e["value"]


def test_default_line_is_compact_and_complete():
    """The driver's command form (`bench.py --gpus 1 --steps K --warmup W`, here with a short K and no CPU leg): the LAST stdout line is
    the compact record -- contract scalars, `config`, `roofline` with the headline kernel's scalars, `measured_GBs` and one flat scalar
    set per extra workload -- under 6 KB; the full record is a file."""
    import tempfile
    rec = os.path.join(tempfile.mkdtemp(), "full.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                        "--min-seconds", "0.1", "--workload-seconds", "0.05", "--full-record", rec], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-4000:]
    last = [ln for ln in p.stdout.splitlines() if ln.strip()][-1]
    assert len(last) < 6144, len(last)
    line = json.loads(last)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in line, key
    assert line["steps"] == 5 and line["warmup"] == 2 and line["n_gpus"] == 1 and line["value"] > 0
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert all(not isinstance(v, (dict, list)) for v in r.values()), "roofline members must be scalars (the driver drops nested ones)"
    full = json.load(open(rec))
    for name, e in full["workloads"].items():
        assert "error" not in e, (name, e)
        assert r[f"{name}_value"] == e["value"] and r[f"{name}_frac"] == e["roofline"]["frac"]
    if r["traffic"] is not None:
        assert abs(r["measured_GBs"] - r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9) < 1.0


@pytest.mark.parametrize("flags,mode", [(["--fixed-pair-replay"], "two graphs + eager all-reduce"),
                                        (["--fixed-pair-replay", "--graph-collectives"], "one graph")])
def test_forced_dist_harness_replay_matches_its_plain_form(plain_fixed, flags, mode):
    """The harness-captured replay of rounds 1-3 (--fixed-pair-replay, a diagnostic since round 4) on the 1-rank RCCL group."""
    line, _ = _bench(flags, {"BENCH_FORCE_DIST": "1", "BENCH_GRAD_HASH": "1", "MASTER_PORT": "29613"})
    assert line["config"]["graph_mode"] == mode and line["config"]["hipgraph"] is True and line["multi_gpu"]["world_size"] == 1
    assert line["grad_hash"] == plain_fixed["grad_hash"], "the RCCL path changed the step's gradients"
