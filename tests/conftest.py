import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- parity record: what the GPU parity tests measured, left behind as JSON (copied into profiles/ per round) ----------------
_PARITY = []


class ParityRecorder:
    """``rec(fixture, quantity, hip, ref64, ref32)`` stores max|hip - ref64|, max|hip - ref32| and the reference's own
    fp32-vs-fp64 gap for one compared tensor and returns (err64, gap)."""

    def __call__(self, fixture, quantity, hip, ref64, ref32=None, note=None):
        import numpy as np
        to_np = lambda v: v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)  # noqa: E731
        hip, ref64 = to_np(hip).astype(np.float64), to_np(ref64).astype(np.float64)
        entry = {"fixture": fixture, "quantity": quantity, "max_abs_ref64": float(np.abs(ref64).max()) if ref64.size else 0.0,
                 "err_vs_ref64": float(np.abs(hip - ref64).max()) if ref64.size else 0.0}
        gap = None
        if ref32 is not None:
            ref32 = to_np(ref32).astype(np.float64)
            gap = float(np.abs(ref32 - ref64).max()) if ref64.size else 0.0
            entry["err_vs_ref32"] = float(np.abs(hip - ref32).max()) if ref64.size else 0.0
            entry["reference_fp32_fp64_gap"] = gap
        if note:
            entry["note"] = note
        _PARITY.append(entry)
        return entry["err_vs_ref64"], gap


@pytest.fixture(scope="session")
def parity():
    return ParityRecorder()


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
            json.dump({"what": "max abs differences measured by tests/ (-m gpu) on the MI355X: HIP path vs the reference's fp64 and fp32 "
                               "runs (golden fixtures from the imported reference, or the oracle on seeded inputs), and the reference's own "
                               "fp32-fp64 gap on the same tensor",
                       "north_star": {"R_t_abs": 1e-4, "logits_abs": 1e-5}, "entries": _PARITY}, f, indent=1)
    except OSError:
        pass
