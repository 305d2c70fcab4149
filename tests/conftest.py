import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """The parity margins of this session (tests/helpers.py: MARGINS) -> gpurun_out/parity_margins.json, when GPU parity
    tests ran.  One record per (test, tensor, bound): the worst |err| / bound observed."""
    try:
        import helpers
    except Exception:
        return
    if not helpers.MARGINS or not any("test_gpu_parity" in k[0] or "test_dist_gpu" in k[0] for k in helpers.MARGINS):
        return
    import json
    recs = [{"test": k[0], "tensor": k[1], "rtol": k[2], "ulps": k[3], "worst_over_bound": round(v["worst"], 4), "calls": v["calls"],
             "fp32_oracle_over_bound": None if v["k32"] is None else round(v["k32"], 4), "calibration": round(v["calibration"], 3),
             "ulps_needed": None if v["ulps_needed"] is None else round(v["ulps_needed"], 3)}
            for k, v in sorted(helpers.MARGINS.items(), key=lambda kv: -kv[1]["worst"])]
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_margins.json"), "w") as f:
        json.dump({"feature_rtol": helpers.FEATURE_RTOL, "grad_rtol": helpers.GRAD_RTOL, "feature_ulps": helpers.FEATURE_ULPS,
                   "grad_ulps": helpers.GRAD_ULPS, "records": recs}, f, indent=0)
