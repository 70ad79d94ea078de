import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases(prefix):
    return sorted(os.path.basename(p)[len(prefix) + 1:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def load_golden(prefix, name):
    return np.load(os.path.join(GOLDEN, "%s_%s.npz" % (prefix, name)))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a HIP device")
    return torch.device("cuda:0")


# ---- observed parity errors: every GPU parity test records (name, observed max-abs, asserted bound); the table is written
# to gpurun_out/parity_observed.json at the end of the session (a copy of one run is committed under profiles/)
_OBSERVED = {}


def record_err(name, err, bound):
    """Assert err <= bound and remember the observed value."""
    _OBSERVED[str(name)] = {"max_abs": float(err), "bound": float(bound)}
    assert err <= bound, (name, err, bound)


def pytest_sessionfinish(session, exitstatus):
    if not _OBSERVED:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_observed.json"), "w") as f:
            json.dump(dict(sorted(_OBSERVED.items())), f, indent=1)
    except OSError:
        pass
