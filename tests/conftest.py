import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
LINEAR_SLOTS = (0, 3, 6, 9, 12, 15)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: torch.from_numpy(np.array(z[k])) for k in z.files}


def golden_params(scale=1.0):
    """Seed-0 default-init parameters of the reference; the 'x3' fixtures scale the
    six Linear weight matrices (tests/golden/make_golden.py: make_model)."""
    params = load_golden("params_seed0")
    if scale != 1.0:
        for slot in LINEAR_SLOTS:
            key = f"prediction_heads.{slot}.weight"
            params[key] = params[key] * scale
    return params


def stable_rays(last_density, eps=1e-5):
    """Mask of rays whose last-interval density is not within eps of the step at 0
    (that interval is 1e10 wide, so its opacity is a step
    function of the density's sign and a rounding-level sign flip moves RGB by O(0.1))."""
    return last_density.abs() > eps


@pytest.fixture(scope="session")
def golden():
    return load_golden
