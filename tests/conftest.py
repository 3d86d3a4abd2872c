import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "quattro-transformer-ilqr_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Golden fixtures are plain arrays: loaded without pickle."""
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)


def per_step_rel(a, b, axis_t=0):
    """max over the time axis of the per-step relative Frobenius error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    a = np.moveaxis(a, axis_t, 0).reshape(a.shape[axis_t], -1)
    b = np.moveaxis(b, axis_t, 0).reshape(b.shape[axis_t], -1)
    den = np.linalg.norm(b, axis=1)
    den = np.where(den > 0, den, 1.0)
    return float(np.max(np.linalg.norm(a - b, axis=1) / den))


def per_step_rel_floor(a, b, floor_frac=0.01, axis_t=0):
    """per-step relative error with the denominator floored at floor_frac * (largest step norm): a step where the
    reference value passes through zero (scalar k of the cart-pole) does not turn round-off into a 'relative' error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    a = np.moveaxis(a, axis_t, 0).reshape(a.shape[axis_t], -1)
    b = np.moveaxis(b, axis_t, 0).reshape(b.shape[axis_t], -1)
    den = np.linalg.norm(b, axis=1)
    den = np.maximum(den, floor_frac * max(float(den.max()), 1e-300))
    return float(np.max(np.linalg.norm(a - b, axis=1) / den))
