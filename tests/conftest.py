"""Shared fixtures.  `-m "not gpu"` runs on the CPU-only build box; `-m gpu` runs on an MI355X
and calls the HIP path through the C ABI (never a fallback)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle .so files are build products; make them if a fresh checkout lacks them."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_r8.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)


@pytest.fixture(scope="session")
def oracles():
    from oracle.pyoracle import Oracle
    return {4: Oracle(4), 8: Oracle(8)}


@pytest.fixture(scope="session")
def hipctx():
    """One HIP context for the whole GPU session; fails loudly without the library/device."""
    from seabreeze_param_amd import hip
    ctx = hip.Context()
    yield ctx
    ctx.close()


def gpu_visible():
    """Is there an AMD GPU on this machine?  Asked of the kernel driver, not of torch: once another HIP runtime in the
    process (libseabreeze_hip.so links ROCm's, torch ships its own) has initialised the device, torch.cuda.is_available()
    can answer False on a GPU box."""
    return os.path.exists("/dev/kfd")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def relerr(a, b, floor=1e-12):
    """max |a-b| / max(|b|, floor) ignoring positions where both are NaN."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a - b) / np.maximum(np.abs(b), floor)
    d[both_nan] = 0.0
    return float(np.max(d)) if d.size else 0.0
