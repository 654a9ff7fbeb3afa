import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The native libraries are build products (git-ignored).  If a checkout arrives without them,
# build before the first import instead of failing every test on ImportError.
if not (os.path.exists(os.path.join(ROOT, "sparsh_amg_amd", "libsparsh_amg.so"))
        and os.path.exists(os.path.join(ROOT, "oracle", "libamg_oracle.so"))):
    import __graft_entry__

    __graft_entry__.build()

REF_DIR = "/root/reference"  # exists only in the build container, never on the GPU box
C0_MATRIX = os.path.join(REF_DIR, "matrix_poisson_P1_14401")
C0_RHS = os.path.join(REF_DIR, "matrix_poisson_P1rhs_14401")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test (still part of the default CPU suite unless deselected)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "appendix_a.json")) as f:
        return json.load(f)


C0_FIXTURE = os.path.join(ROOT, "tests", "golden", "c0_matrix.npz")  # data fixture made by tests/golden/make_c0_fixture.py


@pytest.fixture(scope="session")
def have_c0():
    """The reference tree itself (build container only); the C0 *data* travels as C0_FIXTURE."""
    return os.path.exists(C0_MATRIX) and os.path.exists(C0_RHS)


def load_c0():
    """BASELINE.json configs[0]: (rowptr, colindex, val, b) of the reference's bundled matrix."""
    z = np.load(C0_FIXTURE)
    return (np.ascontiguousarray(z["rowptr"], dtype=np.int32), np.ascontiguousarray(z["colindex"], dtype=np.int32),
            np.ascontiguousarray(z["val"], dtype=np.float64), np.ascontiguousarray(z["b"], dtype=np.float64))


@pytest.fixture(scope="session")
def c0_arrays():
    return load_c0()


def rel_close(a, b, rtol):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return np.all(np.abs(a - b) <= rtol * np.abs(b))


def hist_tolerance(ref_hist):
    """Per-iteration tolerance of SURVEY.md §8(d): 1e-6 relative while r_k >= 1e-6 r_0,
    1e-3 relative below that."""
    ref_hist = np.asarray(ref_hist, dtype=float)
    r0 = ref_hist[0]
    return np.where(ref_hist >= 1e-6 * r0, 1e-6, 1e-3)
