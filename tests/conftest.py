import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "tests", "data")
TEST_JSON = os.path.join(DATA, "test.json")
SCENE_407 = os.path.join(DATA, "tfrecord-00002-of-01000_407.json")
SCENE_4 = os.path.join(DATA, "tfrecord-00000-of-01000_4.json")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def scenes():
    return dict(test=TEST_JSON, s407=SCENE_407, s4=SCENE_4)


@pytest.fixture(autouse=True, scope="session")
def _rank_audit_is_checked_whenever_a_simulator_is_closed():
    """Every index the rank kernels derive from another kernel's data is range-checked on the device (gd_stat 21 counts the
    violations and the index is clamped): a broken invariant must fail a test, not show up as a quietly wrong row.  Every GPU
    simulator a test closes is asked for the counter first."""
    try:
        from gpudrive_lab_amd import madrona_gpudrive_impl as M
    except Exception:  # (collection on a box without the library: the CPU suite never builds a simulator)
        yield
        return
    plain_close = M.SimManager.close

    def checked_close(self):
        if getattr(self, "_h", None):
            bad = self.stat(21)
            plain_close(self)
            assert bad == 0, "rank-path bounds audit: %d violations (gd_stat 21)" % bad
        else:
            plain_close(self)
    M.SimManager.close = checked_close
    yield
    M.SimManager.close = plain_close
