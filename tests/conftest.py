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
