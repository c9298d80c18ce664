import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy access to the fixtures captured from the reference (tests/golden/capture_golden.py)."""

    def __init__(self):
        self._files = {}

    def __call__(self, group: str, key: str) -> np.ndarray:
        if group not in self._files:
            self._files[group] = np.load(os.path.join(GOLDEN, f"{group}.npz"))
        return self._files[group][key]

    def keys(self, group: str):
        self(group, next(iter(np.load(os.path.join(GOLDEN, f"{group}.npz")).files)))
        return list(self._files[group].files)


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def meta():
    import json
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)
