import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def coco_anchors():
    import numpy as np
    return [
        np.array([[112, 74], [149, 190], [370, 328]], dtype=np.float32),
        np.array([[28, 17], [56, 112], [57, 35]], dtype=np.float32),
        np.array([[9, 10], [13, 28], [28, 55]], dtype=np.float32),
    ]
