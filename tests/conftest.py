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


def _host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


try:        # torch-CPU oracle: never oversubscribe a shared box (os.cpu_count() reports the whole host)
    import torch
    torch.set_num_threads(min(_host_cores(), 16))
except Exception:
    pass
