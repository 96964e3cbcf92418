"""What the prefetching generator delivers on its own (no training step beside it): seconds per batch of 16 images at 608 x 608
from 64 PNG files, for the loader-process and the loader-thread mode, next to the in-process decode of one batch
(load_batch) - the three figures tests/test_gpu_host_loop.py::test_prefetching_generator_hides_the_host_path reasons with.
usage: python3 tools/loader_rate.py [epochs]"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench  # noqa: E402
from test_gpu_host_loop import _png_dataset_608  # noqa: E402
from multigriddet_amd.data.generators import MultiGridDataGenerator  # noqa: E402


def main():
    n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    tmp = tempfile.mkdtemp(prefix="mgd_loader_")
    lines = _png_dataset_608(tmp, 64) * 4
    S, B = 608, 16
    nw = min(16, os.cpu_count() or 8)
    anchors = bench.coco_anchors()

    def gen(prefetch, mode):
        return MultiGridDataGenerator(lines, B, (S, S), anchors, 80, augment=False, shuffle=True, seed=3, num_workers=nw,
                                      prefetch_factor=prefetch, host_augment=False, max_boxes_per_image=10, worker_mode=mode)
    g = gen(4, "process")
    g.load_batch(0)
    t0 = time.perf_counter()
    for i in range(8):
        g.load_batch(i, pinned=True)
    print(f"load_batch in process ({nw} threads): {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per batch")
    for mode in ("process", "thread"):
        g = gen(4, mode)
        for _ in g:                                   # start-up epoch
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cnt = 0
        for _ in range(n_ep):
            for (x, _) in g:
                cnt += 1
        torch.cuda.synchronize()
        print(f"prefetching generator alone, worker_mode={mode}: {(time.perf_counter() - t0) / cnt * 1e3:.2f} ms per batch ({cnt} batches)")
        g.close()


if __name__ == "__main__":
    main()
