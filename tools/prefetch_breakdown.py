"""Where a train step fed by the prefetching generator spends its time: per step the host time inside the generator's
__next__, inside TrainStep.step (enqueue), and the wall time per step, for the synthetic batch, the loader-process and the
loader-thread mode (same set-up as tests/test_gpu_host_loop.py::test_prefetching_generator_hides_the_host_path).
usage: python3 tools/prefetch_breakdown.py"""
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
from multigriddet_amd.engine import Network  # noqa: E402
from multigriddet_amd.train_step import TrainStep  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="mgd_loader_")
    lines = _png_dataset_608(tmp, 64) * 4
    S, B = 608, 16
    nw = min(16, os.cpu_count() or 8)
    anchors = bench.coco_anchors()
    dev = torch.device("cuda:0")
    net = Network(80, 3, dev, seed=0)
    ts = TrainStep(net, anchors, 80, (S, S), B, lr=1e-4).enable_plan(True)
    img, bx = bench.synth_batch(0, B, S)
    img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
    for _ in range(4):
        ts.step(img, bx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        ts.step(img, bx)
    torch.cuda.synchronize()
    print(f"synthetic batch: {(time.perf_counter() - t0) / 16 * 1e3:.2f} ms per step")

    def run(g, n_ep, what):
        for (x, _) in g:                                   # start-up epoch
            ts.step(x[0], y_true=list(x[1:]))
        torch.cuda.synchronize()
        t_next = t_step = 0.0
        cnt = 0
        t0 = time.perf_counter()
        for _ in range(n_ep):
            it = iter(g)
            while True:
                a = time.perf_counter()
                try:
                    (x, _) = next(it)
                except StopIteration:
                    break
                b = time.perf_counter()
                ts.step(x[0], y_true=list(x[1:]))
                c = time.perf_counter()
                t_next += b - a
                t_step += c - b
                cnt += 1
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / cnt
        print(f"{what}: {tot * 1e3:.2f} ms per step; host inside next() {t_next / cnt * 1e3:.2f} ms, inside step() {t_step / cnt * 1e3:.2f} ms")

    for mode in ("process", "thread"):
        g = MultiGridDataGenerator(lines, B, (S, S), anchors, 80, augment=False, shuffle=True, seed=3, num_workers=nw,
                                   prefetch_factor=4, host_augment=False, max_boxes_per_image=10, worker_mode=mode)
        run(g, 3, f"prefetching generator, worker_mode={mode}")
        g.close()
    # the same device work without any loader: y_true given, batches resident
    g = MultiGridDataGenerator(lines, B, (S, S), anchors, 80, augment=False, shuffle=True, seed=3, num_workers=nw,
                               prefetch_factor=0, host_augment=False, max_boxes_per_image=10)
    (x, _) = next(iter(g))
    xs = [t.clone() for t in x]
    for _ in range(4):
        ts.step(xs[0], y_true=list(xs[1:]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        ts.step(xs[0], y_true=list(xs[1:]))
    torch.cuda.synchronize()
    print(f"resident batch with y_true given (no loader): {(time.perf_counter() - t0) / 16 * 1e3:.2f} ms per step")


if __name__ == "__main__":
    main()
