"""Does running the main chain on a high-priority stream (weight gradients stay on a default-priority side stream) help?"""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigriddet_amd.engine import Network
from multigriddet_amd.train_step import TrainStep
dev = torch.device("cuda:0")
net = Network(80, 3, dev, seed=0)
ts = TrainStep(net, bench.coco_anchors(), 80, (608, 608), 16, lr=1e-4)
img, bx = bench.synth_batch(0, 16, 608)
img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
hi = torch.cuda.Stream(device=dev, priority=-1)
res = {"default": [], "main_high": []}
for rnd in range(4):
    for mode in res:
        ctx = torch.cuda.stream(hi) if mode == "main_high" else torch.cuda.stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        with ctx:
            for _ in range(2):
                ts.step(img, bx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                ts.step(img, bx)
            torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 8 * 1e3)
for m, v in res.items():
    print(m, "ms/step median %.3f min %.3f" % (statistics.median(v), min(v)))
