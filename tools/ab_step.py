"""Interleaved A/B of engine switches on the benchmark train step (one process, same device, median of rounds).
usage: python tools/ab_step.py attr=v1,v2 [attr2=...]   e.g.  fuse_bn_reduce=0,1 overlap_wgrad=0,1 ts.use_graph=0,1
(plain names are attributes of the Network, ts.* of the TrainStep)"""
import itertools, os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigriddet_amd.engine import Network
from multigriddet_amd.train_step import TrainStep

specs = [a.split("=") for a in sys.argv[1:]]
names = [s[0] for s in specs]
vals = [[int(v) for v in s[1].split(",")] for s in specs]
dev = torch.device("cuda:0")
net = Network(80, 3, dev, seed=0)
ts = TrainStep(net, bench.coco_anchors(), 80, (608, 608), 16, lr=1e-4)
img, bx = bench.synth_batch(0, 16, 608)
img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
combos = list(itertools.product(*vals))
res = {c: [] for c in combos}
for rnd in range(4):
    for c in combos:
        for n, v in zip(names, c):
            if n.startswith("ts."):
                setattr(ts, n[3:], bool(v))
            else:
                setattr(net, n, bool(v))
        for _ in range(2):
            ts.step(img, bx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            ts.step(img, bx)
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / 8 * 1e3)
for c in combos:
    print(dict(zip(names, c)), "ms/step median %.3f min %.3f" % (statistics.median(res[c]), min(res[c])))
