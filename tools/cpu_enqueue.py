"""Host-side enqueue time of the train step's phases (no device sync inside): is the step launch-bound?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigriddet_amd import ops
from multigriddet_amd.engine import Network
from multigriddet_amd.train_step import TrainStep
dev = torch.device("cuda:0")
net = Network(80, 3, dev, seed=0)
ts = TrainStep(net, bench.coco_anchors(), 80, (608, 608), 16, lr=1e-4)
img, bx = bench.synth_batch(0, 16, 608)
img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
for _ in range(3):
    ts.step(img, bx)
torch.cuda.synchronize()
acc = {"targets": 0, "forward": 0, "loss": 0, "backward": 0, "opt": 0}
N = 5
for _ in range(N):
    torch.cuda.synchronize()
    t = time.perf_counter()
    y = ops.build_targets(bx, (608, 608), ts.anchors, 80, ts._grids(608, 608)); t1 = time.perf_counter(); acc["targets"] += t1 - t
    outs = net.forward(img); t2 = time.perf_counter(); acc["forward"] += t2 - t1
    net.zero_grad(); runner, douts = ts._loss_runner(16, 608, 608); comp = runner.run(y, outs, grad_bf16=douts); t3 = time.perf_counter(); acc["loss"] += t3 - t2
    net.backward(douts); t4 = time.perf_counter(); acc["backward"] += t4 - t3
    ts.step_count += 1; ts.apply_optimizer(); t5 = time.perf_counter(); acc["opt"] += t5 - t4
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms of host time per phase; total", round(sum(acc.values()) / N * 1e3, 3))

# the same step from a recorded launch plan (TrainStep.enable_plan): host time of ONE mgd_plan_run
ts.enable_plan(True)
for _ in range(4):
    ts.step(img, bx)
torch.cuda.synchronize()
tp = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    t = time.perf_counter()
    ts.step(img, bx)
    tp += time.perf_counter() - t
plan = [st["plan"] for st in ts._plans.values()][0]
print("launch plan:", plan.size if plan else plan, "recorded operations;", round(tp / N * 1e3, 3), "ms of host time per step (no device sync inside)")
