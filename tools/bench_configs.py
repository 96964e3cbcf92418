"""The BASELINE.json configs that are not bench.py's headline line, on one GPU:
  config 1: Darknet53 416x416, single image, forward + decode + DIoU-NMS (latency, ms)
  config 3: multi-scale training, S cycled over {320, ..., 608} per step at batch 16 (images/s over two full cycles)
  config 5 (single-GPU part): Mosaic + MixUp + GridMask on the device at 608x608, batch 16 (ms per batch)
Prints one JSON line per config.  Usage: python tools/bench_configs.py"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigriddet_amd.engine import Network
from multigriddet_amd.train_step import TrainStep
from multigriddet_amd.data import augment as aug

dev = torch.device("cuda:0")
B = 16

# ---- config 3
net = Network(80, 3, dev, seed=0)
ts = TrainStep(net, bench.coco_anchors(), 80, (320, 320), B, lr=1e-4,
               loss_kwargs=dict(loss_option=2, loss_normalization=["batch"]))
sizes = list(range(320, 609, 32))
batches = {}
for s in sizes:
    img, bx = bench.synth_batch(s, B, s)
    batches[s] = (torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev))
for _ in range(2):
    for s in sizes:
        ts.step(*batches[s])
torch.cuda.synchronize()
t0 = time.perf_counter()
cycles = 2
for _ in range(cycles):
    for s in sizes:
        ts.step(*batches[s])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"config": "multi-scale {320..608 step 32} B=16 train, one TrainStep, arenas per resolution",
                  "images_per_sec": round(cycles * len(sizes) * B / dt, 1), "ms_per_step_avg": round(1e3 * dt / (cycles * len(sizes)), 3)}))
del ts, net, batches
torch.cuda.empty_cache()

# ---- config 1
from multigriddet_amd.models import build_multigriddet_darknet
from multigriddet_amd.postprocess import MultiGridDecoder
anchors = bench.coco_anchors()
model, _ = build_multigriddet_darknet((416, 416, 3), [len(a) for a in anchors], 80)
model.net.training = False
dec = MultiGridDecoder(anchors, 80, (416, 416))
x = torch.rand(1, 416, 416, 3, device=dev)
shapes = np.array([[416, 416]], np.int32)
for fold in (False, True):
    model.fold_bn(fold)
    for _ in range(10):
        heads = model(x, training=False)
        dec.postprocess_batch(heads, shapes, confidence=0.008, nms_method="diou")
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        heads = model(x, training=False)
        out = dec.postprocess_batch(heads, shapes, confidence=0.008, nms_method="diou")
    torch.cuda.synchronize()
    print(json.dumps({"config": "Darknet53 416x416 single image forward + decode + DIoU-NMS", "fold_bn": fold,
                      "latency_ms": round(1e3 * (time.perf_counter() - t0) / n, 3)}))
model.fold_bn(False)

# ---- config 5 (device augmentation)
rng = np.random.default_rng(0)
img = torch.from_numpy((rng.random((B, 608, 608, 3), dtype=np.float32) * 255).astype(np.float32)).to(dev)
bx = np.zeros((B, 100, 5), np.float32)
for b in range(B):
    for t in range(8):
        w, h = rng.uniform(6, 300, 2)
        cx, cy = rng.uniform(w / 2, 608 - w / 2), rng.uniform(h / 2, 608 - h / 2)
        bx[b, t] = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.integers(0, 80)]
bxd = torch.from_numpy(bx).to(dev)
def aug_once():
    i2, b2 = aug.mosaic(img, bxd, *aug.draw_mosaic(rng, B, 608))
    i2, b2 = aug.mixup(i2, b2, *aug.draw_mixup(rng, B))
    apply, par = aug.draw_gridmask(rng, B, 608, 1.0)
    aug.gridmask(i2, b2, apply, par)
for _ in range(3):
    aug_once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    aug_once()
torch.cuda.synchronize()
print(json.dumps({"config": "Mosaic + MixUp + GridMask on the device, 608x608, batch 16", "ms_per_batch": round(1e3 * (time.perf_counter() - t0) / 20, 3)}))
