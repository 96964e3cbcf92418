"""Inference throughput incl. decode + NMS (BASELINE metric, second half): images/s of forward (moving BN statistics)
+ batched decode + DIoU-NMS on synthetic 608x608 batches resident in HBM.  One JSON line.
usage: python tools/bench_infer.py [--batch 16] [--size 608] [--steps 30] [--method diou|soft|cluster|wbf]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigriddet_amd.models import build_multigriddet_darknet
from multigriddet_amd.postprocess import MultiGridDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=608)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--method", default="diou")
ap.add_argument("--confidence", type=float, default=0.1)
ap.add_argument("--fold-bn", action="store_true", help="BatchNorm folded into the convs (one launch per conv block)")
ap.add_argument("--graph", action="store_true", help="replay the forward pass from a captured hipGraph")
args = ap.parse_args()
dev = torch.device("cuda:0")
model, _ = build_multigriddet_darknet(input_shape=(args.size, args.size, 3), num_classes=80)
if args.fold_bn:
    model.fold_bn(True)
if args.graph:
    model.enable_graph(True)
dec = MultiGridDecoder(bench.coco_anchors(), 80, (args.size, args.size))
img = torch.from_numpy(np.random.default_rng(0).random((args.batch, args.size, args.size, 3), dtype=np.float32)).to(dev)
shapes = [(480, 640)] * args.batch
kw = dict(max_boxes=100, confidence=args.confidence, nms_threshold=0.45, nms_method="diou" if args.method == "wbf" else args.method,
          use_wbf=args.method == "wbf")


def step():
    outs = model(img, training=False)
    return dec.postprocess_batch(outs, shapes, **kw)


for _ in range(args.warmup):
    r = step()
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
t0 = time.perf_counter()
fwd_ms = 0.0
for _ in range(args.steps):
    e0.record(); outs = model(img, training=False); e1.record()
    r = dec.postprocess_batch(outs, shapes, **kw); e2.record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"metric": "inference images/sec incl. decode+NMS", "value": round(args.batch * args.steps / dt, 1),
                  "unit": "images/sec", "batch": args.batch, "size": args.size, "nms": args.method,
                  "fold_bn": bool(args.fold_bn), "graph": bool(args.graph), "ms_per_batch": round(1e3 * dt / args.steps, 3), "last_forward_ms": round(e0.elapsed_time(e1), 3),
                  "last_decode_nms_ms": round(e1.elapsed_time(e2), 3), "detections_last_batch": int(r[3].sum()),
                  "dtype": "bf16", "data": "synthetic, random-init weights"}))
