#!/usr/bin/env python3
"""bench.py - images/sec of one MultiGridDet train step (fwd + MultiGridLoss + bwd + Adam) at 608x608,
batch 16 per GPU, on N MI355X of one node (BASELINE.json `metric`).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Inputs (synthetic COCO-shaped images + boxes) are resident in HBM before
the timed region.  `roofline` is for the dominant kernel - whichever of the large-tile bf16 MFMA convolution kernel
families (the library reports the family of every launch: mgd_last_kernel) takes the largest share of the step: algorithmic
conv FLOPs of its launches / their summed durations, measured with HIP events on the launch stream inside the timed region;
the others are listed under `other_kernels`.  `roofline.per_layer`: every distinct 3x3 convolution of the backbone at the
benchmark shape, forward / data gradient / weight gradient each launched alone (north_star's per-layer MFMA figure, from
this run).  `infer`: forward + decode + DIoU-NMS at batch 16 and 1, each with its own MFMA roofline.
`cpu_baseline` times the oracle (torch-CPU restatement of the same train step) on the host cores, rank 0,
N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIZE = 608
BATCH = 16
FLOP_PER_IMAGE_FWD = 112.111328256e9          # 69 convs at 608^2
PEAK_BF16_TFLOPS = 2500.0                     # dense bf16 MFMA, MI355X_MICROARCH.md


def coco_anchors():
    return [np.array([[112, 74], [149, 190], [370, 328]], np.float32),
            np.array([[28, 17], [56, 112], [57, 35]], np.float32),
            np.array([[9, 10], [13, 28], [28, 55]], np.float32)]


def synth_batch(rank, batch, size, max_boxes=100):
    """images default_rng(0) (+rank), boxes n~U{1..20}, w,h=exp(U(ln8, ln400))."""
    rng = np.random.default_rng(0 + rank)
    images = rng.random((batch, size, size, 3), dtype=np.float32)
    rng = np.random.default_rng(1 + rank)
    boxes = np.zeros((batch, max_boxes, 5), np.float32)
    for b in range(batch):
        for t in range(int(rng.integers(1, 21))):
            w = min(float(np.exp(rng.uniform(np.log(8), np.log(400)))), size - 2)
            h = min(float(np.exp(rng.uniform(np.log(8), np.log(400)))), size - 2)
            cx, cy = rng.uniform(w / 2, size - w / 2), rng.uniform(h / 2, size - h / 2)
            boxes[b, t] = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.integers(0, 80)]
    return images, boxes


def host_cores():
    """Cores this process may really use: min(affinity, cgroup cpu quota).  os.cpu_count() on a
    shared box reports every core of the host and oversubscribing them makes torch-CPU crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(size, batch):
    """Oracle train step (torch-CPU fp32 conv/BN/autograd + restated loss/targets + Adam) on the host:
    bounded sample = ONE step of the benchmark batch at the benchmark resolution (10-20 s of host work), after a tiny
    warm-up step."""
    from oracle import model as om
    from oracle.loss import MultiGridLossOracle
    from oracle import targets as ot
    cores = host_cores()
    torch.set_num_threads(cores)
    params = om.torch_params(om.init_params(0), requires_grad=True)
    state = {}

    def one(sz, batch):
        images, boxes = synth_batch(0, batch, sz)
        lo = MultiGridLossOracle(coco_anchors(), 80, (sz, sz))
        yt = ot.tf_preprocess_true_boxes(boxes, (sz, sz), coco_anchors(), 80)
        outs = om.forward(torch.from_numpy(images), params, training=True)
        loss = lo([torch.from_numpy(y) for y in yt], outs)
        loss.backward()
        om.adam_step(params, state, lr=1e-4)
        return float(loss.detach())

    print("[bench] cpu_baseline warm-up ...", file=sys.stderr, flush=True)
    one(128, 1)
    print("[bench] cpu_baseline timed step ...", file=sys.stderr, flush=True)
    t0 = time.time()
    one(size, batch)
    dt = time.time() - t0
    return {"value": round(batch / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"1 train step (targets+fwd+loss+bwd+Adam) of the torch-CPU oracle at {size}x{size}, "
                      f"batch {batch} (the benchmark batch), {cores} threads, after a 128x128 warm-up step; {dt:.2f} s"}


def infer_bench(dev, size, steps=20, warmup=3):
    """Second half of the BASELINE metric ("infer FPS+NMS"): forward on moving BatchNorm statistics + batched decode +
    DIoU-NMS + top-100 + xyxy (reference inference_engine.py:98-140 -> postprocess/multigrid_decode.py:237-345), synthetic
    608x608 batches resident in HBM, random-init weights.  Outside the train-step timed region; one sub-record per
    (batch, fold_bn).  confidence 0.008 so that the random-init heads keep ~100 detections per image and NMS does real
    work (at the reference's 0.1 nothing survives the filter on random weights)."""
    from multigriddet_amd.models import build_multigriddet_darknet
    from multigriddet_amd.postprocess import MultiGridDecoder
    model, _ = build_multigriddet_darknet(input_shape=(size, size, 3), num_classes=80)
    dec = MultiGridDecoder(coco_anchors(), 80, (size, size))
    kw = dict(max_boxes=100, confidence=0.008, nms_threshold=0.45, nms_method="diou")
    runs = []
    for batch, folds in ((16, (False, True)), (1, (False, True)), (32, (True,))):       # batch 32: what a batched service would run
        img = torch.from_numpy(np.random.default_rng(0).random((batch, size, size, 3), dtype=np.float32)).to(dev)
        shapes = [(480, 640)] * batch
        for fold in folds:
            model.fold_bn(fold)
            for _ in range(warmup):
                r = dec.postprocess_batch(model(img, training=False), shapes, **kw)
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            t0 = time.perf_counter()
            for _ in range(steps):
                e0.record()
                outs = model(img, training=False)
                e1.record()
                r = dec.postprocess_batch(outs, shapes, **kw)
                e2.record()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            fwd_ms = e0.elapsed_time(e1)
            ach = batch * FLOP_PER_IMAGE_FWD * (size / 608) ** 2 / (fwd_ms * 1e-3) / 1e12
            runs.append({"batch": batch, "fold_bn": fold, "images_per_sec": round(batch * steps / dt, 1),
                         "ms_per_batch": round(1e3 * dt / steps, 3), "forward_ms": round(fwd_ms, 3),
                         "decode_nms_ms": round(e1.elapsed_time(e2), 3), "detections_last_batch": int(r[3].sum()),
                         "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                                      "note": "conv FLOPs of the forward pass (112.11 GFLOP per 608x608 image) / its HIP-event time, last batch"}})
    model.fold_bn(False)
    return {"metric": "inference images/sec incl. decode + DIoU-NMS", "size": size, "steps": steps, "warmup": warmup,
            "nms": "diou", "confidence": 0.008, "max_boxes": 100, "dtype": "bf16",
            "data": "synthetic, random-init weights, moving BN statistics", "runs": runs}


def per_layer_bench(dev, size, batch, iters=5):
    """north_star's "fraction of the conv-MFMA roofline on backbone 3x3 convs", from the driver's own run: every distinct
    3x3 convolution of the backbone at the benchmark shape, forward / data gradient / weight gradient each launched ALONE
    (HIP events around `iters` launches, one warm-up), as TFLOP/s and as a fraction of the dense bf16 peak.  The weight gradient
    gets the engine's slab workspace and, being alone, one block per CU (inside the step it is capped at ops.WGRAD_ROW_BLOCKS)."""
    from multigriddet_amd import ops
    rows = []
    h = size
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)      # slab workspace of the kernel-row weight gradient
    for f in (64, 128, 256, 512, 1024):
        for (ci, co, s, hin, count) in ((f // 2, f, 2, h, 1), (f // 2, f, 1, h // 2, {64: 1, 128: 2, 256: 8, 512: 8, 1024: 4}[f])):
            ho = hin // s
            x = torch.randn(batch, hin, hin, ci, device=dev).to(torch.bfloat16)
            dy = torch.randn(batch, ho, ho, co, device=dev).to(torch.bfloat16)
            w = torch.randn(co, 9, ci, device=dev) * 0.05
            pk = ops.PackedConv(co, ci, 3, s, dev)
            pk.refresh(w)
            y = torch.empty(batch, ho, ho, co, dtype=torch.bfloat16, device=dev)
            dx = torch.empty(batch, hin, hin, ci, dtype=torch.bfloat16, device=dev)
            dw = torch.zeros(co, 9, ci, device=dev)
            st = torch.zeros(ops.STATS_REPLICAS, 2, co, device=dev)
            fl = 2.0 * batch * ho * ho * 9 * ci * co

            def t(fn):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e-3 / iters
            tf = [fl / t(fn) / 1e12 for fn in (lambda: ops.conv_fwd(x, pk, out=y, stats=st),
                                               lambda: ops.conv_dgrad(dy, pk, (hin, hin), out=dx),
                                               lambda: ops.conv_wgrad(x, dy, dw, 3, s, ws=ws, row_blocks=0))]
            rows.append({"layer": f"{ci}->{co} 3x3 s{s} @{hin}", "count": count, "gflop": round(fl / 1e9, 2),
                         "fwd_tflops": round(tf[0], 1), "dgrad_tflops": round(tf[1], 1), "wgrad_tflops": round(tf[2], 1),
                         "fwd_frac": round(tf[0] / PEAK_BF16_TFLOPS, 3), "dgrad_frac": round(tf[1] / PEAK_BF16_TFLOPS, 3),
                         "wgrad_frac": round(tf[2] / PEAK_BF16_TFLOPS, 3)})
        h //= 2
    return rows


def _profile_rows(kind, prefix):
    """(file, [(launches, value)]) of every kernel instantiation whose name starts with `prefix` in the newest committed summary
    profiles/rNN_<kind>.txt; the value is the last (pmc_traffic: corrected MB per launch) or the MFMA-utilisation column."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for rnd in ("r04", "r03", "r02", "r01"):                        # newest committed summary first
        name = f"{rnd}_{kind}.txt"
        rows = []
        try:
            for line in open(os.path.join(root, name)):
                if line.startswith(prefix):
                    c = [x.strip() for x in line.split("|")]
                    rows.append((int(c[1]), float(c[-1] if kind == "pmc_traffic" else c[4])))
        except (OSError, ValueError, IndexError):
            continue
        if rows:
            return f"profiles/{name}", rows
    return None, []


def pmc_traffic(prefix):
    """HBM bytes per launch of a kernel FAMILY (every instantiation whose name starts with `prefix`, weighted by the launches
    sampled - the same launch set as the HIP-event figures) from the committed PMC summary (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of this command; FETCH_SIZE x2 on gfx950); (None, reason) if absent."""
    src, rows = _profile_rows("pmc_traffic", prefix)
    if not rows:
        return None, "no PMC summary found"
    n = sum(r[0] for r in rows)
    return int(sum(r[0] * r[1] for r in rows) / n * 1024 * 1024), f"{src} ({len(rows)} instantiation(s), {n} launches sampled; separate --pmc passes)"


def pmc_mfma_busy(prefix):
    """SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles) of the same family, launches-weighted (own --pmc pass: solo)."""
    src, rows = _profile_rows("mfma_util", prefix)
    if not rows:
        return None, "no MFMA-busy summary found"
    n = sum(r[0] for r in rows)
    return round(sum(r[0] * r[1] for r in rows) / n, 4), f"{src} ({n} launches sampled, counter pass serialises the kernels)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=SIZE)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # rehearsal on a one-GPU box: MGD_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo for the exchange
    # (RCCL refuses two ranks on one device); the data-parallel code path is otherwise the one the 8-GPU run takes
    share = os.environ.get("MGD_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from multigriddet_amd import ops
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep

    net = Network(80, 3, dev, seed=0)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(net.params, 0)
        net.refresh_packed()
    ts = TrainStep(net, coco_anchors(), 80, (args.size, args.size), args.batch, lr=1e-4, world_size=world,
                   loss_kwargs=dict(loss_option=2, loss_normalization=["batch"]))
    images, boxes = synth_batch(rank, args.batch, args.size)
    images, boxes = torch.from_numpy(images).to(dev), torch.from_numpy(boxes).to(dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        comp = ts.step(images, boxes)
    barrier()
    # HIP events bracket every launch of the dominant kernel during the LAST timed step only (event records
    # between back-to-back kernels cost ~4 ms per step when applied to every step).
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1 and not args.no_kernel_events:
            ops.PROFILE = []
        comp = ts.step(images, boxes)
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.PROFILE
    ops.PROFILE = None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss = float(comp[7])

    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    ips = world * args.batch * args.steps / dt
    roof = None
    if prof:
        # kernel families as the library names them (mgd_last_kernel): description + the kernel-name prefix of the PMC summary
        fam = {
            "conv_gather_gemm(global weight fragments)": ("conv_gemm8_kernel<2,3> (128x128-tile bf16 MFMA gather-GEMM, weight fragments from "
                                                          "global memory, pixel tile through an LDS-DMA ring, three blocks per CU)", "conv_gemm8_kernel<"),
            "conv_gather_gemm(producer/consumer)": ("conv_gemm6_kernel<2,2,4,4,4> (128x128-tile gather-GEMM, producer/consumer waves)", "conv_gemm6_kernel<"),
            "conv_gather_gemm(counted pipeline)": ("conv_gemm9_kernel<4,NT,NST> (gather-GEMM, hand-counted asm memory pipeline, two blocks per CU)", "conv_gemm9_kernel<4,"),
            "conv_gather_gemm(counted pipeline, ping-pong)": ("conv_gemm9_kernel<8,8,4,true> (256x128-tile gather-GEMM, 8 waves in ping-pong phases)", "conv_gemm9_kernel<8,"),
            "conv_gather_gemm(phased, 8 waves)": ("conv_gemm12_kernel (256x192-tile gather-GEMM, 8 waves, both operands by LDS-DMA, counted vmcnt across barriers)", "conv_gemm12_kernel<"),
            "conv_wgrad(descriptor-addressed)": ("conv_wgrad4_kernel (128x64-tile bf16 MFMA weight gradient, per-tap blocks, split-K, "
                                                 "descriptor-addressed operands; all its instantiations: 3x3 and 1x1 launches)", "conv_wgrad4_kernel<"),
            "conv_wgrad": ("conv_wgrad2_kernel (per-tap weight gradient, carried addresses: stride-2 layers)", "conv_wgrad2_kernel<"),
        }
        step_ms = dt * 1e3 / args.steps

        def summarise(tag, only=None):
            sel = [p for p in prof if p[3] == tag and (only is None or p[4] == only)]
            fl = sum(p[2] for p in sel)
            ms = sum(p[0].elapsed_time(p[1]) for p in sel)
            n = len(sel)
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            by = sum(p[5] for p in sel)
            return {"achieved": round(ach, 2), "frac": round(ach / PEAK_BF16_TFLOPS, 4), "launches": n,
                    "avg_launch_us": round(1e3 * ms / max(n, 1), 2), "share_of_step_time": round(ms / step_ms, 3),
                    "algorithmic_flop_per_launch_avg": round(fl / max(n, 1) / 1e9, 2),
                    "algorithmic_bytes_per_launch_avg": int(by / max(n, 1))}
        tags = sorted({p[3] for p in prof})
        per = {t: summarise(t) for t in tags}
        big = [t for t in tags if t in fam]                              # the large-tile MFMA kernels
        dom = max(big, key=lambda t: per[t]["share_of_step_time"])       # the kernel the step spends most time in
        traffic, tsrc = pmc_traffic(fam[dom][1])                         # the whole family: the launch set of frac / avg_launch_us
        busy, bsrc = pmc_mfma_busy(fam[dom][1])
        alg = per[dom]["algorithmic_bytes_per_launch_avg"]
        roof = {"bound": "mfma", "kernel": fam[dom][0], "achieved": per[dom]["achieved"], "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": per[dom]["frac"], "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_source": tsrc, "algorithmic_bytes_per_launch": alg,
                "traffic_over_algorithmic": round(traffic / alg, 2) if traffic and alg else None,
                "mfma_busy": busy, "mfma_busy_source": bsrc,
                "launches": per[dom]["launches"], "avg_launch_us": per[dom]["avg_launch_us"],
                "event_steps": 1, "share_of_step_time": per[dom]["share_of_step_time"],
                "algorithmic_flop_per_launch_avg": per[dom]["algorithmic_flop_per_launch_avg"],
                "note": "HIP events on the launch stream around every conv launch of the last timed step; the backward pass "
                        "runs two streams, so backward launches share the CUs with kernels of the other stream; "
                        "forward_only = the same kernel family over its forward launches, which run alone",
                "other_kernels": {(fam[t][0] if t in fam else t).split(" ")[0]: per[t] for t in tags if t != dom},
                "forward_only": summarise("conv_gather_gemm(global weight fragments)", "conv_fwd"),
                "whole_step_conv_tflops": round(ips / world * 3 * FLOP_PER_IMAGE_FWD * (args.size / 608) ** 2 / 1e12, 1)}
        if world == 1:
            roof["per_layer"] = per_layer_bench(dev, args.size, args.batch)
    out = {
        "metric": "images/sec (train step, 608x608, bs/GPU=16)", "value": round(ips, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Darknet53+FPN+DenseYOLO head, {args.size}x{args.size} bs={args.batch}/GPU, "
                               f"train step (targets+fwd+MultiGridLoss+bwd+Adam), COCO-80 synthetic",
                   "global_batch": world * args.batch, "parallelism": f"dp{world}", "loss": round(loss, 4),
                   "train_tflops_per_gpu": round(ips / world * 3 * FLOP_PER_IMAGE_FWD * (args.size / 608) ** 2 / 1e12, 1)},
        "roofline": roof,
    }
    if world == 1 and not args.no_infer:
        del ts, net
        torch.cuda.empty_cache()
        out["infer"] = infer_bench(dev, args.size)
    else:
        out["infer"] = None
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.batch)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
