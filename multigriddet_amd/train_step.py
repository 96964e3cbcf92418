"""One training step of the hot path: targets -> forward -> MultiGridLoss fwd/bwd -> backward ->
(gradient all-reduce over RCCL, overlapped with backward) -> optimiser -> re-pack bf16 weights.

Replaces what `model.fit` does per batch in the reference (multigriddet/trainers/trainer.py:572-581 with
the train model of models/multigriddet_darknet.py:551-751 and the tf.data target hook
data/generators.py:2112).  The reference is single-device; data parallelism here is new design:
one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI), gradients averaged.
"""
import torch
import torch.distributed as dist

from . import ops
from .dp import GradBuckets


class TrainStep:
    def __init__(self, net, anchors, num_classes, input_shape, batch, loss_kwargs=None, lr=1e-4,
                 optimizer="adam", opt_kwargs=None, world_size=1, bucket_mb=32.0, target_mode=0,
                 class_weights=None):
        self.net = net
        self.anchors = anchors
        self.num_classes = num_classes
        self.input_shape = tuple(input_shape)
        self.batch = batch
        self.loss_kwargs = dict(loss_kwargs or {})
        self.lr = lr
        self.optimizer = optimizer
        self.opt_kwargs = dict(opt_kwargs or {})
        self.world = world_size
        self.target_mode = target_mode
        self.class_weights = class_weights
        self.step_count = 0
        dev = net.device
        self.m = torch.zeros_like(net.params)
        self.v = torch.zeros_like(net.params) if optimizer in ("adam", "adamw") else None
        self._loss = {}
        self._douts = {}
        self.comm_stream = torch.cuda.Stream(device=dev) if world_size > 1 else None
        self.dp = GradBuckets(net.grads, [cv.off_w for cv in net.layers], world_size, bucket_mb, self.comm_stream,
                              producer_streams=(net.wg_stream,))

    # ------------------------------------------------------------------ helpers
    def _grids(self, H, W):
        return [(H // 32, W // 32), (H // 16, W // 16), (H // 8, W // 8)]

    def _loss_runner(self, B, H, W):
        key = (B, H, W)
        if key not in self._loss:
            cfg = ops.make_loss_cfg(self.anchors, self.num_classes, (H, W), B, self._grids(H, W), **self.loss_kwargs)
            self._loss[key] = ops.LossRunner(cfg, self.net.device, class_weights=self.class_weights)
            F = 5 + len(self.anchors[0]) + self.num_classes
            self._douts[key] = [torch.empty(B, g[0], g[1], F, dtype=torch.bfloat16, device=self.net.device)
                                for g in self._grids(H, W)]
        return self._loss[key], self._douts[key]

    # ------------------------------------------------------------------ the step
    def step(self, images, boxes=None, y_true=None):
        """images fp32 CUDA [B,H,W,3] in [0,1]; boxes fp32 CUDA [B,M,5] (x1,y1,x2,y2,cls) or ready y_true.
        Returns the device tensor of 8 loss components (index 7 = total)."""
        net = self.net
        B, H, W, _ = images.shape
        if y_true is None:
            y_true = ops.build_targets(boxes, (H, W), self.anchors, self.num_classes, self._grids(H, W),
                                       mode=self.target_mode)
        outs = net.forward(images)
        net.zero_grad()
        runner, douts = self._loss_runner(B, H, W)
        comp = runner.run(y_true, outs, grad_bf16=douts)
        if self.world > 1:
            rng_ = net.trainable_range()
            self.dp.reset(lo=rng_[0] if rng_ else 0)
            net.backward(douts, on_layer_done=self.dp.on_layer_done)
            self.dp.finish()      # makes the compute stream wait for the collectives
        else:
            net.backward(douts)
        self.apply_optimizer()
        return comp

    def apply_optimizer(self):
        net = self.net
        self.step_count += 1
        gs = 1.0 / self.world
        if net.freeze_all_but_pred:
            ranges = [(cv.off_w, cv.end) for cv in net.layers if cv.role == "pred"]
        else:
            ranges = [net.trainable_range()]
        for b, e in ranges:
            p, g, m = net.params[b:e], net.grads[b:e], self.m[b:e]
            if self.optimizer in ("adam", "adamw"):
                ops.adam_step(p, g, m, self.v[b:e], self.lr, self.step_count,
                              b1=self.opt_kwargs.get("beta_1", 0.9), b2=self.opt_kwargs.get("beta_2", 0.999),
                              eps=self.opt_kwargs.get("epsilon", 1e-7), grad_scale=gs,
                              weight_decay=self.opt_kwargs.get("weight_decay", 5e-4) if self.optimizer == "adamw" else 0.0)
            else:
                ops.sgd_step(p, g, m, self.lr, momentum=self.opt_kwargs.get("momentum", 0.937),
                             nesterov=self.opt_kwargs.get("nesterov", True), grad_scale=gs)
        first = 0
        if net.freeze_backbone:
            first = 52
        net.refresh_packed(first)
