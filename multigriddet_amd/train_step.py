"""One training step of the hot path: targets -> forward -> MultiGridLoss fwd/bwd -> backward ->
(gradient all-reduce over RCCL, overlapped with backward) -> optimiser -> re-pack bf16 weights.

Replaces what `model.fit` does per batch in the reference (multigriddet/trainers/trainer.py:572-581 with
the train model of models/multigriddet_darknet.py:551-751 and the tf.data target hook
data/generators.py:2112).  The reference is single-device; data parallelism here is new design:
one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI), gradients averaged.
"""
import os

import torch

from .streams import shared_stream
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
        self.comm_stream = shared_stream("comm", dev) if world_size > 1 else None
        cabi = None
        if world_size > 1 and os.environ.get("MGD_DP_COMM", "torch") == "cabi":
            import torch.distributed as dist
            from .dp import CabiComm
            cabi = CabiComm(dist.get_rank(), world_size, dev)       # the library's own RCCL binding (mgd_comm_*)
        self.dp = GradBuckets(net.grads, [cv.off_w for cv in net.layers], world_size, bucket_mb, self.comm_stream,
                              producer_streams=(net.wg_stream,), cabi_comm=cabi)
        # N > 1: the optimiser and the weight re-pack of a bucket ride on the communication stream right behind that
        # bucket's all-reduce, under the rest of backward - no serial optimiser tail after the last collective
        self.bucket_optimizer = True
        self._bucket_packs = {}
        self.main_stream = shared_stream("main", dev, priority=-1)
        # hipGraph replay of the whole step (single process, Adam/AdamW): see enable_graph()
        self.use_graph = False
        self.early_optimizer = True     # see _step_body
        self.targets_on_side_stream = True   # build_targets on the weight-gradient stream, under the forward pass
        self._graphs = {}
        self._hyper = torch.zeros(2, dtype=torch.float32, device=dev)
        # launch plan (enable_plan): the step's ~600 C-ABI calls recorded once, replayed by mgd_plan_run without the interpreter
        # (on by default for single-process Adam / AdamW runs; MGD_PLAN=0 or enable_plan(False): every launch from Python)
        self.use_plan = os.environ.get("MGD_PLAN", "1") == "1"
        self._plans = {}
        self._targets = {}
        self._ytrue = {}

    # ------------------------------------------------------------------ helpers
    def _grids(self, H, W):
        return [(H // 32, W // 32), (H // 16, W // 16), (H // 8, W // 8)]

    def _loss_runner(self, B, H, W):
        key = (B, H, W)
        if key not in self._loss:
            cfg = ops.make_loss_cfg(self.anchors, self.num_classes, (H, W), B, self._grids(H, W), **self.loss_kwargs)
            self._loss[key] = ops.LossRunner(cfg, self.net.device, class_weights=self.class_weights)
            F = 5 + len(self.anchors[0]) + self.num_classes
            self._douts[key] = [torch.empty(B, g[0], g[1], F, dtype=self.net.act_dtype, device=self.net.device)
                                for g in self._grids(H, W)]
        return self._loss[key], self._douts[key]

    # ------------------------------------------------------------------ the step
    def enable_graph(self, on=True):
        """Replay the step from a captured hipGraph (torch.cuda.CUDAGraph over both streams) instead of launching
        its ~450 kernels from Python: the backward pass is otherwise launch-bound on the host (0.85 ms of gaps on
        the compute stream).  Single-process Adam/AdamW only; the first two steps of a new input shape run
        eagerly (they size arenas and set kernel attributes), the third is captured.  Adam's bias-corrected step
        size lives in device memory (mgd_adam_step_dev) and is refreshed before every replay."""
        self.use_graph = bool(on)

    def enable_plan(self, on=True):
        """Replay the step from a recorded LAUNCH PLAN (csrc/plan.cpp): the third step of a new input shape runs eagerly while
        every call into the library is recorded - entry point, arguments, stream, cross-stream waits; from then on one call,
        mgd_plan_run, issues the same launches on the same two streams from C with the interpreter lock released (a step costs
        Python ~6 ms of enqueueing otherwise - time the loader's threads of the same process cannot use).  Not a hipGraph: the
        streams remain two queues.  Single process, Adam / AdamW (the step size lives in device memory, as under enable_graph);
        the batch's images / boxes may be new tensors every step (their addresses are parameters of the plan)."""
        self.use_plan = bool(on)
        self._plans = {}
        return self

    def _plannable(self, images, boxes, y_true):
        if not (self.use_plan and self.world == 1 and self.optimizer in ("adam", "adamw") and not self.net.fp32):
            return False
        if ops.PROFILE is not None or (boxes is None) == (y_true is None):      # (per-launch event brackets need the eager path)
            return False
        return images.is_contiguous() and images.dtype == torch.float32 and (boxes is None or (boxes.is_contiguous() and boxes.dtype == torch.float32))

    def _step_plan(self, images, boxes, y_true=None):
        from . import _lib as L
        net = self.net
        key = (tuple(images.shape), tuple(boxes.shape) if boxes is not None else "y_true", net.freeze_backbone, net.freeze_all_but_pred,
               net.freeze_bn)
        st = self._plans.get(key)
        if st is None:
            st = self._plans[key] = {"eager": 0, "plan": None}
        if st["plan"] is False or (st["plan"] is None and st["eager"] < 2):
            st["eager"] += 1                              # allocations (arenas, workspaces, kernel attributes) happen here
            return self._step_eager(images, boxes, y_true)
        if y_true is not None:
            # ready targets (the generator's device part built them): they arrive in new tensors every batch, and the loss takes
            # them through a pointer table - the plan reads them from buffers of its own (three device copies, 43 MB at 608 x 608)
            yk = tuple(tuple(y.shape) for y in y_true)
            buf = self._ytrue.get(yk)
            if buf is None:
                buf = self._ytrue[yk] = [torch.empty_like(y) for y in y_true]
            for d, y in zip(buf, y_true):
                d.copy_(y)
            y_true = buf
        self.step_count += 1
        lr_t, lr_wd = self._adam_hyper()
        self._hyper[0:1].fill_(lr_t)
        self._hyper[1:2].fill_(lr_wd)
        cur = torch.cuda.current_stream()
        self.main_stream.wait_stream(cur)
        params = [images] if boxes is None else [images, boxes]
        if st["plan"] is None:
            rec = L.Recorder([self.main_stream, net.wg_stream], params=params)
            L.RECORDER = rec
            try:
                with torch.cuda.stream(self.main_stream):
                    st["comp"] = self._step_body(images, boxes, y_true, dev_hyper=True)
            finally:
                L.RECORDER = None
            if rec.error is not None:                     # something in this configuration cannot be replayed: stay eager
                import warnings
                warnings.warn(f"launch plan not used: {rec.error}")
                st["plan"] = False
            else:
                st["plan"] = rec.finish()
        else:
            st["plan"].run(params)
        cur.wait_stream(self.main_stream)
        return st["comp"]

    def _graphable(self, boxes):
        return self.use_graph and self.world == 1 and boxes is not None and self.optimizer in ("adam", "adamw")

    def _adam_hyper(self):
        b1, b2 = self.opt_kwargs.get("beta_1", 0.9), self.opt_kwargs.get("beta_2", 0.999)
        t = self.step_count
        lr_t = self.lr * (1.0 - b2 ** t) ** 0.5 / (1.0 - b1 ** t)
        wd = self.opt_kwargs.get("weight_decay", 5e-4) if self.optimizer == "adamw" else 0.0
        return lr_t, self.lr * wd

    def _step_graph(self, images, boxes):
        net = self.net
        key = (tuple(images.shape), tuple(boxes.shape), net.freeze_backbone, net.freeze_all_but_pred, net.freeze_bn)
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"eager": 0, "graph": None}
        if st["graph"] is None and st["eager"] < 2:
            st["eager"] += 1
            return self._step_eager(images, boxes, None)
        self.step_count += 1
        lr_t, lr_wd = self._adam_hyper()
        self._hyper[0:1].fill_(lr_t)
        self._hyper[1:2].fill_(lr_wd)
        if st["graph"] is None:
            st["images"], st["boxes"] = images.clone(), boxes.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["comp"] = self._step_body(st["images"], st["boxes"], None, dev_hyper=True)
            st["graph"] = g
        if images.data_ptr() != st["images"].data_ptr():
            st["images"].copy_(images)
        if boxes.data_ptr() != st["boxes"].data_ptr():
            st["boxes"].copy_(boxes)
        st["graph"].replay()
        return st["comp"]

    def step(self, images, boxes=None, y_true=None):
        """images fp32 CUDA [B,H,W,3] in [0,1]; boxes fp32 CUDA [B,M,5] (x1,y1,x2,y2,cls) or ready y_true.
        Returns the device tensor of 8 loss components (index 7 = total)."""
        if self._plannable(images, boxes, y_true):
            return self._step_plan(images, boxes, y_true)
        if y_true is None and self._graphable(boxes):
            return self._step_graph(images, boxes)
        return self._step_eager(images, boxes, y_true)

    def _step_eager(self, images, boxes, y_true):
        """The step runs on an internal HIGH-priority stream: the dgrad/BatchNorm chain is the critical path of the
        backward pass and the weight gradients of the (default-priority) side stream should only fill what it leaves
        (measured: 15.35 -> 15.22 ms/step).  The caller's stream waits for it on the way out."""
        self.step_count += 1
        cur = torch.cuda.current_stream()
        self.main_stream.wait_stream(cur)
        with torch.cuda.stream(self.main_stream):
            comp = self._step_body(images, boxes, y_true, dev_hyper=False)
        cur.wait_stream(self.main_stream)
        return comp

    def _step_body(self, images, boxes, y_true, dev_hyper):
        net = self.net
        B, H, W, _ = images.shape
        if y_true is None:
            # the targets of a shape always land in the same buffers (a recorded plan holds their addresses)
            tkey = (tuple(boxes.shape), H, W)
            tout = self._targets.get(tkey)
            if tout is None:
                ys = ops.build_targets(boxes, (H, W), self.anchors, self.num_classes, self._grids(H, W), mode=self.target_mode)
                need = ops.L.load().mgd_build_targets_workspace_size(B, boxes.shape[1], len(ys),
                                                                     (ops.C.c_int32 * (2 * len(ys)))(*[int(v) for g in self._grids(H, W) for v in g]))
                tout = self._targets[tkey] = (ys, torch.empty(need, dtype=torch.uint8, device=boxes.device))
            bxs = boxes if boxes.is_contiguous() else boxes.contiguous()
        else:
            bxs = None
        # Work the forward pass does not depend on goes to the weight-gradient stream, which is idle until the first data
        # gradient exists: the targets (needed by the loss only) and the zeroing of the 250-MB gradient buffer (its last
        # reader, the optimiser of the previous step, is behind us on the main stream).  The main stream joins before the loss.
        side = net.wg_stream if (self.targets_on_side_stream and net.overlap_wgrad) else None
        if side is not None:
            ops.stream_wait(side)
            with torch.cuda.stream(side):
                if bxs is not None:
                    y_true = ops.build_targets(bxs, (H, W), self.anchors, self.num_classes, self._grids(H, W), mode=self.target_mode, out=tout)
                net.zero_grad()
        elif bxs is not None:
            y_true = ops.build_targets(bxs, (H, W), self.anchors, self.num_classes, self._grids(H, W), mode=self.target_mode, out=tout)
        outs = net.forward(images)
        if side is not None:
            ops.stream_wait(torch.cuda.current_stream(), side)
        else:
            net.zero_grad()
        runner, douts = self._loss_runner(B, H, W)
        comp = runner.run(y_true, outs, **({"grad_f32": douts} if net.fp32 else {"grad_bf16": douts}))
        if self.world > 1:
            rng_ = net.trainable_range()
            lo = rng_[0] if rng_ else 0
            self.dp.reset(lo=lo)
            # fp32 (strict-parity) networks keep no packed bf16 images: their optimiser runs on the tail (apply_optimizer)
            per_bucket = self.bucket_optimizer and rng_ is not None and not net.fp32
            self.dp.after_bucket = (lambda k, b, e: self._bucket_update(k, b, e, dev_hyper)) if per_bucket else None
            net.backward(douts, on_layer_done=self.dp.on_layer_done)
            self.dp.finish()      # makes the compute stream wait for the collectives (and the per-bucket updates)
            if per_bucket:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            else:
                self.apply_optimizer(dev_hyper=dev_hyper)
        elif self._early_ok():
            # Adam + re-pack of the layers whose gradients are final early (stage 5 + head: two thirds of the parameters,
            # then stage 4: another quarter) go to the weight-gradient side stream in the middle of backward - HBM-bound
            # work under the MFMA-bound data-gradient chain - instead of sitting on the serial tail of the step
            from .engine import EARLY_SPLITS
            offs = [net.layers[l].off_w for l in EARLY_SPLITS]
            ends = [net.n_params] + offs[:-1]

            def hook(ld):
                if ld not in EARLY_SPLITS:
                    return
                k = EARLY_SPLITS.index(ld)
                ops.stream_wait(net.wg_stream)        # BN-affine gradients of these layers come from the main stream
                with torch.cuda.stream(net.wg_stream):
                    self._optimizer_range(offs[k], ends[k], dev_hyper)
                    net._pack_seg[k].run()
            net.backward(douts, on_layer_done=hook)   # ends with the main stream joined to the side stream
            self._optimizer_range(0, offs[-1], dev_hyper)
            net._pack_seg[-1].run()
        else:
            net.backward(douts)
            self.apply_optimizer(dev_hyper=dev_hyper)
        return comp

    def _bucket_update(self, k, b, e, dev_hyper):
        """Runs on the communication stream behind bucket k's all-reduce: optimiser over the summed slice [b, e) and the
        bf16 re-pack of the convs whose weights lie in it."""
        net = self.net
        self._optimizer_range(b, e, dev_hyper)
        pk = self._bucket_packs.get((k, b))
        if pk is None:
            pairs = [(cv.pk, cv.wpack) for cv in net.layers if cv.pk is not None and b <= cv.off_w < e]
            pk = self._bucket_packs[(k, b)] = ops.PackBatch(pairs, net.device) if pairs else False
        if pk:
            pk.run()

    def _early_ok(self):
        net = self.net
        return (self.early_optimizer and self.world == 1 and net.overlap_wgrad and not net.freeze_backbone
                and not net.freeze_all_but_pred)

    def _optimizer_range(self, b, e, dev_hyper=False):
        """One optimiser launch over the flat slice [b, e) on the current stream."""
        net = self.net
        gs = 1.0 / self.world
        p, g, m = net.params[b:e], net.grads[b:e], self.m[b:e]
        if self.optimizer in ("adam", "adamw"):
            b1, b2 = self.opt_kwargs.get("beta_1", 0.9), self.opt_kwargs.get("beta_2", 0.999)
            eps = self.opt_kwargs.get("epsilon", 1e-7)
            if dev_hyper:
                ops.adam_step_dev(p, g, m, self.v[b:e], self._hyper, b1=b1, b2=b2, eps=eps, grad_scale=gs)
            else:
                ops.adam_step(p, g, m, self.v[b:e], self.lr, self.step_count, b1=b1, b2=b2, eps=eps, grad_scale=gs,
                              weight_decay=self.opt_kwargs.get("weight_decay", 5e-4) if self.optimizer == "adamw" else 0.0)
        else:
            ops.sgd_step(p, g, m, self.lr, momentum=self.opt_kwargs.get("momentum", 0.937),
                         nesterov=self.opt_kwargs.get("nesterov", True), grad_scale=gs)

    def apply_optimizer(self, dev_hyper=False):
        """step_count has been advanced by the caller."""
        net = self.net
        gs = 1.0 / self.world
        if net.freeze_all_but_pred:
            ranges = [(cv.off_w, cv.end) for cv in net.layers if cv.role == "pred"]
        else:
            ranges = [net.trainable_range()]
        for b, e in ranges:
            self._optimizer_range(b, e, dev_hyper)
        first = 0
        if net.freeze_backbone:
            first = 52
        net.refresh_packed(first)
