"""Static executor of the Darknet53 + FPN + DenseYOLO-head graph on libmgd_hip.so.

The reference builds this graph with Keras layers and lets TensorFlow differentiate it
(multigriddet/models/backbones/darknet.py:19-40, models/heads/multigrid_head.py:38-74, 275-313,
models/multigriddet_darknet.py:488-548).  Here the topology is fixed at construction, every
activation lives in a pre-allocated NHWC bf16 arena sized for the resolution in use, parameters /
gradients / optimiser state are three flat fp32 buffers (so the optimiser is one launch and the
data-parallel all-reduce works on contiguous slices), and forward/backward are explicit sequences of
C-ABI calls - no autograd tape, no tracing compiler.  torch is the allocator and stream provider.
"""
import math

import numpy as np
import torch

from .streams import shared_stream

from . import _lib as L
from . import ops
from . import ops32

STAGES = ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4))
BACKBONE_CONVS = 52
EARLY_SPLITS = (43, 26)   # first convs of backbone stages 5 and 4 (the stride-2 convs): 67 % + 26 % of the parameters


def conv_specs(num_classes=80, num_anchors=3):
    """Conv list in graph-construction order (same order as the reference's Keras layer creation)."""
    specs = []

    def c(cin, cout, k, s=1, bn=True, role=""):
        specs.append(dict(cin=cin, cout=cout, k=k, s=s, bn=bn, role=role))

    c(3, 32, 3, role="stem")
    ch = 32
    for f, n in STAGES:
        c(ch, f, 3, 2, role="down")
        for _ in range(n):
            c(f, f // 2, 1, role="res1")
            c(f // 2, f, 3, role="res2")
        ch = f
    out = num_anchors + num_classes + 5
    cin = 1024
    for n, mult, skip in ((256, 8, 512), (128, 4, 256), (64, 2, None)):
        c(cin, n, 1, role="h1")
        c(n, 2 * n, 3, role="h2")
        c(2 * n, n, 1, role="h3")
        c(n, mult * out, 3, role="h4")
        c(mult * out, out, 1, bn=False, role="pred")
        if skip is not None:
            c(n, n // 2, 1, role="lat")
            cin = n // 2 + skip
    return specs


class Conv:
    """One conv (+BN+LeakyReLU) with views into the flat parameter / gradient buffers."""
    pass


class Network:
    def __init__(self, num_classes=80, num_anchors=3, device="cuda:0", seed=0, precision="bf16"):
        """precision: "bf16" (bf16 storage, fp32 accumulation on the matrix cores - the fast path) or "fp32" (every
        activation, conv and BatchNorm in fp32 through the direct kernels of csrc/fp32ref.hip: the reference's default
        numeric type, for strict end-to-end parity runs; slow by design)."""
        L.require_gpu()
        L.load()
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        self.precision = precision
        self.fp32 = precision == "fp32"
        self.O = ops32 if self.fp32 else ops          # the op set forward / backward call
        self._lat_ws = {}                             # stream handle -> ops.LatencyWorkspace (latency_workspace())
        self._wg_ws = None                            # fp32 slabs of the kernel-row weight gradient (_wgrad_workspace())
        self.act_dtype = torch.float32 if self.fp32 else torch.bfloat16
        self.device = torch.device(device)
        self.num_classes, self.num_anchors = num_classes, num_anchors
        self.out_ch = num_anchors + num_classes + 5
        self.specs = conv_specs(num_classes, num_anchors)
        # ---- flat parameter layout
        off = 0
        self.layers = []
        for i, sp in enumerate(self.specs):
            cv = Conv()
            cv.idx, cv.cin, cv.cout, cv.k, cv.s, cv.bn, cv.role = i, sp["cin"], sp["cout"], sp["k"], sp["s"], sp["bn"], sp["role"]
            cv.T = cv.k * cv.k
            cv.off_w = off
            off += cv.cout * cv.T * cv.cin
            cv.off_a = off          # gamma | bias
            off += cv.cout
            if cv.bn:
                cv.off_b = off      # beta
                off += cv.cout
            cv.end = off
            self.layers.append(cv)
        self.n_params = off
        self.backbone_end = self.layers[BACKBONE_CONVS - 1].end
        dev = self.device
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        n_bn = sum(l.cout for l in self.layers if l.bn)
        self.moving = torch.zeros(2 * n_bn, dtype=torch.float32, device=dev)
        self.bnwork = torch.zeros(4 * n_bn, dtype=torch.float32, device=dev)          # scale, shift, mean, invstd
        R = self.O.STATS_REPLICAS
        self.stats_all = torch.zeros(n_bn * (2 * R + 2 * (R + 1)), dtype=torch.float32, device=dev)
        bo, so = 0, 0
        for cv in self.layers:
            C = cv.cout
            cv.w = self.params[cv.off_w:cv.off_w + C * cv.T * cv.cin].view(C, cv.T, cv.cin)
            cv.dw = self.grads[cv.off_w:cv.off_w + C * cv.T * cv.cin].view(C, cv.T, cv.cin)
            if cv.bn:
                cv.gamma, cv.beta = self.params[cv.off_a:cv.off_a + C], self.params[cv.off_b:cv.off_b + C]
                cv.dgamma, cv.dbeta = self.grads[cv.off_a:cv.off_a + C], self.grads[cv.off_b:cv.off_b + C]
                cv.mm, cv.mv = self.moving[bo:bo + C], self.moving[n_bn + bo:n_bn + bo + C]
                cv.scale, cv.shift = self.bnwork[bo:bo + C], self.bnwork[n_bn + bo:n_bn + bo + C]
                cv.smean, cv.sinv = self.bnwork[2 * n_bn + bo:2 * n_bn + bo + C], self.bnwork[3 * n_bn + bo:3 * n_bn + bo + C]
                cv.stats = self.stats_all[so:so + 2 * R * C].view(R, 2, C)
                so += 2 * R * C
                cv.sums = self.stats_all[so:so + 2 * (R + 1) * C]
                so += 2 * (R + 1) * C
                bo += C
            else:
                cv.bias = self.params[cv.off_a:cv.off_a + C]
                cv.dbias = self.grads[cv.off_a:cv.off_a + C]
            if cv.role == "stem":       # forward and weight gradient read the fp32 master weights / image directly
                cv.pk = None
                cv.wpack = None
            elif self.fp32:             # fp32 mode: no packed images, the kernels read the master weights
                cv.pk = ops32.PackedConv(cv.cout, cv.cin, cv.k, cv.s)
                cv.pk.refresh(cv.w)
                cv.wpack = None
            else:
                cv.pk = ops.PackedConv(cv.cout, cv.cin, cv.k, cv.s, dev, need_dgrad=True)
                cv.wpack = cv.w
        if self.fp32:
            class _NoPack:
                def run(self):
                    pass
            self._pack_all = self._pack_head = _NoPack()
            self._pack_seg = [_NoPack() for _ in EARLY_SPLITS + (0,)]
        else:
            self._pack_all = ops.PackBatch([(cv.pk, cv.wpack) for cv in self.layers if cv.pk is not None], dev)
            self._pack_head = ops.PackBatch([(cv.pk, cv.wpack) for cv in self.layers[BACKBONE_CONVS:]], dev)
        # early-optimiser segments: stage 5 of the backbone + the head, then stage 4, have their gradients long before
        # backward ends
        # segments [EARLY_SPLITS[k], EARLY_SPLITS[k-1]) from the end of the network, then the rest
        if not self.fp32:
            self._pack_seg = []
            hi = len(self.layers)
            for lo in EARLY_SPLITS + (0,):
                self._pack_seg.append(ops.PackBatch([(cv.pk, cv.wpack) for cv in self.layers[lo:hi] if cv.pk is not None], dev))
                hi = lo
        self.training = True
        self.freeze_backbone = False
        self.freeze_all_but_pred = False
        self.freeze_bn = False          # every BatchNorm on its moving statistics (convs stay trainable)
        # Weight gradients are off the critical path of backward (nothing consumes them before the optimiser):
        # they run on a side stream and fill the CUs that the small / tail-heavy kernels of the dgrad -> BN chain
        # leave idle.  Needs one dy buffer per layer (no scratch reuse while a side-stream wgrad may read it).
        self.overlap_wgrad = True
        self.wg_stream = shared_stream("wgrad", dev)      # one per process and device (streams.py: hardware queues are few)
        self.side_bias_grad = True      # bias gradients of the prediction convs ride on the weight-gradient side stream
        self.parallel_heads = True      # forward: prediction branches of the first two scales on the side stream (see forward)
        self.parallel_heads_min_pixels = 8 * 608 * 608
        self.fuse_stem_bn = not self.fp32        # stem: BN backward applied inside the weight-gradient kernel (dy0 never written)
        self.fuse_bn_reduce = not self.fp32      # BN-backward reduction inside the dgrad epilogue that produces `da` (A/B: 0.9 ms/step faster)
        self._arenas = {}
        self.reset_parameters(seed)

    # ------------------------------------------------------------------ parameters
    def reset_parameters(self, seed=0):
        """Glorot-uniform kernels (Keras default), gamma=1, beta=0, bias=0, moving mean 0 / var 1."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.n_params, np.float32)
        for cv in self.layers:
            lim = math.sqrt(6.0 / (cv.T * cv.cin + cv.T * cv.cout))
            k = rng.uniform(-lim, lim, size=(cv.k, cv.k, cv.cin, cv.cout)).astype(np.float32)   # Keras HWIO
            host[cv.off_w:cv.off_w + k.size] = np.transpose(k, (3, 0, 1, 2)).reshape(-1)       # -> OHWI
            if cv.bn:
                host[cv.off_a:cv.off_a + cv.cout] = 1.0
        self.params.copy_(torch.from_numpy(host))
        n_bn = self.moving.numel() // 2
        self.moving[:n_bn] = 0.0
        self.moving[n_bn:] = 1.0
        self.refresh_packed()

    def load_keras_style(self, plist):
        """plist: list (graph order) of dicts with 'kernel' (kh,kw,cin,cout) and gamma/beta/moving_* or bias."""
        host = self.params.cpu().numpy().copy()
        mov = self.moving.cpu().numpy().copy()
        n_bn = mov.size // 2
        bo = 0
        for cv, p in zip(self.layers, plist):
            k = np.asarray(p["kernel"], np.float32)
            host[cv.off_w:cv.off_w + k.size] = np.transpose(k, (3, 0, 1, 2)).reshape(-1)
            if cv.bn:
                host[cv.off_a:cv.off_a + cv.cout] = p["gamma"]
                host[cv.off_b:cv.off_b + cv.cout] = p["beta"]
                mov[bo:bo + cv.cout] = p["moving_mean"]
                mov[n_bn + bo:n_bn + bo + cv.cout] = p["moving_var"]
                bo += cv.cout
            else:
                host[cv.off_a:cv.off_a + cv.cout] = p["bias"]
        self.params.copy_(torch.from_numpy(host))
        self.moving.copy_(torch.from_numpy(mov))
        self.refresh_packed()

    def export_keras_style(self):
        host = self.params.cpu().numpy()
        out = []
        for cv in self.layers:
            w = host[cv.off_w:cv.off_w + cv.cout * cv.T * cv.cin].reshape(cv.cout, cv.k, cv.k, cv.cin)
            p = {"kernel": np.transpose(w, (1, 2, 3, 0)).copy()}
            if cv.bn:
                p["gamma"] = host[cv.off_a:cv.off_a + cv.cout].copy()
                p["beta"] = host[cv.off_b:cv.off_b + cv.cout].copy()
                p["moving_mean"] = cv.mm.cpu().numpy().copy()
                p["moving_var"] = cv.mv.cpu().numpy().copy()
            else:
                p["bias"] = host[cv.off_a:cv.off_a + cv.cout].copy()
            out.append(p)
        return out

    def refresh_packed(self, first=0):
        """Rewrite the bf16 GEMM images from the fp32 masters: one launch for the whole network."""
        (self._pack_all if first == 0 else self._pack_head).run()

    def count_params(self):
        return self.n_params + self.moving.numel()

    def trainable_range(self):
        """[begin, end) slice of the flat buffers that the optimiser updates."""
        if self.freeze_all_but_pred:
            return None     # handled per layer (three disjoint slices)
        return (self.backbone_end if self.freeze_backbone else 0, self.n_params)

    # ------------------------------------------------------------------ arena
    def arena(self, B, H, W):
        key = (B, H, W)
        if key in self._arenas:
            return self._arenas[key]
        dev = self.device
        A = {"y": {}, "a": {}, "hw": {}}
        bf = self.act_dtype

        def alloc(i, h, w):
            cv = self.layers[i]
            A["hw"][i] = (h, w)
            if cv.bn:
                A["y"][i] = torch.empty(B, h, w, cv.cout, dtype=bf, device=dev)
                A["a"][i] = torch.empty(B, h, w, cv.cout, dtype=bf, device=dev)
            else:
                A["y"][i] = torch.empty(B, h, w, cv.cout, dtype=torch.float32, device=dev)

        i = 0
        alloc(0, H, W)
        h, w = H, W
        i = 1
        for f, n in STAGES:
            h, w = h // 2, w // 2
            alloc(i, h, w)
            i += 1
            for _ in range(n):
                alloc(i, h, w)
                alloc(i + 1, h, w)
                i += 2
        gh, gw = H // 32, W // 32
        A["cat"] = {}
        for sc in range(3):
            for j in range(5):
                alloc(i + j, gh, gw)
            i += 5
            if sc < 2:
                alloc(i, gh, gw)
                lat = self.layers[i]
                skip_c = 512 if sc == 0 else 256
                A["cat"][sc] = torch.empty(B, 2 * gh, 2 * gw, lat.cout + skip_c, dtype=bf, device=dev)
                i += 1
                gh, gw = gh * 2, gw * 2
        A["scratch"] = {}
        self._arenas[key] = A
        return A

    def _scratch(self, A, tag, shape, dtype=None):
        dtype = self.act_dtype if dtype is None else dtype
        key = (tag, tuple(shape), dtype)
        t = A["scratch"].get(key)
        if t is None:
            t = torch.empty(*shape, dtype=dtype, device=self.device)
            A["scratch"][key] = t
        return t

    # ------------------------------------------------------------------ forward
    def _bn_training(self, cv):
        if not self.training or self.freeze_bn:
            return False
        if self.freeze_all_but_pred:
            return False
        if self.freeze_backbone and cv.idx < BACKBONE_CONVS:
            return False           # Keras: trainable=False puts BatchNormalization in inference mode
        return True

    # ------------------------------------------------------------------ BatchNorm-folded inference
    def fold_bn(self, on=True):
        """Inference with every BatchNorm folded into its conv: weights pre-scaled by gamma / sqrt(moving_var + eps), the BN
        shift as bias, LeakyReLU and the residual add in the conv epilogue - one launch per DarknetConv2D_BN_Leaky
        (models/layers.py:88-95) instead of conv + BN/activation pass.  Opt-in (the unfolded path is the one the parity
        tests pin; folding rounds the scaled weights to bf16 instead of rounding y).  Call again after the weights change."""
        if on and self.fp32:
            raise RuntimeError("fold_bn is a bf16 inference mode")
        if on and self.training:
            raise RuntimeError("fold_bn is an inference-only mode: set training = False first (a training forward through "
                               "folded convs would leave the raw conv outputs backward needs stale)")
        self.folded = bool(on)
        if not on:
            return
        if getattr(self, "_fold_imgs", None) is None:
            self._fold_imgs = {cv.idx: torch.zeros_like(cv.pk.fwd) for cv in self.layers if cv.bn and cv.pk is not None}
            self._fold_w = torch.zeros_like(self.params)
            self._fold_shift = {}
        fw = self._fold_w
        fw.copy_(self.params)
        for cv in self.layers:
            if not cv.bn:
                continue
            sc = cv.gamma / torch.sqrt(cv.mv + ops.BN_EPS)
            self._fold_shift[cv.idx] = (cv.beta - cv.mm * sc).contiguous()
            n = cv.cout * cv.T * cv.cin
            fw[cv.off_w:cv.off_w + n].view(cv.cout, -1).mul_(sc[:, None])
        jobs = []
        for cv in self.layers:
            if cv.bn and cv.pk is not None:
                n = cv.cout * cv.T * cv.cin
                jobs.append((cv, fw[cv.off_w:cv.off_w + n].view(cv.cout, cv.T, cv.cin)))
        for cv, w in jobs:                      # one-off: pack the scaled masters into the folded forward images
            keep = cv.pk.fwd
            cv.pk.fwd = self._fold_imgs[cv.idx]
            cv.pk.refresh_fwd(w)
            cv.pk.fwd = keep

    def latency_workspace(self):
        """This network's workspace of the latency-form convolutions ON THE CURRENT STREAM (ops.LatencyWorkspace: uncached
        tickets + partial tiles).  One per (network, stream): two models, or one model driven from two streams, never share
        tickets (no global mutable state, safe to call concurrently on different streams)."""
        if self.fp32 or not ops.LATENCY:
            return None
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._lat_ws.get(key)
        if ws is None:
            ws = self._lat_ws[key] = ops.LatencyWorkspace(self.device)
        return ws

    def _conv_bn_act(self, A, i, x, residual=None):
        cv = self.layers[i]
        tr = self._bn_training(cv)
        y = A["y"][i]
        if getattr(self, "folded", False) and not tr and cv.role == "stem":
            n = cv.cout * cv.T * cv.cin
            return self.O.stem_fwd_act(x, self._fold_w[cv.off_w:cv.off_w + n], self._fold_shift[cv.idx], ops.LEAKY_SLOPE,
                                       out=A["a"][i])
        if getattr(self, "folded", False) and not tr:
            return self.O.conv_fwd(x, cv.pk, out=A["a"][i], bias=self._fold_shift[cv.idx], act_slope=ops.LEAKY_SLOPE,
                                addend=residual, wimg=self._fold_imgs[cv.idx], lat_ws=self.latency_workspace())
        if cv.role == "stem":       # matrix-core stem straight from the fp32 image (no im2col image in the forward pass)
            self.O.stem_fwd(x, cv.w, out=y, stats=cv.stats if tr else None)
        else:
            self.O.conv_fwd(x, cv.pk, out=y, stats=cv.stats if tr else None)
        P = y.numel() // cv.cout
        return self.O.bn_act_fwd_fused(cv.stats, float(P), cv.gamma, cv.beta, cv.mm, cv.mv, cv.scale, cv.shift,
                                    cv.smean, cv.sinv, y, A["a"][i], residual=residual, training=tr)

    def forward(self, images):
        """images: fp32 CUDA [B,H,W,3] in [0,1].  Returns [y1, y2, y3] raw head tensors (fp32 NHWC)."""
        B, H, W, _ = images.shape
        assert H % 32 == 0 and W % 32 == 0
        if self.training and getattr(self, "folded", False):
            raise RuntimeError("training forward with fold_bn on: call fold_bn(False) first")
        A = self.arena(B, H, W)
        A["image"] = images
        if self.training:
            ops.memset0(self.stats_all)         # (a launch of its own: not in front of an inference forward)
        x = self._conv_bn_act(A, 0, images)
        i = 1
        feats = {}
        for f, n in STAGES:
            x = self._conv_bn_act(A, i, x)
            i += 1
            for _ in range(n):
                a1 = self._conv_bn_act(A, i, x)
                x = self._conv_bn_act(A, i + 1, a1, residual=x)
                i += 2
            feats[f] = x
        skips = (feats[512], feats[256])
        x = feats[1024]
        outs = []
        # Behind `xb` a scale splits into its prediction branch (3x3 + 1x1, a few thousand pixels: launches that fill a fraction
        # of the chip) and the 1x1 + up-sampling that feeds the next scale.  The prediction branches of the first two scales run
        # on the side stream (idle during a forward pass) beside the next scale's trunk; the main stream joins at the end.
        # Measured: 12.09 -> 12.06 ms per train step, 2.96 -> 2.93 ms per batch-16 inference forward; at batch 1 the two
        # cross-stream waits cost more than the overlap saves (0.82 -> 0.87 ms), so small batches stay on one stream.  (The same
        # for the BACKWARD pass - those branches first on the side stream, picked up behind an event - measured 0.04 ms slower:
        # they delay the weight gradients the side stream is there for.)
        side = self.wg_stream if (self.parallel_heads and not self.fp32 and B * H * W >= self.parallel_heads_min_pixels
                                  and not torch.cuda.is_current_stream_capturing()) else None
        main = torch.cuda.current_stream(self.device)
        forked = False
        for sc in range(3):
            x = self._conv_bn_act(A, i, x)
            x = self._conv_bn_act(A, i + 1, x)
            xb = self._conv_bn_act(A, i + 2, x)
            pred = self.layers[i + 4]
            if side is not None and sc < 2:
                ops.stream_wait(side, main)
                with torch.cuda.stream(side):
                    a4 = self._conv_bn_act(A, i + 3, xb)
                    outs.append(self.O.conv_fwd(a4, pred.pk, out=A["y"][i + 4], bias=pred.bias, out_f32=True))
                forked = True
            else:
                a4 = self._conv_bn_act(A, i + 3, xb)
                outs.append(self.O.conv_fwd(a4, pred.pk, out=A["y"][i + 4], bias=pred.bias, out_f32=True))
            i += 5
            if sc < 2:
                a6 = self._conv_bn_act(A, i, xb)
                x = self.O.upsample_concat_fwd(a6, skips[sc], A["cat"][sc])
                i += 1
        if forked:
            ops.stream_wait(main, side)
        A["outs"] = outs
        self._last = A
        return outs

    # ------------------------------------------------------------------ backward
    def _bnred(self, A, i):
        """BN context of layer i for a dgrad launch that writes layer i's `da`: the backward reduction
        (sum dyh, sum dyh*yhat) is then fused into that launch's epilogue.  None if the layer's BN is frozen."""
        cv = self.layers[i]
        if not self.fuse_bn_reduce or not self._bn_training(cv):
            return None
        A["reduced"].add(i)
        return (A["y"][i], cv.scale, cv.shift, cv.smean, cv.sinv, cv.sums)

    def _bwd_bn(self, A, i, da):
        """da: grad wrt the activated output of layer i -> returns dy (grad wrt the raw conv output)."""
        cv = self.layers[i]
        y = A["y"][i]
        dy = self._scratch(A, ("dy", i) if self.overlap_wgrad else "dy", y.shape)
        frozen = not self._bn_training(cv)
        self.O.bn_act_bwd(da, y, cv.scale, cv.shift, cv.smean, cv.sinv, cv.sums, cv.dgamma, cv.dbeta, dy, frozen=frozen,
                       reduced=i in A["reduced"])
        return dy

    def _wgrad_workspace(self):
        """fp32 workspace of the kernel-row weight gradient's per-split slabs (mgd_wgrad_desc.partial): one buffer per network,
        used by the weight-gradient launches in stream order (they all run on one stream).  64 MB covers every layer of the
        graph at batch 16, 608 x 608 (<= 256 slabs of a 128 x 128 x 3-tap tile); a launch that needs more falls back to atomics."""
        if not ops.WGRAD_ROW_FORM:
            return None
        if self._wg_ws is None:
            self._wg_ws = torch.empty(16 << 20, dtype=torch.float32, device=self.device)
        return self._wg_ws

    def _wgrad(self, x, dy, dw, k, s, dbias=None):
        """Weight gradient (and, for the biased prediction convs, the bias gradient: nothing on the dgrad/BN chain
        reads it, so it leaves the critical stream too)."""
        kw = {} if self.fp32 else {"ws": self._wgrad_workspace()}
        if not self.overlap_wgrad:
            self.O.conv_wgrad(x, dy, dw, k, s, **kw)
            if dbias is not None:
                self.O.bias_grad(dy, dbias)
            return
        ops.stream_wait(self.wg_stream)
        with torch.cuda.stream(self.wg_stream):
            self.O.conv_wgrad(x, dy, dw, k, s, **kw)
            if dbias is not None:
                self.O.bias_grad(dy, dbias)

    def backward(self, douts, on_layer_done=None):
        """douts: three bf16 grads wrt the head outputs.  Accumulates into self.grads (zero it first).
        on_layer_done(i) is called when every gradient of layers >= i is final (DP bucket hook)."""
        A = self._last
        Lr = self.layers
        B = douts[0].shape[0]
        acts = A["a"]
        hw = A["hw"]
        train_head_only = self.freeze_backbone
        pred_only = self.freeze_all_but_pred

        if "fin" not in A:
            self._wire(A)
        fin = A["fin"]
        A["reduced"] = set()

        def inp(i):
            return fin[i]

        head0 = BACKBONE_CONVS
        d_up = {}            # grads for lateral activations, keyed by scale
        d_skip = {}          # grads for the backbone taps
        g_f1 = None
        for sc in (2, 1, 0):
            base = head0 + sc * 6
            c1, c2, c3, c4, c5 = base, base + 1, base + 2, base + 3, base + 4
            pred = Lr[c5]
            dy5 = douts[sc]
            a4 = acts[c4]
            if self.side_bias_grad:
                self._wgrad(a4, dy5, pred.dw, 1, 1, dbias=pred.dbias)
            else:
                self.O.bias_grad(dy5, pred.dbias)
                self._wgrad(a4, dy5, pred.dw, 1, 1)
            if on_layer_done and pred_only:
                on_layer_done(c5)
            if pred_only:
                continue
            d_a4 = self.O.conv_dgrad(dy5, pred.pk, hw[c4], out=self._scratch(A, "da_a", a4.shape), bnred=self._bnred(A, c4))
            dy4 = self._bwd_bn(A, c4, d_a4)
            xb = acts[c3]
            self._wgrad(xb, dy4, Lr[c4].dw, 3, 1)
            # xb (= activation of c3) has a second consumer (the lateral conv c6) unless this is the last scale:
            # the launch that writes the FINAL d_xb carries c3's BN reduction
            d_xb = self.O.conv_dgrad(dy4, Lr[c4].pk, hw[c3], out=self._scratch(A, "da_b", xb.shape),
                                  bnred=self._bnred(A, c3) if sc == 2 else None)
            if sc < 2:
                c6 = base + 5
                dy6 = self._bwd_bn(A, c6, d_up[sc])
                self._wgrad(xb, dy6, Lr[c6].dw, 1, 1)
                self.O.conv_dgrad(dy6, Lr[c6].pk, hw[c3], out=d_xb, addend=d_xb, bnred=self._bnred(A, c3))
            dy3 = self._bwd_bn(A, c3, d_xb)
            a2 = acts[c2]
            self._wgrad(a2, dy3, Lr[c3].dw, 1, 1)
            d_a2 = self.O.conv_dgrad(dy3, Lr[c3].pk, hw[c2], out=self._scratch(A, "da_a", a2.shape), bnred=self._bnred(A, c2))
            dy2 = self._bwd_bn(A, c2, d_a2)
            a1 = acts[c1]
            self._wgrad(a1, dy2, Lr[c2].dw, 3, 1)
            d_a1 = self.O.conv_dgrad(dy2, Lr[c2].pk, hw[c1], out=self._scratch(A, "da_b", a1.shape), bnred=self._bnred(A, c1))
            dy1 = self._bwd_bn(A, c1, d_a1)
            xin = A["cat"][sc - 1] if sc > 0 else acts[BACKBONE_CONVS - 1]
            self._wgrad(xin, dy1, Lr[c1].dw, 1, 1)
            if sc > 0:
                d_cat = self.O.conv_dgrad(dy1, Lr[c1].pk, hw[c1], out=self._scratch(A, "dcat", xin.shape))
                lat = Lr[head0 + (sc - 1) * 6 + 5]
                gh, gw = hw[lat.idx]
                du = self._scratch(A, f"du{sc}", (B, gh, gw, lat.cout))
                skip_c = xin.shape[-1] - lat.cout
                dsk = self._scratch(A, f"dskip{sc}", (B, 2 * gh, 2 * gw, skip_c))
                self.O.upsample_concat_bwd(d_cat, du, dsk)
                d_skip[sc - 1] = dsk
                d_up[sc - 1] = du
            elif not train_head_only:
                g_f1 = self.O.conv_dgrad(dy1, Lr[c1].pk, hw[c1], out=self._scratch(A, "g5", acts[BACKBONE_CONVS - 1].shape),
                                      bnred=self._bnred(A, BACKBONE_CONVS - 1))
            # the hook comes AFTER the last reader of c1's packed weights (its data gradient above): with the per-bucket
            # optimiser it re-packs Lr[c1].pk on the communication stream, which only waits for work enqueued before it
            if on_layer_done:
                on_layer_done(c1)
        if pred_only or train_head_only:
            self._join_wgrad()
            if on_layer_done:
                on_layer_done(0)
            return
        # ---- backbone, last stage first.  g = grad wrt the running residual sum x.
        g = g_f1
        i = BACKBONE_CONVS - 1
        for st in range(len(STAGES) - 1, -1, -1):
            f, n = STAGES[st]
            for _ in range(n):
                l2, l1 = i, i - 1
                x_in = inp(l1)
                dy2 = self._bwd_bn(A, l2, g)                     # residual branch: d a2 = g
                a1 = acts[l1]
                self._wgrad(a1, dy2, Lr[l2].dw, 3, 1)
                d_a1 = self.O.conv_dgrad(dy2, Lr[l2].pk, hw[l1], out=self._scratch(A, "da_a", a1.shape),
                                      bnred=self._bnred(A, l1))
                dy1 = self._bwd_bn(A, l1, d_a1)
                self._wgrad(x_in, dy1, Lr[l1].dw, 1, 1)
                # g <- g + dgrad (in place); g is then the `da` of the layer that produced x_in (l1 - 1)
                self.O.conv_dgrad(dy1, Lr[l1].pk, hw[l1], out=g, addend=g, bnred=self._bnred(A, l1 - 1))
                if on_layer_done:
                    on_layer_done(l1)
                i -= 2
            ld = i
            x_prev = inp(ld)
            dyd = self._bwd_bn(A, ld, g)
            if st == 0:
                self._wgrad(x_prev, dyd, Lr[ld].dw, 3, 2)
                g = self.O.conv_dgrad(dyd, Lr[ld].pk, hw[0], out=self._scratch(A, "g0", x_prev.shape), bnred=self._bnred(A, 0))
            else:
                self._wgrad(x_prev, dyd, Lr[ld].dw, 3, 2)
                add = None
                if st == 4:
                    add = d_skip[0]      # f2 (stage-4 output) also fed the scale-2 concat
                elif st == 3:
                    add = d_skip[1]      # f3 (stage-3 output) also fed the scale-3 concat
                g = self.O.conv_dgrad(dyd, Lr[ld].pk, hw[ld - 1], out=self._scratch(A, f"g{st}", x_prev.shape), addend=add,
                                   bnred=self._bnred(A, ld - 1))
            if on_layer_done:
                on_layer_done(ld)
            i -= 1
        c0 = Lr[0]
        if self.fuse_stem_bn and 0 in A["reduced"] and self._bn_training(c0):
            # the sums are already there (fused into layer 1's input gradient): apply BN backward inside the weight
            # gradient, dy0 never touches HBM
            self.O.stem_wgrad_bn(A["image"], g, A["y"][0], c0.scale, c0.shift, c0.smean, c0.sinv, c0.sums, c0.dgamma,
                              c0.dbeta, c0.dw)
        else:
            dy0 = self._bwd_bn(A, 0, g)
            side = self.wg_stream if self.overlap_wgrad else torch.cuda.current_stream()
            ops.stream_wait(side)
            with torch.cuda.stream(side):
                self.O.stem_wgrad(A["image"], dy0, c0.dw)       # matrix cores, straight from the fp32 image
        self._join_wgrad()
        if on_layer_done:
            on_layer_done(0)

    def _join_wgrad(self):
        if self.overlap_wgrad:
            ops.stream_wait(torch.cuda.current_stream(), self.wg_stream)

    def _wire(self, A):
        """Forward input activation of each backbone conv, per arena."""
        acts = A["a"]
        fin = {}
        i = 1
        prev = 0
        for f, n in STAGES:
            fin[i] = acts[prev]
            x = i
            i += 1
            for _ in range(n):
                fin[i] = acts[x]
                fin[i + 1] = acts[i]
                x = i + 1
                i += 2
            prev = x
        A["fin"] = fin

    def zero_grad(self):
        ops.memset0(self.grads)
