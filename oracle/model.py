"""CPU oracle for the Darknet53 + top-down FPN + DenseYOLO-head network.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, as the checker / CPU baseline.  The product path never imports `oracle`.

PARITY UNPINNED at the Keras boundary (conv/BN arithmetic lives in TensorFlow, which is absent
here, and the reference's tests hold no activations).  This is a torch-CPU fp32
restatement of the graph the reference builds:
  M1  DarknetConv2D_BN_Leaky      multigriddet/models/layers.py:43-49, 88-95
      conv (no bias; 'valid' iff stride 2 else 'same') -> BatchNorm(eps 1e-3, momentum 0.99,
      gamma 1, beta 0, batch statistics in training) -> LeakyReLU(0.1)
  M2  darknet53_body              multigriddet/models/backbones/darknet.py:19-40
      stride-2 convs are preceded by ZeroPadding2D(((1,0),(1,0))) = pad top/left only
  M3  make_last_layers            multigriddet/models/heads/multigrid_head.py:38-74
  M4  multigriddet_predictions    multigriddet/models/heads/multigrid_head.py:275-313
  M5  build_multigriddet_darknet  multigriddet/models/multigriddet_darknet.py:488-548
      taps: layers[92] (stride 8, 256 ch), layers[152] (stride 16, 512 ch), output (stride 32)
Weights use the Keras kernel layout (kh, kw, cin, cout) so they can be shared verbatim with the
product's parameter list (same order: conv kernels / BN gamma,beta in graph order).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
LEAKY = 0.1


def layer_specs(num_classes=80, num_anchors=3):
    """List of conv specs in graph-construction order: dict(k, s, cin, cout, bn)."""
    specs = []

    def c(cin, cout, k, s=1, bn=True):
        specs.append(dict(k=k, s=s, cin=cin, cout=cout, bn=bn))

    c(3, 32, 3)
    ch = 32
    for f, n in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        c(ch, f, 3, 2)
        for _ in range(n):
            c(f, f // 2, 1)
            c(f // 2, f, 3)
        ch = f
    out = num_anchors + num_classes + 5
    cin = 1024
    for i, (n, mult, skip) in enumerate(((256, 8, 512), (128, 4, 256), (64, 2, None))):
        c(cin, n, 1)
        c(n, 2 * n, 3)
        c(2 * n, n, 1)
        c(n, mult * out, 3)
        c(mult * out, out, 1, bn=False)
        if skip is not None:
            c(n, n // 2, 1)
            cin = n // 2 + skip
    return specs


def init_params(seed=0, num_classes=80, num_anchors=3):
    """Glorot-uniform kernels (Keras default), BN gamma=1 beta=0, zero bias; numpy float32 dict list."""
    rng = np.random.default_rng(seed)
    params = []
    for sp in layer_specs(num_classes, num_anchors):
        k, cin, cout = sp["k"], sp["cin"], sp["cout"]
        lim = math.sqrt(6.0 / (k * k * cin + k * k * cout))
        p = {"kernel": rng.uniform(-lim, lim, size=(k, k, cin, cout)).astype(np.float32)}
        if sp["bn"]:
            p["gamma"] = np.ones(cout, np.float32)
            p["beta"] = np.zeros(cout, np.float32)
            p["moving_mean"] = np.zeros(cout, np.float32)
            p["moving_var"] = np.ones(cout, np.float32)
        else:
            p["bias"] = np.zeros(cout, np.float32)
        params.append(p)
    return params


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def conv_bn_leaky(x, p, sp, training, stats_out=None, emulate_bf16=False, residual=None):
    """x: NCHW torch tensor. Returns activated output (+ residual).
    emulate_bf16=True rounds the weights, the raw conv output and the activation to bf16 at the points
    where the gfx950 path stores bf16 (fp32 accumulation in between), so that the comparison with the
    product isolates logic errors from bf16 storage noise."""
    w = p["kernel"].permute(3, 2, 0, 1)                     # (kh,kw,cin,cout) -> OIHW
    if emulate_bf16:
        w = _bf16(w)
        if sp["cin"] == 3:
            x = _bf16(x)            # the product's stem consumes a bf16 im2col image
    if sp["s"] == 2:
        x = F.pad(x, (1, 0, 1, 0))                            # left, right, top, bottom: top/left only
        y = F.conv2d(x, w, stride=2)
    else:
        y = F.conv2d(x, w, padding=sp["k"] // 2)
    if not sp["bn"]:
        return y + p["bias"].view(1, -1, 1, 1)
    if emulate_bf16:
        y = _bf16(y)
    if training:
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            stats_out.append((mean.detach(), var.detach()))
    else:
        mean, var = p["moving_mean"], p["moving_var"]
    yn = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
    yn = yn * p["gamma"].view(1, -1, 1, 1) + p["beta"].view(1, -1, 1, 1)
    a = F.leaky_relu(yn, LEAKY)
    if residual is not None:
        a = a + residual
    return _bf16(a) if emulate_bf16 else a


def forward(images_nhwc, params, training=True, stats_out=None, taps_out=None, acts_out=None, emulate_bf16=False):
    """images: [B,H,W,3] float in [0,1].  Returns the three raw head tensors, NHWC.
    acts_out (optional list) receives every conv's output tensor (NCHW) in graph order; for the second
    conv of a residual block the entry is the block output (after the Add)."""
    specs = layer_specs()
    it = iter(zip(specs, params))

    def nxt(x, residual=None):
        sp, p = next(it)
        y = conv_bn_leaky(x, p, sp, training, stats_out, emulate_bf16, residual)
        if acts_out is not None:
            acts_out.append(y)
        return y

    x = images_nhwc.permute(0, 3, 1, 2)
    x = nxt(x)
    feats = {}
    for f, n in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        x = nxt(x)
        for _ in range(n):
            x = nxt(nxt(x), residual=x)
        feats[f] = x
    f1, f2, f3 = feats[1024], feats[512], feats[256]
    if taps_out is not None:
        taps_out.extend([f3, f2, f1])
    outs = []
    x = f1
    for skip in (f2, f3, None):
        x = nxt(nxt(nxt(x)))
        y = nxt(nxt(x))
        outs.append(y.permute(0, 2, 3, 1).contiguous())
        if skip is not None:
            x = nxt(x)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = torch.cat([x, skip], dim=1)
    return outs


def torch_params(params, requires_grad=False):
    out = []
    for p in params:
        q = {k: torch.tensor(v, dtype=torch.float32) for k, v in p.items()}
        if requires_grad:
            for k in ("kernel", "gamma", "beta", "bias"):
                if k in q:
                    q[k].requires_grad_(True)
        out.append(q)
    return out


def adam_step(tp, state, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam (config/model_builder.py:86-96): m,v EMA; lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    p -= lr_t * m / (sqrt(v) + eps)."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    with torch.no_grad():
        for i, p in enumerate(tp):
            for k in ("kernel", "gamma", "beta", "bias"):
                if k not in p or p[k].grad is None:
                    continue
                g = p[k].grad
                m = state.setdefault((i, k, "m"), torch.zeros_like(g))
                v = state.setdefault((i, k, "v"), torch.zeros_like(g))
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                p[k].sub_(lr_t * m / (v.sqrt() + eps))
                p[k].grad = None


def count_params(params):
    return sum(int(np.prod(v.shape)) for p in params for v in p.values())
