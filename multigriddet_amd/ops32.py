"""fp32 twins of the tensor-level wrappers in ops.py, used by Network(precision="fp32") - the strict-parity run of
the graph in the reference's default numeric type (Keras float32, models/layers.py:43-95).  Same call signatures as the
bf16 wrappers where the engine uses them; activations are fp32 NHWC, weights the fp32 OHWI masters (no packed images).
Direct kernels (csrc/fp32ref.hip): correctness tooling of the product, not its fast path."""
import ctypes as C

import torch

from . import _lib as L
from .ops import BN_EPS, BN_MOMENTUM, LEAKY_SLOPE          # noqa: F401  (same constants)

STATS_REPLICAS = 1


class PackedConv:
    """Carries what a conv needs in fp32 mode: the geometry; the weights are read from the master buffer."""

    def __init__(self, co, ci, k, s, device=None, need_dgrad=True, ci_master=None):
        self.co, self.ci, self.k, self.s = co, ci, k, s
        self.w = None                      # fp32 [Co, k*k, Ci] view of the master weights (set by the engine)

    def refresh(self, w):
        self.w = w


def _f(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()
    return L.ptr(t)


def conv_fwd(x, pk, out=None, bias=None, stats=None, out_f32=True, act_slope=0.0, addend=None, wimg=None, lat_ws=None):
    N, H, W, Ci = x.shape
    assert Ci == pk.ci and act_slope == 0.0 and addend is None and wimg is None
    Ho, Wo = (H // 2, W // 2) if pk.s == 2 else (H, W)
    if out is None:
        out = torch.empty(N, Ho, Wo, pk.co, dtype=torch.float32, device=x.device)
    L.check(L.load().mgd_conv2d_f32_fwd(_f(x), _f(pk.w), _f(out), L.ptr(bias), N, H, W, Ci, pk.co, pk.k, pk.s,
                                        L.stream_ptr()), "conv2d_f32_fwd")
    if stats is not None:
        L.check(L.load().mgd_bn_stats_f32(_f(out), C.c_int64(N * Ho * Wo), pk.co, _f(stats), L.stream_ptr()), "bn_stats_f32")
    return out


def stem_fwd(image, w, out=None, stats=None):
    pk = PackedConv(32, 3, 3, 1)
    pk.w = w
    return conv_fwd(image, pk, out=out, stats=stats)


def conv_dgrad(dy, pk, in_hw, out=None, addend=None, bnred=None):
    N, Ho, Wo, Co = dy.shape
    H, W = in_hw
    assert Co == pk.co and bnred is None
    if out is None:
        out = torch.empty(N, H, W, pk.ci, dtype=torch.float32, device=dy.device)
    L.check(L.load().mgd_conv2d_f32_dgrad(_f(dy), _f(pk.w), _f(out), L.ptr(addend), N, H, W, pk.ci, Co, pk.k, pk.s,
                                          L.stream_ptr()), "conv2d_f32_dgrad")
    return out


def conv_wgrad(x, dy, dw, k, s, splits=None):
    N, H, W, Ci = x.shape
    Co = dy.shape[-1]
    L.check(L.load().mgd_conv2d_f32_wgrad(_f(x), _f(dy), _f(dw), N, H, W, Ci, Co, k, s, L.stream_ptr()), "conv2d_f32_wgrad")
    return dw


def stem_wgrad(image, dy, dw):
    return conv_wgrad(image, dy, dw, 3, 1)


def bn_act_fwd_fused(stats, count, gamma, beta, mm, mv, scale, shift, smean, sinv, y, out, residual=None, training=True):
    Cn = y.shape[-1]
    P = y.numel() // Cn
    lib = L.load()
    L.check(lib.mgd_bn_finalize(L.ptr(stats), 1, Cn, C.c_float(count), L.ptr(gamma), L.ptr(beta), L.ptr(mm), L.ptr(mv),
                                L.ptr(scale), L.ptr(shift), L.ptr(smean), L.ptr(sinv), C.c_float(BN_EPS),
                                C.c_float(BN_MOMENTUM), int(training), L.stream_ptr()), "bn_finalize")
    L.check(lib.mgd_bn_act_fwd_f32(_f(y), L.ptr(scale), L.ptr(shift), L.ptr(residual), _f(out), C.c_int64(P), Cn,
                                   C.c_float(LEAKY_SLOPE), L.stream_ptr()), "bn_act_fwd_f32")
    return out


def bn_act_bwd(da, y, scale, shift, smean, sinv, sums, dgamma, dbeta, dy, frozen=False, reduced=False):
    assert not reduced
    Cn = y.shape[-1]
    P = y.numel() // Cn
    L.check(L.load().mgd_bn_act_bwd_f32(_f(da), _f(y), L.ptr(scale), L.ptr(shift), L.ptr(smean), L.ptr(sinv), L.ptr(sums),
                                        L.ptr(dgamma), L.ptr(dbeta), _f(dy), C.c_int64(P), Cn, C.c_float(LEAKY_SLOPE),
                                        int(frozen), L.stream_ptr()), "bn_act_bwd_f32")
    return dy


def upsample_concat_fwd(u, skip, out):
    N, h, w, Cu = u.shape
    L.check(L.load().mgd_upsample_concat_fwd_f32(_f(u), _f(skip), _f(out), N, h, w, Cu, skip.shape[-1], L.stream_ptr()),
            "upsample_concat_fwd_f32")
    return out


def upsample_concat_bwd(dout, du, dskip):
    N, h, w, Cu = du.shape
    L.check(L.load().mgd_upsample_concat_bwd_f32(_f(dout), _f(du), _f(dskip), N, h, w, Cu, dskip.shape[-1], L.stream_ptr()),
            "upsample_concat_bwd_f32")


def bias_grad(dy, dbias):
    Cn = dy.shape[-1]
    L.check(L.load().mgd_bias_grad_f32(_f(dy), L.ptr(dbias), C.c_int64(dy.numel() // Cn), Cn, L.stream_ptr()), "bias_grad_f32")
