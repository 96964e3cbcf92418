"""Thin tensor-level wrappers over the C-ABI (one Python function per entry point family).

Everything here takes/returns torch CUDA tensors and enqueues on torch's current stream.  Layouts:
activations NHWC bf16, master weights fp32 OHWI ([Co, kh*kw, Ci]), packed GEMM images bf16.
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib as L

BN_EPS = 1e-3          # Keras BatchNormalization default (reference models/layers.py:94)
BN_MOMENTUM = 0.99
LEAKY_SLOPE = 0.1
STATS_REPLICAS = 16
# bench.py sets this to a list to bracket every gather-GEMM launch with HIP events on the launch
# stream: entries are (start_event, end_event, algorithmic_flops, kernel_variant, pass).
PROFILE = None


# Kernel form of the gather-GEMM launches (mgd_conv_desc.form / form_arg; 0 = the library's dispatch).  The library itself
# reads no environment: tests and tools set these module attributes, MGD_CONV_FORM / MGD_CONV_FORM_ARG pre-set them.
CONV_FORM = int(os.environ.get("MGD_CONV_FORM", "0"))
CONV_FORM_ARG = int(os.environ.get("MGD_CONV_FORM_ARG", "0"))


CONV_FORM_SOFT = os.environ.get("MGD_CONV_FORM_SOFT", "0") == "1"   # measurement runs: a forced form that cannot run a layer falls back


def _gather_gemm(lib, d, what):
    rc = lib.mgd_conv_gather_gemm(C.byref(d), L.stream_ptr())
    if rc == -1 and d.form and CONV_FORM_SOFT:
        d.form = d.form_arg = 0
        rc = lib.mgd_conv_gather_gemm(C.byref(d), L.stream_ptr())
    L.check(rc, what)


def _launch_gemm(d, what):
    lib = L.load()
    if PROFILE is None:
        _gather_gemm(lib, d, what)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _gather_gemm(lib, d, what)
    e1.record()
    variant = lib.mgd_last_kernel().decode()      # the kernel family the library dispatched this launch to
    # algorithmic bytes: source + packed weights + destination (+ residual addend, + the y of a fused BN-backward reduction)
    nd = d.N * d.Hd * d.Wd * d.Co
    by = 2.0 * d.N * d.Hs * d.Ws * d.Ci + 2.0 * d.Co_pad * d.K_pad + nd * (4 if d.dst_f32 else 2) + \
        (2.0 * nd if d.addend else 0.0) + (2.0 * nd if d.bn_y else 0.0)
    PROFILE.append((e0, e1, 2.0 * d.N * d.Hg * d.Wg * d.ntaps * d.Ci * d.Co, variant, what, by))


def memset0(t):
    """Zero-fill on the current stream through the library (hipMemsetAsync): part of a recorded launch plan, unlike t.zero_()."""
    L.check(L.load().mgd_memset_async(L.ptr(t), 0, C.c_int64(t.numel() * t.element_size()), L.stream_ptr()), "memset_async")
    return t


def stream_wait(waiting, signalling=None):
    """`waiting` (a torch stream) waits for everything enqueued so far on `signalling` (default: the current stream); recorded
    when a launch plan is being recorded (_lib.Recorder)."""
    sig = signalling if signalling is not None else torch.cuda.current_stream()
    if waiting.cuda_stream == sig.cuda_stream:
        return
    ev = torch.cuda.Event()
    ev.record(sig)
    waiting.wait_event(ev)
    if L.RECORDER is not None:
        L.RECORDER.wait(waiting, sig)


def _ru(x, m):
    return (x + m - 1) // m * m


def pad_cout(co):
    for tile in (128, 64, 32):
        p = _ru(co, tile)
        if p / co - 1.0 <= 0.125:
            return p
    return _ru(co, 32)


# ------------------------------------------------------------------------------------------- taps
def taps_fwd(k):
    if k == 1:
        return [0], [0]
    return [kh - 1 for kh in range(3) for _ in range(3)], [kw - 1 for _ in range(3) for kw in range(3)]


def taps_dgrad_s2(ph, pw):
    """Output-parity class (ph, pw) of the stride-2 transposed conv: (dh, dw, source tap)."""
    out = []
    for kh in ((1,) if ph == 0 else (0, 2)):
        for kw in ((1,) if pw == 0 else (0, 2)):
            out.append(((ph + 1 - kh) // 2, (pw + 1 - kw) // 2, kh * 3 + kw))
    return out


# ------------------------------------------------------------------------------------------- packing
class PackedConv:
    """Packed bf16 images of one conv's weights: forward + data-gradient variants."""

    def __init__(self, co, ci, k, s, device, need_dgrad=True, ci_master=None):
        self.co, self.ci, self.k, self.s = co, ci, k, s
        self.ci_master = ci if ci_master is None else ci_master    # channels of the fp32 master (stem: 27 of 32)
        T = k * k
        self.T = T
        self.fwd_kpad = _ru(T * ci, 64)
        self.fwd_copad = pad_cout(co)
        self.fwd = torch.zeros(self.fwd_copad, self.fwd_kpad, dtype=torch.bfloat16, device=device)
        self.dgrad = []
        if need_dgrad:
            if s == 1:
                kp, cp = _ru(T * co, 64), pad_cout(ci)
                self.dgrad.append((torch.zeros(cp, kp, dtype=torch.bfloat16, device=device), kp, cp,
                                   [T - 1 - t for t in range(T)], None))
            else:
                for ph in range(2):
                    for pw in range(2):
                        tp = taps_dgrad_s2(ph, pw)
                        kp, cp = _ru(len(tp) * co, 64), pad_cout(ci)
                        self.dgrad.append((torch.zeros(cp, kp, dtype=torch.bfloat16, device=device), kp, cp,
                                           [t[2] for t in tp], (ph, pw, tp)))

    def refresh_fwd(self, w):
        """w: fp32 [Co, T, Ci] -> rewrite the forward image only."""
        src = (C.c_int32 * 9)(*range(self.T), *([0] * (9 - self.T)))
        L.check(L.load().mgd_pack_weights(L.ptr(w), L.ptr(self.fwd), self.co, self.T, self.ci_master, 0, self.T, src,
                                          self.fwd_copad, self.fwd_kpad, L.stream_ptr()), "pack fwd")

    def refresh(self, w):
        """w: fp32 [Co, T, Ci] master weights -> rewrite the packed images."""
        lib = L.load()
        src = (C.c_int32 * 9)(*range(self.T), *([0] * (9 - self.T)))
        L.check(lib.mgd_pack_weights(L.ptr(w), L.ptr(self.fwd), self.co, self.T, self.ci_master, 0, self.T, src,
                                     self.fwd_copad, self.fwd_kpad, L.stream_ptr()), "pack fwd")
        for img, kp, cp, st, _ in self.dgrad:
            src = (C.c_int32 * 9)(*st, *([0] * (9 - len(st))))
            L.check(lib.mgd_pack_weights(L.ptr(w), L.ptr(img), self.co, self.T, self.ci_master, 1, len(st), src, cp, kp,
                                         L.stream_ptr()), "pack dgrad")


class PackBatch:
    """All packed images of a list of (PackedConv, master weight view) in one launch."""

    def __init__(self, pairs, device):
        jobs = []
        begin = 0
        for pk, w in pairs:
            imgs = [(pk.fwd, 0, pk.T, list(range(pk.T)), pk.fwd_copad, pk.fwd_kpad)]
            for img, kp, cp, st, _ in pk.dgrad:
                imgs.append((img, 1, len(st), st, cp, kp))
            for img, tr, nt, st, rows_pad, kpad in imgs:
                j = L.PackJob()
                j.w, j.out = w.data_ptr(), img.data_ptr()
                j.Co, j.T, j.Ci, j.transpose, j.ntaps_out = pk.co, pk.T, pk.ci_master, tr, nt
                j.rows_pad, j.K_pad = rows_pad, kpad
                if rows_pad % 128 == 0 and kpad % 64:
                    # the job table lives in device memory, so mgd_pack_weights_batch cannot check it: 128-row images are
                    # written in fragment order (blocks of 128 rows x 64 K), as mgd_pack_weights enforces
                    raise ValueError(f"PackBatch: K_pad={kpad} must be a multiple of 64 for a {rows_pad}-row image")
                code = 0
                for t, sidx in enumerate(st):
                    code |= int(sidx) << (4 * t)
                j.srccode, j.begin = code, begin
                rows, cin = (pk.ci_master, pk.co) if tr else (pk.co, pk.ci_master)
                begin += nt * (-(-cin // 64)) * (-(-rows // 32))      # 32-row x 64-column tiles of the valid region
                jobs.append(j)
        self.n, self.total = len(jobs), begin
        self.pks = [pk for pk, _ in pairs]
        arr = (L.PackJob * len(jobs))(*jobs)
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)

    def run(self):
        L.check(L.load().mgd_pack_weights_batch(L.ptr(self.table), self.n, C.c_int64(self.total), L.stream_ptr()),
                "pack_batch")


def stem_im2col(image, out=None):
    N, H, W, _ = image.shape
    if out is None:
        out = torch.empty(N, H, W, 32, dtype=torch.bfloat16, device=image.device)
    L.check(L.load().mgd_stem_im2col(L.ptr(image), L.ptr(out), N, H, W, L.stream_ptr()), "stem_im2col")
    return out


def _desc(src, wpk, dst, N, Hs, Ws, Ci, Hg, Wg, Hd, Wd, Co, in_stride, out_stride, off, dh, dw, K_pad, Co_pad,
          bias=None, addend=None, stats=None, dst_f32=False, bnred=None, act_slope=0.0):
    d = L.ConvDesc()
    d.act_slope = act_slope
    d.form, d.form_arg = CONV_FORM, CONV_FORM_ARG
    d.src, d.wpk, d.dst = src.data_ptr(), wpk.data_ptr(), dst.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.addend = addend.data_ptr() if addend is not None else None
    d.stats = stats.data_ptr() if stats is not None else None
    d.N, d.Hs, d.Ws, d.Ci, d.Hg, d.Wg, d.Hd, d.Wd, d.Co = N, Hs, Ws, Ci, Hg, Wg, Hd, Wd, Co
    d.in_stride, d.out_stride, d.out_off_h, d.out_off_w = in_stride, out_stride, off[0], off[1]
    d.ntaps = len(dh)
    for i, (a, b) in enumerate(zip(dh, dw)):
        d.dh[i], d.dw[i] = a, b
    d.K_pad, d.Co_pad, d.dst_f32 = K_pad, Co_pad, int(dst_f32)
    d.stats_replicas = STATS_REPLICAS if (stats is not None or bnred is not None) else 0
    if bnred is not None:       # (y, scale, shift, mean, invstd, sums) of the layer whose `da` this launch writes
        y, sc, sh, mu, iv, sums = bnred
        d.bn_y, d.bn_scale, d.bn_shift = y.data_ptr(), sc.data_ptr(), sh.data_ptr()
        d.bn_mean, d.bn_invstd, d.bn_sums = mu.data_ptr(), iv.data_ptr(), sums.data_ptr()
        d.bn_slope = LEAKY_SLOPE
    return d


def conv_fwd(x, pk, out=None, bias=None, stats=None, out_f32=False, act_slope=0.0, addend=None, wimg=None, lat_ws=None):
    """x: bf16 [N,H,W,Ci] -> [N,Ho,Wo,Co]; 'same' for stride 1, top/left pad + 'valid' for stride 2.
    act_slope / addend / wimg: BatchNorm-folded inference - LeakyReLU(acc + bias) + residual from pre-scaled weights.
    lat_ws: the caller's LatencyWorkspace (one per model and stream); without one the latency form runs without K ranges."""
    N, H, W, Ci = x.shape
    assert Ci == pk.ci and x.dtype == torch.bfloat16
    Ho, Wo = (H // 2, W // 2) if pk.s == 2 else (H, W)
    if out is None:
        out = torch.empty(N, Ho, Wo, pk.co, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    dh, dw = taps_fwd(pk.k)
    d = _desc(x, pk.fwd if wimg is None else wimg, out, N, H, W, Ci, Ho, Wo, Ho, Wo, pk.co, pk.s, 1, (0, 0), dh, dw,
              pk.fwd_kpad, pk.fwd_copad, bias=bias, stats=stats, dst_f32=out_f32, act_slope=act_slope, addend=addend)
    lat = latency_plan(N * Ho * Wo, pk.fwd_copad, pk.fwd_kpad, len(dh), Ci) if (stats is None and not out_f32 and LATENCY and not d.form) else 0
    if lat:
        # a few thousand pixels (small-batch inference): (tile, K range) blocks with everything in flight, ranges added in-kernel
        if lat > 1 and lat_ws is None:
            lat = 1
        d.latency, d.splitk = 1, lat
        if lat > 1:
            tiles = (pk.fwd_copad // 128) * -(-(N * Ho * Wo) // 64)
            d.partial, d.partial_bytes = lat_ws.get(tiles, lat)
    _launch_gemm(d, "conv_fwd")
    return out


# Latency form (mgd_conv_desc.latency; MGD_LATENCY=0 turns it off): launches of at most LAT_TILES tiles of 128 channels x 64
# pixels - a 608 x 608 forward at batch 1 - 2.  K ranges: as many as bring the launch to about LAT_BLOCKS blocks (one per
# CU and round) while every block keeps at least two K-steps.
LATENCY = os.environ.get("MGD_LATENCY", "1") == "1"
LAT_TILES = int(os.environ.get("MGD_LAT_TILES", "256"))
LAT_BLOCKS = int(os.environ.get("MGD_LAT_BLOCKS", "256"))
LAT_RANGES = int(os.environ.get("MGD_LAT_RANGES", "4"))
LAT_MIN_STEPS = int(os.environ.get("MGD_LAT_MIN_STEPS", "32"))


def latency_plan(M, co_pad, k_pad, ntaps, ci):
    """0: regular dispatch; S >= 1: the latency form with S K ranges."""
    if co_pad % 128 or not (ntaps == 1 or ci % 64 == 0):
        return 0
    tiles, nk = (co_pad // 128) * -(-M // 64), k_pad // 64
    if tiles > LAT_TILES:
        return 0
    if nk < LAT_MIN_STEPS:
        return 1
    return max(1, min(nk // 4, LAT_RANGES, LAT_BLOCKS // tiles))


class LatencyWorkspace:
    """Caller-owned workspace of the latency form's K ranges: uncached device memory (mgd_uncached_alloc) holding the tile
    tickets (zero between launches) and the fp32 partial tiles.  The library keeps no buffer of its own: every model instance
    owns one workspace per stream it runs on (engine.Network.latency_workspace), so concurrent forwards never share tickets.
    Sized for 1024 (tile, range) blocks up front - the largest launch latency_plan() admits - and never re-allocated."""

    BLOCKS = 1024

    def __init__(self, device):
        self.device = torch.device(device)
        lib = L.load()
        self.nbytes = int(lib.mgd_latency_workspace_size(self.BLOCKS, 1))
        p = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(lib.mgd_uncached_alloc(C.c_int64(self.nbytes), C.byref(p)), "uncached_alloc")
        self.ptr = p.value

    def get(self, tiles, ranges):
        need = int(L.load().mgd_latency_workspace_size(tiles, ranges))
        if need > self.nbytes:
            raise L.MgdError(f"latency workspace: {tiles} tiles x {ranges} ranges need {need} bytes, have {self.nbytes}")
        return self.ptr, self.nbytes

    def tickets(self):
        """Test hook: the 4096 tickets after the current stream has drained."""
        out = (C.c_uint * 4096)()
        L.check(L.load().mgd_latency_tickets(C.c_void_p(self.ptr), out, L.stream_ptr()), "latency_tickets")
        return np.frombuffer(out, dtype=np.uint32).copy()

    def close(self):
        if getattr(self, "ptr", None):
            with torch.cuda.device(self.device):
                L.load().mgd_uncached_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def conv_dgrad(dy, pk, in_hw, out=None, addend=None, bnred=None):
    """dy: bf16 [N,Ho,Wo,Co] -> dx bf16 [N,H,W,Ci] (+ addend).  bnred = (y, scale, shift, mean, invstd, sums) of
    the BatchNorm layer that produced the forward input: its backward reduction is fused into the epilogue."""
    N, Ho, Wo, Co = dy.shape
    H, W = in_hw
    assert Co == pk.co and dy.dtype == torch.bfloat16
    if out is None:
        out = torch.empty(N, H, W, pk.ci, dtype=torch.bfloat16, device=dy.device)
    lib = L.load()
    if pk.s == 1:
        img, kp, cp, _, _ = pk.dgrad[0]
        dh, dw = taps_fwd(pk.k)
        d = _desc(dy, img, out, N, Ho, Wo, Co, H, W, H, W, pk.ci, 1, 1, (0, 0), dh, dw, kp, cp, addend=addend,
                  bnred=bnred)
        _launch_gemm(d, "conv_dgrad")
    elif pk.k == 3 and pk.ci == 32 and pk.co == 64 and _S2_PATCH:
        # the first down-sampling layer: all four parity classes in one patch-form launch (dy read once)
        d = L.DgradS2Desc()
        d.dy, d.dx = dy.data_ptr(), out.data_ptr()
        d.addend = addend.data_ptr() if addend is not None else None
        for c, (img, kp, cp, _, _) in enumerate(pk.dgrad):
            d.wpk[c], d.K_pad[c] = img.data_ptr(), kp
        d.N, d.Ho, d.Wo, d.Co, d.H, d.W, d.Ci = N, Ho, Wo, Co, H, W, pk.ci
        d.stats_replicas = STATS_REPLICAS
        if bnred is not None:
            y, sc, sh, mu, iv, sums = bnred
            d.bn_y, d.bn_scale, d.bn_shift = y.data_ptr(), sc.data_ptr(), sh.data_ptr()
            d.bn_mean, d.bn_invstd, d.bn_sums = mu.data_ptr(), iv.data_ptr(), sums.data_ptr()
            d.bn_slope = LEAKY_SLOPE
        L.check(lib.mgd_conv_dgrad_s2_patch(C.byref(d), L.stream_ptr()), "conv_dgrad_s2_patch")
    else:
        descs = [_desc(dy, img, out, N, Ho, Wo, Co, H // 2, W // 2, H, W, pk.ci, 1, 2, (ph, pw),
                       [t[0] for t in tp], [t[1] for t in tp], kp, cp, addend=addend, bnred=bnred)
                 for img, kp, cp, _, (ph, pw, tp) in pk.dgrad]
        if S2_CLASSES and descs[0].Co_pad % 128 == 0 and Co % 64 == 0 and not CONV_FORM:
            # the four output-parity classes in ONE launch (mgd_conv_gather_gemm_classes)
            if S2_CLASSES_ARG:
                for dd in descs:
                    dd.form, dd.form_arg = 8, S2_CLASSES_ARG
            arr = (L.ConvDesc * 4)(*descs)
            if PROFILE is None:
                L.check(lib.mgd_conv_gather_gemm_classes(arr, 4, L.stream_ptr()), "conv_dgrad_s2")
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(lib.mgd_conv_gather_gemm_classes(arr, 4, L.stream_ptr()), "conv_dgrad_s2")
                e1.record()
                d0 = descs[0]
                fl = sum(2.0 * d.N * d.Hg * d.Wg * d.ntaps * d.Ci * d.Co for d in descs)
                nd = d0.N * d0.Hd * d0.Wd * d0.Co
                by = 2.0 * d0.N * d0.Hs * d0.Ws * d0.Ci + sum(2.0 * d.Co_pad * d.K_pad for d in descs) + 2.0 * nd * (1 + (addend is not None) + (bnred is not None))
                PROFILE.append((e0, e1, fl, lib.mgd_last_kernel().decode(), "conv_dgrad_s2", by))
        else:
            for d in descs:
                _launch_gemm(d, "conv_dgrad_s2")
    return out


_S2_PATCH = os.environ.get("MGD_S2_PATCH", "1") != "0"
# (stride-2 data gradient with 64 output channels: padding its images to the 128-channel form so that the classes run as one
#  launch measured 173 us against 155 for the four 64-channel launches - not done)
S2_CLASSES = os.environ.get("MGD_S2_CLASSES", "1") != "0"      # stride-2 data gradient: the four parity classes in one launch
S2_CLASSES_ARG = int(os.environ.get("MGD_S2_CLASSES_ARG", "0"))   # measurement: 256 = tap-major K order, 4 = tile-major blocks
_WGRAD_BLOCKS = int(os.environ.get("MGD_WGRAD_BLOCKS", "0"))      # 0: by tile shape
# kernel form of the weight-gradient launches (mgd_wgrad_desc.form / form_arg; 0 = the library's dispatch): tests and tools
WGRAD_ROW_FORM = os.environ.get("MGD_WGRAD_ROW", "1") == "1"      # engine: give the kernel-row form its slab workspace (0: never dispatched)
WGRAD_FORM = int(os.environ.get("MGD_WGRAD_FORM", "0"))
WGRAD_FORM_ARG = int(os.environ.get("MGD_WGRAD_FORM_ARG", "0"))
# Blocks of a kernel-row weight-gradient launch under the library's own dispatch.  One of its blocks holds every register of a CU,
# and in the step the weight gradients run on a side stream UNDER the data-gradient / BatchNorm chain: with 256 blocks the chain's
# kernels find no CU (same-box A/B of bench.py: 12.38 ms per step at 256 blocks, 12.12 at 192, 12.12 at 128, 12.16 at 96;
# 12.46 without the form).  Alone a launch is fastest at 256 (tools/bench_wgrad_forms.py sets it).
WGRAD_ROW_BLOCKS = int(os.environ.get("MGD_WGRAD_ROW_BLOCKS", "128"))


def wgrad_splits(P, co, ci, T, target_blocks=None):
    """Split-K factor of the per-tap weight-gradient kernel.  Cost model fitted to tools/bench_wgrad_splits.py (it
    reproduces the measured optimum on the layers of the graph to within a few percent): blocks = tiles * splits run three
    per CU (128 x 64 tiles, 48 KB of LDS each; two per CU for the other tile shapes), so ceil(blocks / slots) rounds of
    K-steps at 1.63 us each, plus the fp32-atomic epilogue, which is bound by the memory-side atomic rate (64 KB per
    128 x 128 block at 1.3 TB/s = 0.05 us per block, not overlapped)."""
    bco = 128 if co > 64 else (64 if co > 32 else 32)
    bci = 128 if ci > 64 else (64 if ci > 32 else 32)
    big = co > 64 and ci > 64
    if big:
        bco, bci = 128, 64
    tiles = -(-co // bco) * -(-ci // bci) * T
    if target_blocks or T == 1:        # 1x1: ~1 block per CU, measured (short blocks, the epilogue dominates)
        target_blocks = target_blocks or 256
        return max(1, min(int(target_blocks / tiles + 0.5), -(-P // 256)))
    slots = _WGRAD_BLOCKS if _WGRAD_BLOCKS else (768 if big else 512)
    atom = 0.05 * (bco * bci) / (128.0 * 128.0)
    best, best_cost = 1, None
    for sp in range(1, max(1, min(256, P // 256)) + 1):
        blocks = tiles * sp
        steps = -(-(-(-P // sp)) // 64)
        cost = -(-blocks // slots) * steps * 1.63 + blocks * atom
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = sp, cost
    return best


def conv_wgrad(x, dy, dw, k, s, splits=None, ws=None, row_blocks=None):
    """dw (fp32 [Co, k*k, Ci]) += x (*) dy.  ws: optional fp32 workspace tensor of the caller (one per stream) for the
    kernel-row form's per-split slabs (mgd_wgrad_desc.partial).  row_blocks: block cap of a kernel-row launch (None:
    WGRAD_ROW_BLOCKS, the in-step value; 0: one block per CU, the fastest for a launch running alone)."""
    N, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    d = L.WgradDesc()
    d.src, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    d.N, d.Hs, d.Ws, d.Ci, d.Hg, d.Wg, d.Co = N, H, W, Ci, Ho, Wo, Co
    d.in_stride = s
    dh, dwo = taps_fwd(k)
    d.ntaps = len(dh)
    for i, (a, b) in enumerate(zip(dh, dwo)):
        d.dh[i], d.dw_off[i] = a, b
    d.splits = splits if splits is not None else wgrad_splits(N * Ho * Wo, Co, Ci, k * k)
    d.form, d.form_arg = WGRAD_FORM, (WGRAD_FORM_ARG if WGRAD_FORM else (WGRAD_ROW_BLOCKS if row_blocks is None else row_blocks))
    if ws is not None:
        d.partial, d.partial_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    if PROFILE is None:
        L.check(L.load().mgd_conv_wgrad(C.byref(d), L.stream_ptr()), "conv_wgrad")
        return dw
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                      # on the stream the launch goes to (the weight-gradient side stream in the engine)
    L.check(L.load().mgd_conv_wgrad(C.byref(d), L.stream_ptr()), "conv_wgrad")
    e1.record()
    variant = L.load().mgd_last_kernel().decode()
    PROFILE.append((e0, e1, 2.0 * N * Ho * Wo * k * k * Ci * Co, variant, "conv_wgrad",
                    2.0 * x.numel() + 2.0 * dy.numel() + 4.0 * Co * k * k * Ci))          # x + dy read once, fp32 dW written once
    return dw


def stem_fwd(image, w, out=None, stats=None):
    N, H, W, _ = image.shape
    if out is None:
        out = torch.empty(N, H, W, 32, dtype=torch.bfloat16, device=image.device)
    L.check(L.load().mgd_stem_fwd(L.ptr(image), L.ptr(w), L.ptr(out), L.ptr(stats),
                                  STATS_REPLICAS if stats is not None else 0, N, H, W, L.stream_ptr()), "stem_fwd")
    return out


def stem_fwd_act(image, w, bias, act_slope, out=None):
    """BatchNorm-folded inference stem: LeakyReLU(conv(image, w) + bias), w pre-scaled, bias = the BN shift."""
    N, H, W, _ = image.shape
    if out is None:
        out = torch.empty(N, H, W, 32, dtype=torch.bfloat16, device=image.device)
    L.check(L.load().mgd_stem_fwd_act(L.ptr(image), L.ptr(w), L.ptr(bias), float(act_slope), L.ptr(out), N, H, W, L.stream_ptr()),
            "stem_fwd_act")
    return out


def stem_wgrad(image, dy, dw):
    N, H, W, _ = image.shape
    L.check(L.load().mgd_stem_wgrad(L.ptr(image), L.ptr(dy), L.ptr(dw), N, H, W, L.stream_ptr()), "stem_wgrad")
    return dw


def stem_wgrad_bn(image, da, y, scale, shift, smean, sinv, sums, dgamma, dbeta, dw):
    """Stem weight gradient with the stem's BN + LeakyReLU backward fused in (dy is never materialised); `sums` must hold
    the reduction the producer of `da` accumulated (conv_dgrad(..., bnred=...))."""
    N, H, W, _ = image.shape
    L.check(L.load().mgd_stem_wgrad_bn(L.ptr(image), L.ptr(da), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(smean),
                                       L.ptr(sinv), L.ptr(sums), STATS_REPLICAS, L.ptr(dgamma), L.ptr(dbeta),
                                       C.c_float(LEAKY_SLOPE), L.ptr(dw), N, H, W, L.stream_ptr()), "stem_wgrad_bn")
    return dw


# ------------------------------------------------------------------------------------------- BN / act
def bn_finalize(stats, count, gamma, beta, mm, mv, scale, shift, smean, sinv, training=True):
    Cn = gamma.numel()
    L.check(L.load().mgd_bn_finalize(L.ptr(stats), STATS_REPLICAS, Cn, C.c_float(count), L.ptr(gamma), L.ptr(beta),
                                     L.ptr(mm), L.ptr(mv), L.ptr(scale), L.ptr(shift), L.ptr(smean), L.ptr(sinv),
                                     C.c_float(BN_EPS), C.c_float(BN_MOMENTUM), int(training), L.stream_ptr()),
            "bn_finalize")


def bn_act_fwd(y, scale, shift, out, residual=None):
    Cn = y.shape[-1]
    P = y.numel() // Cn
    L.check(L.load().mgd_bn_act_fwd(L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(residual), L.ptr(out),
                                    C.c_int64(P), Cn, C.c_float(LEAKY_SLOPE), L.stream_ptr()), "bn_act_fwd")
    return out


def bn_act_fwd_fused(stats, count, gamma, beta, mm, mv, scale, shift, smean, sinv, y, out, residual=None,
                     training=True):
    """bn_finalize + bn_act_fwd in one launch."""
    Cn = y.shape[-1]
    P = y.numel() // Cn
    L.check(L.load().mgd_bn_act_fwd_fused(L.ptr(stats), STATS_REPLICAS, C.c_float(count), L.ptr(gamma), L.ptr(beta),
                                          L.ptr(mm), L.ptr(mv), L.ptr(scale), L.ptr(shift), L.ptr(smean),
                                          L.ptr(sinv), C.c_float(BN_EPS), C.c_float(BN_MOMENTUM), int(training),
                                          L.ptr(y), L.ptr(residual), L.ptr(out), C.c_int64(P), Cn,
                                          C.c_float(LEAKY_SLOPE), L.stream_ptr()), "bn_act_fwd_fused")
    return out


def bn_act_bwd(da, y, scale, shift, smean, sinv, sums, dgamma, dbeta, dy, frozen=False, reduced=False):
    """sums: fp32 [(R+1)*2*C] zeroed scratch.  Writes dy, accumulates dgamma/dbeta.  reduced=True: the sums were
    already accumulated by the producer of `da` (conv_dgrad(..., bnred=...))."""
    Cn = y.shape[-1]
    P = y.numel() // Cn
    lib = L.load()
    if not frozen and not reduced:
        L.check(lib.mgd_bn_act_bwd_reduce(L.ptr(da), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(smean),
                                          L.ptr(sinv), L.ptr(sums), STATS_REPLICAS, C.c_int64(P), Cn,
                                          C.c_float(LEAKY_SLOPE), L.stream_ptr()), "bn_act_bwd_reduce")
    L.check(lib.mgd_bn_act_bwd_apply(L.ptr(da), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(smean), L.ptr(sinv),
                                     L.ptr(sums), STATS_REPLICAS, L.ptr(dgamma), L.ptr(dbeta), L.ptr(dy),
                                     C.c_int64(P), Cn, C.c_float(LEAKY_SLOPE), int(frozen), L.stream_ptr()),
            "bn_act_bwd_apply")
    return dy


def upsample_concat_fwd(u, skip, out):
    N, h, w, Cu = u.shape
    L.check(L.load().mgd_upsample_concat_fwd(L.ptr(u), L.ptr(skip), L.ptr(out), N, h, w, Cu, skip.shape[-1],
                                             L.stream_ptr()), "upsample_concat_fwd")
    return out


def upsample_concat_bwd(dout, du, dskip):
    N, h, w, Cu = du.shape
    L.check(L.load().mgd_upsample_concat_bwd(L.ptr(dout), L.ptr(du), L.ptr(dskip), N, h, w, Cu, dskip.shape[-1],
                                             L.stream_ptr()), "upsample_concat_bwd")


def bias_grad(dy, dbias):
    Cn = dy.shape[-1]
    L.check(L.load().mgd_bias_grad(L.ptr(dy), L.ptr(dbias), C.c_int64(dy.numel() // Cn), Cn, L.stream_ptr()),
            "bias_grad")


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-7, grad_scale=1.0, weight_decay=0.0):
    L.check(L.load().mgd_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), C.c_int64(p.numel()), C.c_float(lr),
                                   C.c_float(b1), C.c_float(b2), C.c_float(eps), int(step), C.c_float(grad_scale),
                                   C.c_float(weight_decay), L.stream_ptr()), "adam")


def adam_step_dev(p, g, m, v, hyper, b1=0.9, b2=0.999, eps=1e-7, grad_scale=1.0):
    """hyper: device fp32 [2] = (lr_t, lr*weight_decay); see mgd_adam_step_dev."""
    L.check(L.load().mgd_adam_step_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), C.c_int64(p.numel()), L.ptr(hyper),
                                       C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(grad_scale),
                                       L.stream_ptr()), "adam_dev")


def sgd_step(p, g, mom, lr, momentum=0.937, nesterov=True, grad_scale=1.0):
    L.check(L.load().mgd_sgd_step(L.ptr(p), L.ptr(g), L.ptr(mom), C.c_int64(p.numel()), C.c_float(lr),
                                  C.c_float(momentum), int(nesterov), C.c_float(grad_scale), L.stream_ptr()), "sgd")


# ------------------------------------------------------------------------------------------- targets
def _anchor_array(anchors):
    a = np.stack([np.asarray(x, dtype=np.float32) for x in anchors], 0)   # [L, A, 2]
    return a


def build_targets(boxes, input_shape, anchors, num_classes, grid_shapes=None, mode=0, return_assignment=False, out=None):
    """boxes: fp32 CUDA [B, M, 5] -> list of fp32 [B, gh, gw, 5+A+C].  out: (ys, workspace) of an earlier call of the same
    shape - the caller keeps them (a training step's targets always land in the same buffers, train_step.TrainStep)."""
    L.require_gpu()
    lib = L.load()
    B, M, _ = boxes.shape
    a = _anchor_array(anchors)
    nl, A = a.shape[0], a.shape[1]
    H, W = int(input_shape[0]), int(input_shape[1])
    if grid_shapes is None:
        grid_shapes = [(H // s, W // s) for s in (32, 16, 8)][:nl]
    ghw = (C.c_int32 * (2 * nl))(*[int(v) for g in grid_shapes for v in g])
    F = 5 + A + num_classes
    need = lib.mgd_build_targets_workspace_size(B, M, nl, ghw)
    if out is not None:
        ys, ws = out
        assert len(ys) == nl and ws.numel() >= need and all(tuple(y.shape) == (B, int(g[0]), int(g[1]), F) for y, g in zip(ys, grid_shapes))
    else:
        ys = [torch.empty(B, int(g[0]), int(g[1]), F, dtype=torch.float32, device=boxes.device) for g in grid_shapes]
        ws = torch.empty(need, dtype=torch.uint8, device=boxes.device)
    assign = torch.empty(B, M, 4, dtype=torch.int32, device=boxes.device) if return_assignment else None
    yp = (C.c_void_p * nl)(*[y.data_ptr() for y in ys])
    af = (C.c_float * a.size)(*[float(v) for v in a.reshape(-1)])      # (a ctypes array: a recorded plan copies it by size)
    assert boxes.is_contiguous()
    L.check(lib.mgd_build_targets(L.ptr(boxes), B, M, af, nl, A,
                                  num_classes, H, W, ghw, yp, L.ptr(assign), mode, L.ptr(ws), C.c_size_t(need),
                                  L.stream_ptr()), "build_targets")
    if out is not None:
        return ys
    if return_assignment:
        return ys, assign
    return ys


# ------------------------------------------------------------------------------------------- loss
def make_loss_cfg(anchors, num_classes, input_shape, batch, grid_shapes, **kw):
    a = _anchor_array(anchors)
    cfg = L.LossCfg()
    cfg.L, cfg.A, cfg.C, cfg.B = a.shape[0], a.shape[1], num_classes, batch
    cfg.in_h, cfg.in_w = int(input_shape[0]), int(input_shape[1])
    for l, g in enumerate(grid_shapes):
        cfg.grid_h[l], cfg.grid_w[l] = int(g[0]), int(g[1])
        for j in range(a.shape[1]):
            cfg.anchors[l][j][0], cfg.anchors[l][j][1] = float(a[l, j, 0]), float(a[l, j, 1])
    norm = kw.get("loss_normalization") or ["batch"]
    if not isinstance(norm, list):
        norm = [norm]
    cfg.norm_batch, cfg.norm_positives, cfg.norm_grid = norm.count("batch"), norm.count("positives"), norm.count("grid")
    cfg.ignore_thresh = kw.get("ignore_thresh", 0.5)
    cfg.label_smoothing = kw.get("label_smoothing", 0.0)
    cfg.loss_option = kw.get("loss_option", 2)
    for k, dflt in (("coord_scale", 1.0), ("object_scale", 1.0), ("no_object_scale", 1.0), ("class_scale", 1.0),
                    ("anchor_scale", 1.0), ("iou_objectness_power", 1.0), ("trainable_nms_power", 2.0),
                    ("consensus_iou_power", 1.5), ("consensus_min_iou", 1e-3), ("consensus_coord_scale", 0.5),
                    ("consensus_obj_scale", 0.5), ("consensus_class_scale", 0.3),
                    ("consensus_center_tolerance", 1e-4), ("focal_alpha", 0.25), ("focal_gamma", 2.0)):
        setattr(cfg, k, float(kw.get(k, dflt)))
    cfg.iou_objectness_ratio = float(min(max(kw.get("iou_objectness_ratio", 1.0), 0.0), 1.0))
    cfg.trainable_nms_weight = float(kw.get("trainable_nms_weight", 0.0))
    cfg.use_iou_aware_objectness = int(bool(kw.get("use_iou_aware_objectness", False)))
    cfg.use_consensus_loss = int(bool(kw.get("use_consensus_loss", False)))
    cfg.consensus_stop_gradient = int(bool(kw.get("consensus_stop_gradient", True)))
    cfg.use_focal_loss = int(bool(kw.get("use_focal_loss", False)))
    cfg.grad_out_scale = float(kw.get("grad_out_scale", 1.0))
    # loss_option 3: the first flag set wins, in the reference's order (multigrid_loss.py:353-364)
    iou = 0
    if cfg.loss_option == 3:
        iou = 1 if kw.get("use_giou_loss") else 2 if kw.get("use_diou_loss") else 3 if kw.get("use_ciou_loss") else 0
    cfg.iou_loss = iou
    compat = kw.get("compat", "tf_ref")
    if compat not in ("tf_ref", "fixed"):
        raise ValueError(f"compat must be 'tf_ref' or 'fixed', got {compat!r}")
    cfg.iou_compat = cfg.softmax_compat = 0 if compat == "tf_ref" else 1
    cfg.use_softmax_focal = int(bool(kw.get("use_softmax_loss", False)))
    if cfg.use_softmax_focal:
        cfg.use_focal_loss = 0          # use_softmax_loss takes precedence (multigrid_loss.py:400-403)
    return cfg


class LossRunner:
    """Holds the workspace; `run` returns (components[8] device tensor, grads)."""

    def __init__(self, cfg, device, class_weights=None):
        self.cfg = cfg
        lib = L.load()
        self.need = lib.mgd_loss_workspace_size(C.byref(cfg))
        self.ws = torch.empty(self.need, dtype=torch.uint8, device=device)
        self.components = torch.zeros(8, dtype=torch.float32, device=device)
        self.class_weights = None if class_weights is None else torch.as_tensor(
            class_weights, dtype=torch.float32, device=device).contiguous()

    def run(self, y_true, y_pred, grad_f32=None, grad_bf16=None):
        nl = self.cfg.L
        yp = (C.c_void_p * nl)(*[t.data_ptr() for t in y_pred])
        yt = (C.c_void_p * nl)(*[t.data_ptr() for t in y_true])
        gf = (C.c_void_p * nl)(*[t.data_ptr() for t in grad_f32]) if grad_f32 is not None else None
        gb = (C.c_void_p * nl)(*[t.data_ptr() for t in grad_bf16]) if grad_bf16 is not None else None
        L.check(L.load().mgd_loss_fwd_bwd(C.byref(self.cfg), yp, yt, L.ptr(self.class_weights), gf, gb,
                                          L.ptr(self.components), L.ptr(self.ws), C.c_size_t(self.need),
                                          L.stream_ptr()), "loss_fwd_bwd")
        return self.components


# ------------------------------------------------------------------------------------------- decode / nms
def make_decode_cfg(anchors, num_classes, input_shape, batch, grid_shapes, confidence, use_softmax=True,
                    rescore=True, cap=None, tag_scale=False):
    a = _anchor_array(anchors)
    cfg = L.DecodeCfg()
    cfg.L, cfg.A, cfg.C, cfg.B = a.shape[0], a.shape[1], num_classes, batch
    cfg.in_h, cfg.in_w = int(input_shape[0]), int(input_shape[1])
    tot = 0
    for l, g in enumerate(grid_shapes):
        cfg.grid_h[l], cfg.grid_w[l] = int(g[0]), int(g[1])
        tot += int(g[0]) * int(g[1])
        for j in range(a.shape[1]):
            cfg.anchors[l][j][0], cfg.anchors[l][j][1] = float(a[l, j, 0]), float(a[l, j, 1])
    cfg.use_softmax, cfg.rescore = int(use_softmax), int(rescore)
    cfg.confidence = float(confidence)
    cfg.cap = int(cap if cap is not None else tot)
    cfg.tag_scale = int(bool(tag_scale))
    return cfg


def decode(cfg, y_pred, image_hw):
    """y_pred: list of fp32 CUDA [B,g,g,F]; image_hw: fp32 CUDA [B,2] -> (boxes[B,cap,4], scores, cls, count)."""
    lib = L.load()
    dev = y_pred[0].device
    need = lib.mgd_decode_workspace_size(C.byref(cfg))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    B, cap = cfg.B, cfg.cap
    # (uninitialised: the kernels write the count and zero every slot past it)
    boxes = torch.empty(B, cap, 4, dtype=torch.float32, device=dev)
    scores = torch.empty(B, cap, dtype=torch.float32, device=dev)
    cls = torch.empty(B, cap, dtype=torch.int32, device=dev)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    yp = (C.c_void_p * cfg.L)(*[t.data_ptr() for t in y_pred])
    L.check(lib.mgd_decode(C.byref(cfg), yp, L.ptr(image_hw), L.ptr(boxes), L.ptr(scores), L.ptr(cls), L.ptr(count),
                           L.ptr(ws), C.c_size_t(need), L.stream_ptr()), "decode")
    return boxes, scores, cls, count


NMS_METHODS = {"standard": 0, "cluster": 0, "iou": 0, "diou": 1, "soft": 2}


NMS_PER_SCALE = 0x100


def nms(boxes, scores, cls, count, image_hw, method="diou", threshold=0.5, max_boxes=100, return_xyxy=True,
        per_scale=False):
    """method: 'standard' / 'cluster' (IoU), 'diou', 'soft' (SoftNMS sigma 0.5, score threshold 1e-3), or 'wbf'
    (Weighted Boxes Fusion with iou_thr = threshold).  Batched: one block per image."""
    lib = L.load()
    B, cap = scores.shape
    dev = boxes.device
    wbf = method == "wbf"
    need = lib.mgd_wbf_workspace_size(B, cap) if wbf else lib.mgd_nms_workspace_size(B, cap)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ob = torch.empty(B, max_boxes, 4, dtype=torch.int32 if return_xyxy else torch.float32, device=dev)
    osc = torch.empty(B, max_boxes, dtype=torch.float32, device=dev)
    ocl = torch.empty(B, max_boxes, dtype=torch.int32, device=dev)
    ocn = torch.empty(B, dtype=torch.int32, device=dev)
    if wbf:
        L.check(lib.mgd_wbf(L.ptr(boxes), L.ptr(scores), L.ptr(cls), L.ptr(count), B, cap, C.c_float(threshold),
                            max_boxes, L.ptr(image_hw), int(return_xyxy), L.ptr(ob), L.ptr(osc), L.ptr(ocl), L.ptr(ocn),
                            L.ptr(ws), C.c_size_t(need), L.stream_ptr()), "wbf")
    else:
        L.check(lib.mgd_nms(L.ptr(boxes), L.ptr(scores), L.ptr(cls), L.ptr(count), B, cap,
                            NMS_METHODS[method] | (NMS_PER_SCALE if per_scale else 0),
                            C.c_float(threshold), max_boxes, L.ptr(image_hw), int(return_xyxy), L.ptr(ob), L.ptr(osc),
                            L.ptr(ocl), L.ptr(ocn), L.ptr(ws), C.c_size_t(need), L.stream_ptr()), "nms")
    return ob, osc, ocl, ocn
