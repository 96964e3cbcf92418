"""ctypes binding of libmgd_hip.so (the C-ABI declared in include/mgd_hip.h).

There is no CPU fallback: importing an op and calling it without the built library, or on a host
without a GPU, raises.  PyTorch is used only as the device allocator / stream provider.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmgd_hip.so")
# the diagnostic build (ablation / stamped instantiations, skeleton kernels; include/mgd_hip_diag.h): tools/ only, see use_diag()
DIAG_LIB_PATH = os.path.join(_HERE, "csrc", "libmgd_hip_diag.so")
_lib = None


class MgdError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("wpk", C.c_void_p), ("dst", C.c_void_p), ("bias", C.c_void_p),
        ("addend", C.c_void_p), ("stats", C.c_void_p),
        ("N", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("Ci", C.c_int32),
        ("Hg", C.c_int32), ("Wg", C.c_int32),
        ("Hd", C.c_int32), ("Wd", C.c_int32), ("Co", C.c_int32),
        ("in_stride", C.c_int32), ("out_stride", C.c_int32), ("out_off_h", C.c_int32), ("out_off_w", C.c_int32),
        ("ntaps", C.c_int32), ("dh", C.c_int32 * 9), ("dw", C.c_int32 * 9),
        ("K_pad", C.c_int32), ("Co_pad", C.c_int32), ("dst_f32", C.c_int32), ("stats_replicas", C.c_int32),
        ("bn_y", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_mean", C.c_void_p),
        ("bn_invstd", C.c_void_p), ("bn_sums", C.c_void_p), ("bn_slope", C.c_float), ("act_slope", C.c_float),
        ("splitk", C.c_int32), ("partial", C.c_void_p), ("partial_bytes", C.c_int64), ("latency", C.c_int32),
        ("form", C.c_int32), ("form_arg", C.c_int32),
    ]


class DgradS2Desc(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("wpk", C.c_void_p * 4), ("dx", C.c_void_p), ("addend", C.c_void_p),
        ("K_pad", C.c_int32 * 4),
        ("N", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Co", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("Ci", C.c_int32), ("stats_replicas", C.c_int32),
        ("bn_y", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_mean", C.c_void_p),
        ("bn_invstd", C.c_void_p), ("bn_sums", C.c_void_p), ("bn_slope", C.c_float),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p),
        ("N", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("Ci", C.c_int32),
        ("Hg", C.c_int32), ("Wg", C.c_int32), ("Co", C.c_int32),
        ("in_stride", C.c_int32), ("ntaps", C.c_int32), ("dh", C.c_int32 * 9), ("dw_off", C.c_int32 * 9),
        ("splits", C.c_int32), ("form", C.c_int32), ("form_arg", C.c_int32),
        ("partial", C.c_void_p), ("partial_bytes", C.c_int64),
    ]


class PackJob(C.Structure):
    _fields_ = [
        ("w", C.c_void_p), ("out", C.c_void_p),
        ("Co", C.c_int32), ("T", C.c_int32), ("Ci", C.c_int32), ("transpose", C.c_int32),
        ("ntaps_out", C.c_int32), ("rows_pad", C.c_int32), ("K_pad", C.c_int32), ("pad_", C.c_int32),
        ("srccode", C.c_uint64), ("begin", C.c_int64),
    ]


class LossCfg(C.Structure):
    _fields_ = [
        ("L", C.c_int32), ("A", C.c_int32), ("C", C.c_int32), ("B", C.c_int32),
        ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("grid_h", C.c_int32 * 4), ("grid_w", C.c_int32 * 4),
        ("anchors", (C.c_float * 2) * 8 * 4),
        ("ignore_thresh", C.c_float), ("label_smoothing", C.c_float),
        ("loss_option", C.c_int32),
        ("coord_scale", C.c_float), ("object_scale", C.c_float), ("no_object_scale", C.c_float),
        ("class_scale", C.c_float), ("anchor_scale", C.c_float),
        ("norm_batch", C.c_int32), ("norm_positives", C.c_int32), ("norm_grid", C.c_int32),
        ("use_iou_aware_objectness", C.c_int32),
        ("iou_objectness_power", C.c_float), ("iou_objectness_ratio", C.c_float),
        ("trainable_nms_weight", C.c_float), ("trainable_nms_power", C.c_float),
        ("use_consensus_loss", C.c_int32),
        ("consensus_iou_power", C.c_float), ("consensus_min_iou", C.c_float),
        ("consensus_coord_scale", C.c_float), ("consensus_obj_scale", C.c_float),
        ("consensus_class_scale", C.c_float), ("consensus_center_tolerance", C.c_float),
        ("consensus_stop_gradient", C.c_int32),
        ("use_focal_loss", C.c_int32), ("focal_alpha", C.c_float), ("focal_gamma", C.c_float),
        ("grad_out_scale", C.c_float),
        ("iou_loss", C.c_int32), ("iou_compat", C.c_int32),
        ("use_softmax_focal", C.c_int32), ("softmax_compat", C.c_int32),
    ]


class DecodeCfg(C.Structure):
    _fields_ = [
        ("L", C.c_int32), ("A", C.c_int32), ("C", C.c_int32), ("B", C.c_int32),
        ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("grid_h", C.c_int32 * 4), ("grid_w", C.c_int32 * 4),
        ("anchors", (C.c_float * 2) * 8 * 4),
        ("use_softmax", C.c_int32), ("rescore", C.c_int32),
        ("confidence", C.c_float), ("cap", C.c_int32), ("tag_scale", C.c_int32),
    ]


# every symbol include/mgd_hip.h declares
EXPORTS = [
    "mgd_last_error", "mgd_version", "mgd_last_kernel", "mgd_plan_create", "mgd_plan_destroy", "mgd_plan_size", "mgd_plan_add_call", "mgd_plan_add_wait", "mgd_plan_run", "mgd_memset_async", "mgd_latency_workspace_size", "mgd_uncached_alloc", "mgd_uncached_free", "mgd_latency_tickets", "mgd_conv_gather_gemm", "mgd_conv_gather_gemm_classes", "mgd_conv_dgrad_s2_patch", "mgd_conv_wgrad", "mgd_conv_wgrad_workspace_size", "mgd_stem_fwd", "mgd_stem_fwd_act", "mgd_stem_wgrad", "mgd_stem_wgrad_bn",
    "mgd_pack_weights", "mgd_pack_weights_batch", "mgd_stem_im2col", "mgd_bn_finalize", "mgd_bn_act_fwd", "mgd_bn_act_fwd_fused", "mgd_bn_act_bwd_reduce", "mgd_bn_act_bwd_apply",
    "mgd_upsample_concat_fwd", "mgd_upsample_concat_bwd", "mgd_bias_grad", "mgd_f32_to_bf16", "mgd_bf16_to_f32",
    "mgd_adam_step", "mgd_adam_step_dev", "mgd_sgd_step", "mgd_build_targets_workspace_size", "mgd_build_targets",
    "mgd_loss_workspace_size", "mgd_loss_fwd_bwd", "mgd_decode_workspace_size", "mgd_decode",
    "mgd_nms_workspace_size", "mgd_nms", "mgd_wbf_workspace_size", "mgd_wbf", "mgd_iou_matrix", "mgd_eval_match", "mgd_mosaic", "mgd_gridmask", "mgd_mixup",
    "mgd_comm_unique_id", "mgd_comm_init", "mgd_comm_allreduce_bucket", "mgd_comm_destroy",
    "mgd_letterbox_workspace_size", "mgd_letterbox_u8",
    "mgd_conv2d_f32_fwd", "mgd_conv2d_f32_dgrad", "mgd_conv2d_f32_wgrad", "mgd_bn_stats_f32", "mgd_bn_act_fwd_f32",
    "mgd_bn_act_bwd_f32", "mgd_upsample_concat_fwd_f32", "mgd_upsample_concat_bwd_f32", "mgd_bias_grad_f32",
]


# what libmgd_hip_diag.so exports on top (include/mgd_hip_diag.h)
DIAG_EXPORTS = ["mgd_diag_set_flags", "mgd_diag_flags_value", "mgd_debug_stamps", "mgd_debug_mfma_peak",
                "mgd_debug_wgrad_skeleton", "mgd_debug_gemm_skeleton", "mgd_debug_vmem_rate"]


# ---------------------------------------------------------------------------------------------- launch plans (csrc/plan.cpp)
RECORDER = None      # while set, load() returns a proxy that runs every entry point AND records it


class Plan:
    """A recorded sequence of C-ABI calls (mgd_plan_*), replayed by one call that holds no interpreter lock."""

    def __init__(self, handle, streams, nparams):
        self.handle, self.streams, self.nparams = handle, list(streams), nparams
        self._sarr = (C.c_void_p * max(1, len(streams)))(*[s.cuda_stream for s in streams])
        self._parr = (C.c_void_p * max(1, nparams))()
        self.size = int(load().mgd_plan_size(handle))

    def run(self, params=()):
        assert len(params) == self.nparams
        for i, t in enumerate(params):
            self._parr[i] = t.data_ptr()
        check(load().mgd_plan_run(self.handle, self._sarr, len(self.streams), self._parr, self.nparams), "plan_run")

    def __del__(self):
        try:
            load().mgd_plan_destroy(self.handle)
        except Exception:
            pass


class Recorder:
    """Records what goes through the library while it is active.  streams: the torch streams of the step in slot order (a call's
    stream argument is matched against them); params: tensors whose addresses change from run to run (the batch) - a call
    argument equal to one of these addresses is replayed from Plan.run(params)."""

    def __init__(self, streams, params=()):
        lib = load()
        h = C.c_void_p()
        check(lib.mgd_plan_create(C.byref(h)), "plan_create")
        self.handle = h
        self.streams = list(streams)
        self.slots = {s.cuda_stream: i for i, s in enumerate(self.streams)}
        self.params = {t.data_ptr(): i for i, t in enumerate(params)}
        self.nparams = len(params)
        self.error = None

    def _slot(self, stream):
        return self.slots[stream.cuda_stream]

    def wait(self, waiting, signalling):
        """`waiting` (torch stream) waits for what `signalling` holds at this point of the sequence."""
        check(load().mgd_plan_add_wait(self.handle, self._slot(waiting), self._slot(signalling)), "plan_add_wait")

    def add(self, name, args):
        import struct
        words, kinds, blob = [], [], bytearray()

        def put_blob(raw):
            while len(blob) % 16:
                blob.append(0)
            off = len(blob)
            blob.extend(raw)
            return off
        for a in args:
            if a is None:
                words.append(0); kinds.append(0)
            elif isinstance(a, bool) or isinstance(a, int):
                v = int(a)
                if v in self.params:
                    words.append(self.params[v]); kinds.append(5)
                else:
                    words.append(v); kinds.append(0)
            elif isinstance(a, float):
                words.append(struct.unpack("<q", struct.pack("<d", a))[0]); kinds.append(2)
            elif isinstance(a, C.c_float):
                words.append(struct.unpack("<I", struct.pack("<f", a.value))[0]); kinds.append(1)
            elif isinstance(a, C.c_double):
                words.append(struct.unpack("<q", struct.pack("<d", a.value))[0]); kinds.append(2)
            elif isinstance(a, C.c_void_p):
                v = a.value or 0
                if v in self.slots:
                    words.append(self.slots[v]); kinds.append(4)
                elif v in self.params:
                    words.append(self.params[v]); kinds.append(5)
                else:
                    words.append(v); kinds.append(0)
            elif isinstance(a, (C.Structure, C.Array)):
                words.append(put_blob(bytes(a))); kinds.append(3)
            elif type(a).__name__ == "CArgObject":              # C.byref(x)
                words.append(put_blob(bytes(a._obj))); kinds.append(3)
            elif isinstance(a, C._Pointer):                       # C.cast(device pointer, POINTER(c_float))
                words.append(C.cast(a, C.c_void_p).value or 0); kinds.append(0)
            elif isinstance(a, C._SimpleCData):
                words.append(int(a.value)); kinds.append(0)
            else:
                raise MgdError(f"plan: cannot record argument {a!r} of {name}")
        n = len(words)
        w = (C.c_int64 * max(1, n))(*[x if x < (1 << 63) else x - (1 << 64) for x in words])
        k = (C.c_uint8 * max(1, n))(*kinds)
        raw = bytes(blob)
        check(load_raw().mgd_plan_add_call(self.handle, name.encode(), w, k, n, raw if raw else None, len(raw)), f"plan_add_call({name})")

    def finish(self):
        return Plan(self.handle, self.streams, self.nparams)


class _RecProxy:
    """What load() returns while a Recorder is active: every entry point runs as usual and is recorded when it succeeds."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        raw = getattr(self._lib, name)
        if not name.startswith("mgd_") or name.startswith(("mgd_plan_", "mgd_last_", "mgd_version")) or name.endswith("_workspace_size"):
            return raw

        def call(*args):
            rc = raw(*args)
            rec = RECORDER
            if rec is not None and rc == 0:
                try:
                    rec.add(name, args)
                except MgdError as e:
                    rec.error = rec.error or str(e)
            return rc
        return call


def load_raw():
    load()
    return _lib


def use_diag():
    """Make this process run on the DIAGNOSTIC library (tools/ only; before the first load()).  It is the same code built with
    -DMGD_DIAG: kernels with ablation switches and stamps, never what a product run measures or ships."""
    global LIB_PATH, _lib
    if _lib is not None and LIB_PATH != DIAG_LIB_PATH:
        raise MgdError("use_diag() must come before the first use of the library")
    LIB_PATH = DIAG_LIB_PATH
    print(f"[multigriddet_amd] DIAGNOSTIC library in use: {DIAG_LIB_PATH}", flush=True)
    return load()


def load():
    """Load the shared library (no GPU needed for this; compute calls need one)."""
    global _lib
    if _lib is not None:
        return _lib if RECORDER is None else _RecProxy(_lib)
    if not os.path.exists(LIB_PATH):
        raise MgdError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C multigriddet_amd/csrc`. There is no CPU fallback for the multigriddet hot path.")
    lib = C.CDLL(LIB_PATH)
    lib.mgd_last_error.restype = C.c_char_p
    lib.mgd_last_kernel.restype = C.c_char_p
    lib.mgd_stem_fwd_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for name in ("mgd_build_targets_workspace_size", "mgd_loss_workspace_size", "mgd_decode_workspace_size",
                 "mgd_nms_workspace_size", "mgd_wbf_workspace_size", "mgd_letterbox_workspace_size"):
        getattr(lib, name).restype = C.c_size_t
    lib.mgd_latency_workspace_size.restype = C.c_int64
    lib.mgd_plan_add_call.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.c_int, C.c_char_p, C.c_int64]
    lib.mgd_plan_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int]
    lib.mgd_plan_add_wait.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.mgd_plan_destroy.argtypes = [C.c_void_p]
    lib.mgd_plan_size.argtypes = [C.c_void_p]
    lib.mgd_memset_async.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    lib.mgd_conv_wgrad_workspace_size.restype = C.c_int64
    lib.mgd_uncached_alloc.argtypes = [C.c_int64, C.POINTER(C.c_void_p)]
    lib.mgd_uncached_free.argtypes = [C.c_void_p]
    lib.mgd_latency_tickets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def require_gpu():
    if not torch.cuda.is_available():
        raise MgdError("multigriddet_amd needs an AMD GPU (gfx950); no CPU execution path exists for the hot path")


def check(rc, what=""):
    if rc != 0:
        raise MgdError(f"{what}: {load().mgd_last_error().decode()} (code {rc})")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device pointer must come from a contiguous CUDA tensor"
    return C.c_void_p(t.data_ptr())


def fptr(t):
    return C.cast(ptr(t), C.POINTER(C.c_float))
