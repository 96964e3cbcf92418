"""CPU: the C-ABI library loads and exports every symbol include/mgd_hip.h declares; host-side logic
(tap tables, padding rules, bucket tiling, layer specs) is consistent.  No compute calls."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _build_if_needed():
    from multigriddet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_header_symbols_exported():
    _lib = _build_if_needed()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "mgd_hip.h")).read()
    names = sorted(set(re.findall(r"\b(mgd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(names) == sorted(_lib.EXPORTS)
    assert lib.mgd_version() >= 1


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multigriddet_amd import _lib
    from multigriddet_amd.engine import Network
    with pytest.raises(_lib.MgdError):
        Network(80, 3, "cuda:0")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "multigriddet_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_layer_specs_match_oracle_graph():
    from multigriddet_amd.engine import conv_specs
    from oracle import model as om
    a, b = conv_specs(), om.layer_specs()
    assert len(a) == len(b) == 69
    for x, y in zip(a, b):
        assert (x["cin"], x["cout"], x["k"], x["s"], x["bn"]) == (y["cin"], y["cout"], y["k"], y["s"], y["bn"])


def test_stride2_dgrad_parity_classes_cover_all_taps():
    from multigriddet_amd.ops import taps_dgrad_s2
    seen = []
    for ph in range(2):
        for pw in range(2):
            for dh, dw, t in taps_dgrad_s2(ph, pw):
                assert dh in (0, 1) and dw in (0, 1)
                kh, kw = divmod(t, 3)
                assert (ph + 1 - kh) % 2 == 0 and (pw + 1 - kw) % 2 == 0
                seen.append(t)
    assert sorted(seen) == list(range(9))


def test_cout_padding_rule():
    from multigriddet_amd.ops import pad_cout
    for co in (32, 64, 88, 128, 176, 256, 352, 512, 704, 1024):
        p = pad_cout(co)
        assert p >= co and p % 32 == 0 and p / co <= 1.125 + 1e-9 or p == -(-co // 32) * 32


def test_bucket_tiling():
    from multigriddet_amd.dp import make_buckets
    from multigriddet_amd.engine import conv_specs
    offs, off = [], 0
    for sp in conv_specs():
        offs.append(off)
        off += sp["cout"] * sp["k"] ** 2 * sp["cin"] + sp["cout"] * (2 if sp["bn"] else 1)
    assert off == 44954760
    bk = make_buckets(offs, off, int(32e6 / 4))
    assert bk[0][2] == off and bk[-1][1] == 0 and bk[-1][0] == 0
    for (i0, b0, e0), (i1, b1, e1) in zip(bk, bk[1:]):
        assert e1 == b0 and i1 < i0
    assert sum(e - b for _, b, e in bk) == off


def test_argument_validation_returns_einval_with_message():
    """Bad arguments are rejected on the host before anything is launched (no GPU needed): MGD_EINVAL (-1) and a
    thread-local message naming the entry point - the C-ABI's error convention (include/mgd_hip.h)."""
    import ctypes as C
    _lib = _build_if_needed()
    lib = _lib.load()
    cases = [
        ("stem_wgrad_bn", lambda: lib.mgd_stem_wgrad_bn(None, None, None, None, None, None, None, None, 16, None, None,
                                                        C.c_float(0.1), None, 1, 8, 8, None)),
        ("stem_wgrad", lambda: lib.mgd_stem_wgrad(None, None, None, 1, 8, 8, None)),
        ("stem_fwd", lambda: lib.mgd_stem_fwd(None, None, None, None, 0, 1, 8, 8, None)),
        ("comm_init", lambda: lib.mgd_comm_init(None, 0, 1, None)),
        ("comm_init", lambda: lib.mgd_comm_init(C.byref(C.c_void_p()), 3, 2, (C.c_ubyte * 128)())),
        ("comm_allreduce_bucket", lambda: lib.mgd_comm_allreduce_bucket(None, None, C.c_int64(4), None)),
        ("comm_unique_id", lambda: lib.mgd_comm_unique_id(None)),
        ("pack_batch", lambda: lib.mgd_pack_weights_batch(None, 0, C.c_int64(0), None)),
        ("wgrad", lambda: lib.mgd_conv_wgrad(None, None)),
        ("conv", lambda: lib.mgd_conv_gather_gemm(None, None)),
    ]
    for name, call in cases:
        assert call() == -1, name
        assert name.encode() in lib.mgd_last_error(), (name, lib.mgd_last_error())
    assert lib.mgd_comm_destroy(None) == 0          # destroying nothing is not an error


def test_pack_weights_rejects_unaligned_k_for_fragment_order_images():
    """Images of 128-row tiles are written in MFMA-fragment order (blocks of 128 rows x 64 K): a K_pad that is not a
    multiple of 64 would write past the image, so the C-ABI refuses it on the host."""
    import ctypes as C
    _lib = _build_if_needed()
    lib = _lib.load()
    taps = (C.c_int32 * 9)(*range(9))
    fake = C.c_void_p(0x10000)                       # never dereferenced: validation fails first
    rc = lib.mgd_pack_weights(fake, fake, 128, 9, 8, 0, 9, taps, 128, 72, None)
    assert rc == -1 and b"multiple of 64" in lib.mgd_last_error(), lib.mgd_last_error()


def test_conv_rejects_tensors_beyond_32bit_offsets():
    """The gather-GEMM kernels address the source tensor with 32-bit byte offsets from a scalar base: a source of 4 GiB
    or more must be refused on the host (MGD_EINVAL), not silently wrapped (include/mgd_hip.h, mgd_conv_desc limits)."""
    import ctypes as C
    _lib = _build_if_needed()
    lib = _lib.load()
    d = _lib.ConvDesc()
    fake = 0x10000                                   # never dereferenced: validation fails first
    d.src, d.wpk, d.dst = fake, fake, fake
    d.N, d.Hs, d.Ws, d.Ci = 400, 304, 304, 64        # 400*304*304*64*2 B = 4.7 GB
    d.Hg, d.Wg, d.Hd, d.Wd, d.Co = 304, 304, 304, 304, 32
    d.in_stride, d.out_stride, d.ntaps = 1, 1, 1
    d.K_pad, d.Co_pad = 64, 32
    assert lib.mgd_conv_gather_gemm(C.byref(d), None) == -1
    assert b"32-bit byte addressing" in lib.mgd_last_error()


def test_bucket_cut_points_follow_layer_boundaries():
    """Gradient buckets are contiguous, tile the flat buffer from the END (the order backward finalises layers in) and
    only ever start at a layer boundary."""
    from multigriddet_amd.dp import make_buckets
    offs = [0, 10, 30, 70, 150, 310]
    bk = make_buckets(offs, 400, 100)
    assert bk[0][2] == 400 and bk[-1][1] == 0
    for (_, b, e), (_, b2, e2) in zip(bk, bk[1:]):
        assert e2 == b and b2 < e2
    assert all(b in offs for _, b, _ in bk)
    assert all(e - b >= 100 for _, b, e in bk[:-1])


def test_product_library_reads_no_environment_and_exports_no_diagnostics():
    """Dispatch of a C-ABI call depends on its arguments only: no getenv anywhere under csrc/, the product library exports no
    mgd_debug_* / mgd_diag_* symbol and include/mgd_hip.h declares none; the diagnostic twin (libmgd_hip_diag.so,
    include/mgd_hip_diag.h) exports everything the product does plus its own entry points."""
    import ctypes as C
    _lib = _build_if_needed()
    csrc = os.path.join(ROOT, "multigriddet_amd", "csrc")
    for dp, _, fs in os.walk(csrc):
        for f in fs:
            if f.endswith((".hip", ".cpp", ".hpp", ".h")):
                assert "getenv" not in open(os.path.join(dp, f)).read(), f
    hdr = open(os.path.join(ROOT, "include", "mgd_hip.h")).read()
    assert "mgd_debug" not in hdr and "mgd_diag" not in hdr
    lib = _lib.load()
    for s in _lib.DIAG_EXPORTS:
        assert not hasattr(lib, s), s
    diag = C.CDLL(_lib.DIAG_LIB_PATH)
    for s in _lib.EXPORTS + _lib.DIAG_EXPORTS:
        assert hasattr(diag, s), s
    dh = open(os.path.join(ROOT, "include", "mgd_hip_diag.h")).read()
    assert sorted(set(re.findall(r"\b(mgd_[a-z0-9_]+)\s*\(", dh))) == sorted(_lib.DIAG_EXPORTS)


def test_forced_kernel_forms_are_validated_on_the_host():
    """mgd_conv_desc.form / mgd_wgrad_desc.form: a form that cannot run the geometry is refused with MGD_EINVAL and a message,
    never silently replaced (include/mgd_hip.h)."""
    import ctypes as C
    _lib = _build_if_needed()
    lib = _lib.load()
    d = _lib.ConvDesc()
    fake = 0x10000                                   # never dereferenced: validation fails first
    d.src, d.wpk, d.dst = fake, fake, fake
    d.N, d.Hs, d.Ws, d.Ci = 1, 8, 8, 32
    d.Hg, d.Wg, d.Hd, d.Wd, d.Co = 8, 8, 8, 8, 64
    d.in_stride, d.out_stride, d.ntaps = 1, 1, 1
    d.K_pad, d.Co_pad = 64, 64
    for form, msg in ((12, b"thin-tile"), (10, b"thin-tile"), (6, b"thin-tile"), (77, b"thin-tile")):
        d.form = form
        assert lib.mgd_conv_gather_gemm(C.byref(d), None) == -1 and msg in lib.mgd_last_error(), (form, lib.mgd_last_error())
    d.Co, d.Co_pad, d.form, d.form_arg = 128, 128, 12, 0        # 256-channel tiles on a 128-channel image
    assert lib.mgd_conv_gather_gemm(C.byref(d), None) == -1 and b"phased" in lib.mgd_last_error()
    d.form, d.form_arg = 9, 5
    d.ntaps, d.Ci, d.K_pad = 9, 64, 576
    for t in range(9):
        d.dh[t], d.dw[t] = t // 3 - 1, t % 3 - 1
    assert lib.mgd_conv_gather_gemm(C.byref(d), None) == -1 and b"pixel tile" in lib.mgd_last_error()
    d.form, d.form_arg, d.splitk = 0, 0, 4           # K ranges without the latency form
    assert lib.mgd_conv_gather_gemm(C.byref(d), None) == -1 and b"latency" in lib.mgd_last_error()
    w = _lib.WgradDesc()
    w.src, w.dy, w.dw = fake, fake, fake
    w.N, w.Hs, w.Ws, w.Ci, w.Hg, w.Wg, w.Co = 1, 16, 16, 64, 8, 8, 128
    w.in_stride, w.ntaps, w.splits = 2, 9, 1
    for t in range(9):
        w.dh[t], w.dw_off[t] = t // 3 - 1, t % 3 - 1
    w.form = 4                                       # descriptor form on a stride-2 layer
    assert lib.mgd_conv_wgrad(C.byref(w), None) == -1 and b"descriptor-addressed" in lib.mgd_last_error()
    w.form = 9
    assert lib.mgd_conv_wgrad(C.byref(w), None) == -1 and b"unknown kernel form" in lib.mgd_last_error()
