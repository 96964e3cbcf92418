"""GPU: every single-GPU BASELINE.json config exercised AT ITS STATED SIZE.

config 2  608x608, batch 16 train step: the real `TrainStep` runs twice; after the second step (run at lr = 0 so
          that the packed weights still match the arena) the raw conv output `y`, the activated output `a` and the
          weight gradient `dw` of real backbone layer shapes (16x304^2x32->64 ... 16x19^2x512->1024, stride 1 and 2,
          1x1 and 3x3) are compared with a plain torch fp32 reference on the SAME bf16 operands taken from the arena -
          this is where the 32-bit offset arithmetic, the XCD remap and the split-K cost model run at full size.
config 3  multi-scale {320, 352, ..., 608} at batch 16 through one Network / TrainStep.
config 5  Mosaic / MixUp / GridMask at 608x608, batch 16 against the numpy oracle ("CSPDarknet53" does not exist in
          the reference - models/backbones/darknet.py:219-222 builds Darknet53 - so Darknet53 is what runs).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bench

pytestmark = pytest.mark.gpu

S, B = 608, 16
# (layer index, what it is) - engine.conv_specs() order
LAYERS = [
    (1, "32->64 3x3 s2 608->304"),
    (2, "64->32 1x1 @304"),
    (3, "32->64 3x3 @304"),
    (4, "64->128 3x3 s2 ->152"),
    (6, "64->128 3x3 @152"),
    (9, "128->256 3x3 s2 ->76"),
    (10, "256->128 1x1 @76"),
    (11, "128->256 3x3 @76"),
    (26, "256->512 3x3 s2 ->38"),
    (28, "256->512 3x3 @38"),
    (43, "512->1024 3x3 s2 ->19"),
    (45, "512->1024 3x3 @19"),
]


def _ref_conv(x_nhwc, w_ohwi, k, s):
    x = x_nhwc.permute(0, 3, 1, 2)
    co, T, ci = w_ohwi.shape
    w = w_ohwi.view(co, k, k, ci).permute(0, 3, 1, 2)
    if s == 2:
        y = F.conv2d(F.pad(x, (1, 0, 1, 0)), w, stride=2)
    else:
        y = F.conv2d(x, w, padding=k // 2)
    return y.permute(0, 2, 3, 1)


@pytest.fixture(scope="module")
def stepped():
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    dev = torch.device("cuda:0")
    net = Network(80, 3, dev, seed=0)
    ts = TrainStep(net, bench.coco_anchors(), 80, (S, S), B, lr=1e-4)
    img, bx = bench.synth_batch(0, B, S)
    img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
    l0 = float(ts.step(img, bx)[7])
    ts.lr = 0.0                       # Adam's update is lr_t * m / (sqrt(v) + eps): no change, packed images stay valid
    p_before = net.params.clone()
    l1 = float(ts.step(img, bx)[7])
    torch.cuda.synchronize()
    assert torch.equal(p_before, net.params)
    return net, ts, (l0, l1)


def test_config2_step_runs_and_trains(stepped):
    net, ts, (l0, l1) = stepped
    assert np.isfinite([l0, l1]).all()
    assert l1 < l0, (l0, l1)
    assert ts.step_count == 2
    g = net.grads
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0


@pytest.mark.parametrize("idx,what", LAYERS)
def test_config2_real_layer_shapes_vs_fp32_torch(stepped, idx, what):
    net, ts, _ = stepped
    A = net._last
    cv = net.layers[idx]
    x = A["fin"][idx]                                  # bf16 input activation of this conv (forward of step 2)
    y = A["y"][idx]
    assert x.shape[0] == B and y.shape[0] == B
    w = cv.w.detach().float().cpu()                    # fp32 master [Co, T, Ci]; the kernels see its bf16 rounding
    wb = w.to(torch.bfloat16).float()
    xc = x.float().cpu().requires_grad_(True)
    wr = wb.clone().requires_grad_(True)
    y_ref = _ref_conv(xc, wr, cv.k, cv.s)
    err = (y.float().cpu() - y_ref.detach()).abs().max().item()
    tol = 0.01 * y_ref.detach().abs().max().item() + 1e-3      # bf16 output rounding: 2^-8 relative
    assert err <= tol, f"{what}: y err {err} tol {tol}"
    # every image of the batch, last one included (the far end of the 32-bit offset range)
    e_last = (y[-1].float().cpu() - y_ref[-1].detach()).abs().max().item()
    assert e_last <= tol
    # a = LeakyReLU(BatchNorm(y)) (+ residual) with batch statistics of the bf16 y (models/layers.py:88-95)
    yf = y.float().cpu().view(-1, cv.cout)
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    z = (yf - mean) * torch.rsqrt(var + 1e-3) * cv.gamma.cpu() + cv.beta.cpu()
    a_ref = torch.where(z > 0, z, 0.1 * z).view(y.shape)
    if cv.role == "res2":
        a_ref = a_ref + A["fin"][idx - 1].float().cpu()       # residual = input of the block's 1x1
    a = A["a"][idx].float().cpu()
    ea = (a - a_ref).abs().max().item()
    assert ea <= 0.01 * a_ref.abs().max().item() + 2e-2, f"{what}: a err {ea}"
    # weight gradient of step 2 against autograd on the arena's own (x, dy)
    dy = A["scratch"][(("dy", idx), tuple(y.shape), torch.bfloat16)]
    y_ref.backward(dy.float().cpu())
    dw = cv.dw.detach().cpu()
    dw_ref = wr.grad
    e = (dw - dw_ref).abs().max().item()
    assert e <= 3e-3 * dw_ref.abs().max().item() + 1e-4, f"{what}: dw err {e} of {dw_ref.abs().max().item()}"
    cos = float((dw * dw_ref).sum() / (dw.norm() * dw_ref.norm() + 1e-30))
    assert cos > 0.9999, f"{what}: dw cosine {cos}"


def test_config2_data_gradient_of_real_shapes(stepped):
    """The data gradients of the same step: dx = dgrad(dy) for a stride-1 3x3 and a stride-2 layer at full size,
    against autograd on the arena's dy (before the residual-gradient addend is folded in: checked through a direct call)."""
    from multigriddet_amd import ops
    net, ts, _ = stepped
    A = net._last
    for idx in (11, 26, 4):
        cv = net.layers[idx]
        y = A["y"][idx]
        dy = A["scratch"][(("dy", idx), tuple(y.shape), torch.bfloat16)]
        x = A["fin"][idx]
        dx = ops.conv_dgrad(dy, cv.pk, (x.shape[1], x.shape[2]))
        torch.cuda.synchronize()
        xr = torch.zeros(x.shape, dtype=torch.float32).requires_grad_(True)
        wb = cv.w.detach().float().cpu().to(torch.bfloat16).float()
        _ref_conv(xr, wb, cv.k, cv.s).backward(dy.float().cpu())
        e = (dx.float().cpu() - xr.grad).abs().max().item()
        assert e <= 0.01 * xr.grad.abs().max().item() + 1e-6, (idx, e)


def test_config3_multiscale_320_to_608_batch16():
    """BASELINE config 3: S cycled over {320, 352, ..., 608} per step at batch 16 through ONE Network / TrainStep (weights,
    Adam state shared; arenas, loss configuration and target grids keyed by resolution and kept)."""
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    dev = torch.device("cuda:0")
    net = Network(80, 3, dev, seed=0)
    ts = TrainStep(net, bench.coco_anchors(), 80, (320, 320), B, lr=1e-4)
    sizes = list(range(320, 609, 32))
    assert sizes[-1] == 608 and len(sizes) == 10
    batches = {}
    for s in sizes:
        img, bx = bench.synth_batch(s, B, s)
        batches[s] = (torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev))
    first, last = {}, {}
    for rnd in range(3):
        for s in sizes:
            v = float(ts.step(*batches[s])[7])
            assert np.isfinite(v), (rnd, s, v)
            first.setdefault(s, v)
            last[s] = v
    torch.cuda.synchronize()
    assert ts.step_count == 3 * len(sizes)
    assert sorted(k[1] for k in net._arenas) == sizes            # one arena per resolution, kept
    better = sum(last[s] < first[s] for s in sizes)
    assert better >= 8, (first, last)                             # the shared weights train across resolutions
    assert torch.isfinite(net.params).all()


def _aug_batch(seed, nb, size, M=25, n=10):
    rng = np.random.default_rng(seed)
    img = (rng.random((nb, size, size, 3), dtype=np.float32) * 255).astype(np.float32)
    bx = np.zeros((nb, M, 5), np.float32)
    for b in range(nb):
        for t in range(int(rng.integers(0, n + 1))):
            w, h = rng.uniform(6, size / 2, 2)
            cx, cy = rng.uniform(w / 2, size - w / 2), rng.uniform(h / 2, size - h / 2)
            bx[b, t] = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.integers(0, 80)]
    return rng, img, bx


def test_config5_mosaic_mixup_gridmask_at_608_batch16():
    """BASELINE config 5's on-GPU augmentations at 608x608, batch 16 (reference data/generators.py:561-1282) against the
    numpy oracle with the same host draws: copied pixels and box lists exact, MixUp blend to fp32 rounding."""
    from multigriddet_amd.data import augment as aug
    from oracle import aug as oa
    rng, img, bx = _aug_batch(7, B, S)
    bx4 = np.concatenate([bx, np.zeros((B, 3 * bx.shape[1], 5), np.float32)], 1)      # 4x capacity (Mosaic contract)
    src, crop = aug.draw_mosaic(rng, B, S)
    gi, gb = aug.mosaic(torch.from_numpy(img).cuda(), torch.from_numpy(bx4).cuda(), src, crop)
    ri, rb = oa.mosaic(img, bx4, src, crop)
    torch.cuda.synchronize()
    assert np.array_equal(gi.cpu().numpy(), ri)
    assert np.array_equal(gb.cpu().numpy(), rb)
    assert (rb[..., 2] > rb[..., 0]).sum() > B
    # GridMask on the mosaic output (the pipeline's order)
    apply, par = aug.draw_gridmask(rng, B, S, prob=0.5)
    apply[0], apply[1] = 1, 0
    gi2, gb2 = gi.clone(), gb.clone()
    aug.gridmask(gi2, gb2, apply, par)
    ri2, rb2 = oa.gridmask(ri.copy(), rb.copy(), apply, par)
    torch.cuda.synchronize()
    assert np.array_equal(gi2.cpu().numpy(), ri2)
    assert np.array_equal(gb2.cpu().numpy(), rb2)
    # MixUp
    bx2 = np.concatenate([bx, np.zeros_like(bx)], 1)
    partner, lam = aug.draw_mixup(rng, B)
    mi, mb = aug.mixup(torch.from_numpy(img).cuda(), torch.from_numpy(bx2).cuda(), partner, lam)
    rmi, rmb = oa.mixup(img, bx2, partner, lam)
    torch.cuda.synchronize()
    np.testing.assert_allclose(mi.cpu().numpy(), rmi, rtol=1e-6, atol=1e-4)
    assert np.array_equal(mb.cpu().numpy(), rmb)


@pytest.mark.parametrize("ci,co,k,h", [(128, 256, 3, 76), (256, 512, 3, 38), (512, 1024, 3, 19), (64, 128, 3, 152), (256, 128, 1, 76)])
def test_config2_weight_gradient_forms_on_random_data(ci, co, k, h):
    """Every weight-gradient kernel form at the benchmark shapes (batch 16) on random bf16 operands against an fp32 reference
    computed tap by tap on the device (einsum over the zero-padded input - the checker, not the product): the per-tap and
    descriptor-addressed forms, the kernel-row form with atomics and with its slab workspace, and the library's own choice.
    fp32 accumulation of the same bf16 products: they agree to accumulation order (2e-5 of the largest entry)."""
    from multigriddet_amd import ops
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    g = torch.Generator(device=dev).manual_seed(ci + co + h)
    x = torch.randn(B, h, h, ci, device=dev, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, h, h, co, device=dev, generator=g).to(torch.bfloat16)
    xf, dyf = x.float(), dy.float()
    if k == 1:
        ref = torch.einsum("nhwo,nhwi->oi", dyf, xf)[:, None, :]
    else:
        xp = F.pad(xf, (0, 0, 1, 1, 1, 1))
        ref = torch.stack([torch.einsum("nhwo,nhwi->oi", dyf, xp[:, t // 3:t // 3 + h, t % 3:t % 3 + h, :]) for t in range(9)], 1)
    scale = ref.abs().max().item()
    ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    old = (ops.WGRAD_FORM, ops.WGRAD_FORM_ARG, ops.WGRAD_ROW_BLOCKS)
    try:
        for form, use_ws, blocks in ((0, False, 0), (0, True, 128), (2, False, 0), (4, False, 0), (5, False, 0), (5, True, 0), (5, True, 96)):
            ops.WGRAD_FORM, ops.WGRAD_FORM_ARG = form, (blocks if form == 5 else 0)
            ops.WGRAD_ROW_BLOCKS = blocks
            dw = torch.zeros(co, k * k, ci, device=dev)
            ops.conv_wgrad(x, dy, dw, k, 1, ws=ws if use_ws else None)
            torch.cuda.synchronize()
            err = (dw - ref).abs().max().item()
            assert err <= 2e-5 * scale, (form, use_ws, blocks, ops.L.load().mgd_last_kernel(), err / scale)
    finally:
        ops.WGRAD_FORM, ops.WGRAD_FORM_ARG, ops.WGRAD_ROW_BLOCKS = old
