"""GPU: the 69-conv graph executor (forward, backward, one optimiser step) against the torch-CPU oracle
with identical weights.  The product computes convs in bf16 MFMA with fp32 accumulation, the oracle
in fp32, so bounds are relative-L2 / cosine (1e-4 is demanded of the loss and box kernels on equal
fp32 inputs - tests/test_gpu_kernels.py - not end to end through 69 bf16 convs)."""
import numpy as np
import pytest
import torch

from conftest import coco_anchors

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available()
    from multigriddet_amd.engine import Network
    from oracle import model as om
    net = Network(80, 3, "cuda:0", seed=0)
    params = om.init_params(seed=3)
    rng = np.random.default_rng(4)
    for p in params:                      # non-trivial BN affine / bias so that their gradients matter
        if "gamma" in p:
            p["gamma"] = rng.uniform(0.7, 1.3, p["gamma"].shape).astype(np.float32)
            p["beta"] = rng.normal(0, 0.1, p["beta"].shape).astype(np.float32)
        else:
            p["bias"] = rng.normal(0, 0.1, p["bias"].shape).astype(np.float32)
    net.load_keras_style(params)
    return net, params


def test_param_count_and_layout(setup):
    net, params = setup
    from oracle import model as om
    assert len(net.layers) == 69
    assert net.count_params() == om.count_params(params) == 44996904
    assert net.n_params == 44954760                     # trainable (README "~45M")
    back = net.export_keras_style()
    for a, b in zip(back, params):
        for k in b:
            np.testing.assert_array_equal(a[k], b[k])


def _boxes(rng, B, S, n=3):
    tb = np.zeros((B, 10, 5), np.float32)
    for b in range(B):
        for t in range(n):
            w, h = rng.uniform(10, 0.5 * S, 2)
            cx, cy = rng.uniform(w / 2, S - w / 2), rng.uniform(h / 2, S - h / 2)
            tb[b, t] = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.integers(0, 80)]
    return tb


def _run_product(net, img, yt, S, B):
    from multigriddet_amd import ops
    outs = net.forward(torch.from_numpy(img).cuda())
    grids = [(S // 32,) * 2, (S // 16,) * 2, (S // 8,) * 2]
    run = ops.LossRunner(ops.make_loss_cfg(coco_anchors(), 80, (S, S), B, grids), net.device)
    douts = [torch.empty_like(o, dtype=torch.bfloat16) for o in outs]
    comp = run.run([torch.from_numpy(y).cuda() for y in yt], outs, grad_bf16=douts)
    net.zero_grad()
    net.backward(douts)
    torch.cuda.synchronize()
    return outs, float(comp[7]), net.grads.cpu().numpy()


def test_backward_wiring_with_fixed_bn_statistics(setup):
    """Every conv trainable, every BatchNorm on fixed (moving) statistics: the graph is then a fixed
    piecewise-linear map, bf16 noise is not amplified by small-batch statistics, and the product must
    match autograd of the bf16-emulating oracle tightly - this pins the whole forward/backward wiring
    (residual adds, FPN concat/upsample gradient split, stride-2 transposed convs, stem-as-GEMM)."""
    net, params = setup
    from oracle import model as om
    from oracle.loss import MultiGridLossOracle
    from oracle import targets as ot
    B, S = 2, 128
    rng = np.random.default_rng(11)
    img = rng.random((B, S, S, 3), dtype=np.float32)
    # fixed statistics = this batch's own statistics (keeps activations O(1) through 69 layers), perturbed
    pr = [dict(p) for p in params]
    st = []
    with torch.no_grad():
        om.forward(torch.from_numpy(img), om.torch_params(pr), training=True, stats_out=st)
    it = iter(st)
    for p in pr:
        if "gamma" in p:
            m, v = next(it)
            # variance x4 => every conv+BN has gain ~0.5: perturbations are damped instead of amplified
            # (a random-init 69-layer net is chaotic at unit gain: layer-by-layer error doubles, see
            # tests/debug_layers.py), which is what makes an end-to-end bound meaningful here
            p["moving_mean"] = (m.numpy() * rng.uniform(0.9, 1.1, m.shape)).astype(np.float32)
            p["moving_var"] = (4.0 * v.numpy() * rng.uniform(0.9, 1.1, v.shape)).astype(np.float32)
    net.load_keras_style(pr)
    tb = _boxes(rng, B, S)
    yt = ot.tf_preprocess_true_boxes(tb, (S, S), coco_anchors(), 80)
    tp = om.torch_params(pr, requires_grad=True)
    outs_ref = om.forward(torch.from_numpy(img), tp, training=False, emulate_bf16=True)
    loss_ref = MultiGridLossOracle(coco_anchors(), 80, (S, S))([torch.from_numpy(y) for y in yt], outs_ref)
    loss_ref.backward()
    net.training, net.freeze_bn, net.freeze_backbone = True, True, False
    try:
        outs, loss, g = _run_product(net, img, yt, S, B)
    finally:
        net.freeze_bn = False
    for l in range(3):
        r = rel_l2(outs[l].cpu().numpy(), outs_ref[l].detach().numpy())
        assert r < 0.04, f"head {l} rel-L2 {r}"      # ~60 bf16 roundings deep; smooth, not chaotic
    assert abs(loss - float(loss_ref)) < 0.01 * abs(float(loss_ref))
    worst = (1.0, -1)
    for cv, p in zip(net.layers, tp):
        gw = g[cv.off_w:cv.off_w + cv.cout * cv.T * cv.cin].reshape(cv.cout, cv.k, cv.k, cv.cin)
        ref = p["kernel"].grad.numpy().transpose(3, 0, 1, 2)
        c = cosine(gw, ref)
        worst = min(worst, (c, cv.idx))
        assert c > 0.99, f"layer {cv.idx} ({cv.role}) kernel-grad cosine {c}"
        nr = np.linalg.norm(gw) / (np.linalg.norm(ref) + 1e-30)
        assert 0.9 < nr < 1.1, f"layer {cv.idx} kernel-grad norm ratio {nr}"
        if not cv.bn:
            assert cosine(g[cv.off_a:cv.off_a + cv.cout], p["bias"].grad.numpy()) > 0.999
    print("worst kernel-grad cosine (fixed BN stats)", worst)
    net.load_keras_style(params)


def test_training_mode_forward_backward_vs_oracle(setup):
    """Training-mode BatchNorm end to end.  Batch statistics over few samples amplify bf16 storage noise
    chaotically in the deepest layers (tests/debug_layers.py: the fp32 and the bf16-emulating ORACLES differ
    from each other by >10 % at 4x4 grids), so this runs at a size where the deepest grid still has 512
    samples per channel and uses coarse bounds; the sharp checks are the kernel tests and the fixed-statistics
    wiring test above."""
    net, params = setup
    from oracle import model as om
    from oracle.loss import MultiGridLossOracle
    from oracle import targets as ot
    B, S = 8, 256
    rng = np.random.default_rng(0)
    img = rng.random((B, S, S, 3), dtype=np.float32)
    tb = _boxes(rng, B, S)
    yt = ot.tf_preprocess_true_boxes(tb, (S, S), coco_anchors(), 80)
    tp = om.torch_params(params, requires_grad=True)
    outs_ref = om.forward(torch.from_numpy(img), tp, training=True, emulate_bf16=True)
    loss_ref = MultiGridLossOracle(coco_anchors(), 80, (S, S))([torch.from_numpy(y) for y in yt], outs_ref)
    loss_ref.backward()
    net.training, net.freeze_backbone = True, False
    outs, loss, g = _run_product(net, img, yt, S, B)
    for l in range(3):
        r = rel_l2(outs[l].cpu().numpy(), outs_ref[l].detach().numpy())
        assert r < 0.3, f"head {l} rel-L2 {r}"
    assert abs(loss - float(loss_ref)) < 0.03 * abs(float(loss_ref))
    cs = []
    for cv, p in zip(net.layers, tp):
        gw = g[cv.off_w:cv.off_w + cv.cout * cv.T * cv.cin].reshape(cv.cout, cv.k, cv.k, cv.cin)
        cs.append(cosine(gw, p["kernel"].grad.numpy().transpose(3, 0, 1, 2)))
    print("kernel-grad cosines: min %.3f median %.3f; head-only min %.3f" % (min(cs), float(np.median(cs)), min(cs[52:])))
    # gradients of the early layers pass through ~60 chaotic layers twice; only the head is bounded here
    assert min(cs[52:]) > 0.8 and float(np.median(cs)) > 0.7


def test_fp32_precision_end_to_end_vs_oracle_strict():
    """Network(precision="fp32"): the whole 69-conv graph in the reference's default numeric type (Keras float32,
    models/layers.py:43-95) - forward with batch-statistic BatchNorm, MultiGridLoss, backward through every layer -
    against the fp32 torch-CPU oracle with identical weights.  With bf16 storage this comparison needs cosine bounds
    (test_training_mode_forward_backward_vs_oracle); in fp32 the bounds are tight: heads to 1e-3 relative L2, loss to
    1e-4, every one of the 69 kernel gradients, every BatchNorm gamma/beta gradient and the bias gradients to cosine
    0.999 and norm ratio 1 +- 1e-2 (float32 summation order is the only difference left; the first layers sit behind
    ~130 layer applications of a random-init network, which amplifies even that: the stem reaches 0.99988)."""
    from multigriddet_amd.engine import Network
    from multigriddet_amd import ops
    from oracle import model as om
    from oracle.loss import MultiGridLossOracle
    from oracle import targets as ot
    net = Network(80, 3, "cuda:0", seed=0, precision="fp32")
    params = om.init_params(seed=3)
    rng = np.random.default_rng(4)
    for p in params:
        if "gamma" in p:
            p["gamma"] = rng.uniform(0.7, 1.3, p["gamma"].shape).astype(np.float32)
            p["beta"] = rng.normal(0, 0.1, p["beta"].shape).astype(np.float32)
        else:
            p["bias"] = rng.normal(0, 0.1, p["bias"].shape).astype(np.float32)
    net.load_keras_style(params)
    B, S = 4, 256
    rng = np.random.default_rng(0)
    img = rng.random((B, S, S, 3), dtype=np.float32)
    tb = _boxes(rng, B, S)
    yt = ot.tf_preprocess_true_boxes(tb, (S, S), coco_anchors(), 80)
    tp = om.torch_params(params, requires_grad=True)
    outs_ref = om.forward(torch.from_numpy(img), tp, training=True, emulate_bf16=False)
    loss_ref = MultiGridLossOracle(coco_anchors(), 80, (S, S))([torch.from_numpy(y) for y in yt], outs_ref)
    loss_ref.backward()
    net.training = True
    outs = net.forward(torch.from_numpy(img).cuda())
    grids = [(S // 32,) * 2, (S // 16,) * 2, (S // 8,) * 2]
    run = ops.LossRunner(ops.make_loss_cfg(coco_anchors(), 80, (S, S), B, grids), net.device)
    douts = [torch.empty_like(o) for o in outs]
    comp = run.run([torch.from_numpy(y).cuda() for y in yt], outs, grad_f32=douts)
    net.zero_grad()
    net.backward(douts)
    torch.cuda.synchronize()
    for l in range(3):
        r = rel_l2(outs[l].cpu().numpy(), outs_ref[l].detach().numpy())
        assert r < 1e-3, f"head {l} rel-L2 {r}"
    assert abs(float(comp[7]) - float(loss_ref.detach())) < 1e-4 * abs(float(loss_ref.detach()))
    g = net.grads.cpu().numpy()
    worst = (1.0, -1)
    for cv, p in zip(net.layers, tp):
        gw = g[cv.off_w:cv.off_w + cv.cout * cv.T * cv.cin].reshape(cv.cout, cv.k, cv.k, cv.cin)
        ref = p["kernel"].grad.numpy().transpose(3, 0, 1, 2)
        c = cosine(gw, ref)
        worst = min(worst, (c, cv.idx))
        assert c > 0.999, f"layer {cv.idx} ({cv.role}) kernel-grad cosine {c}"
        nr = np.linalg.norm(gw) / (np.linalg.norm(ref) + 1e-30)
        assert abs(nr - 1.0) < 1e-2, f"layer {cv.idx} kernel-grad norm ratio {nr}"
        if cv.bn:
            assert cosine(g[cv.off_a:cv.off_a + cv.cout], p["gamma"].grad.numpy()) > 0.999, f"layer {cv.idx} dgamma"
            assert cosine(g[cv.off_b:cv.off_b + cv.cout], p["beta"].grad.numpy()) > 0.999, f"layer {cv.idx} dbeta"
        else:
            assert cosine(g[cv.off_a:cv.off_a + cv.cout], p["bias"].grad.numpy()) > 0.9999
    assert min(cosine(g[cv.off_w:cv.off_w + cv.cout * cv.T * cv.cin].reshape(cv.cout, cv.k, cv.k, cv.cin),
                      p["kernel"].grad.numpy().transpose(3, 0, 1, 2)) for cv, p in list(zip(net.layers, tp))[52:]) > 0.9995
    # (the head layers, 52 ... 68: five times tighter than the bound on every layer above.  Run to run the minimum moves
    # between 0.99987 and 0.99996 with the order of the fp32 atomics in the BatchNorm sums - seen over the round-4 full-suite
    # runs - so a bound of 0.9999 failed one run in five without anything being wrong.)
    print("fp32 end to end: worst kernel-grad cosine", worst)
    # moving statistics follow Keras' momentum update from the batch statistics (momentum 0.99)
    n_bn = net.moving.numel() // 2
    assert float(net.moving[:n_bn].abs().max()) > 0 and torch.isfinite(net.moving).all()
    # one optimiser step in fp32 mode through TrainStep
    from multigriddet_amd.train_step import TrainStep
    ts = TrainStep(net, coco_anchors(), 80, (S, S), B, lr=1e-3)
    l0 = float(ts.step(torch.from_numpy(img).cuda(), torch.from_numpy(tb).cuda())[7])
    l1 = float(ts.step(torch.from_numpy(img).cuda(), torch.from_numpy(tb).cuda())[7])
    assert np.isfinite([l0, l1]).all() and l1 < l0


def test_inference_mode_and_frozen_backbone(setup):
    net, params = setup
    from oracle import model as om
    B, S = 2, 96
    img = np.random.default_rng(1).random((B, S, S, 3), dtype=np.float32)
    net.load_keras_style(params)          # earlier training-mode forwards moved the moving statistics
    tp = om.torch_params(params)
    ref = om.forward(torch.from_numpy(img), tp, training=False, emulate_bf16=True)
    net.training = False
    outs = net.forward(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    for l in range(3):
        assert rel_l2(outs[l].cpu().numpy(), ref[l].numpy()) < 0.05
    net.training = True


def test_train_step_reduces_loss(setup):
    net, _ = setup
    from multigriddet_amd.train_step import TrainStep
    net.reset_parameters(0)
    B, S = 4, 128
    rng = np.random.default_rng(5)
    img = torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).cuda()
    tb = np.zeros((B, 10, 5), np.float32)
    for b in range(B):
        tb[b, 0] = [20, 30, 90, 100, 3]
        tb[b, 1] = [60, 10, 120, 50, 7]
    ts = TrainStep(net, coco_anchors(), 80, (S, S), B, lr=1e-3)
    losses = []
    boxes = torch.from_numpy(tb).cuda()
    for _ in range(12):
        losses.append(float(ts.step(img, boxes)[7]))
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.7 * losses[0], losses


@pytest.mark.gpu
def test_graph_replay_matches_eager_step():
    """TrainStep.enable_graph(): the captured hipGraph (both streams, device-resident Adam step size) gives the same
    loss trajectory as launching the kernels from Python (weight-gradient atomics make the two runs differ in the
    last bits, hence a tolerance)."""
    import torch
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    import bench
    dev = torch.device("cuda:0")
    img, bx = bench.synth_batch(0, 4, 256)
    img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
    losses = {}
    for mode in (False, True):
        net = Network(80, 3, dev, seed=0)
        ts = TrainStep(net, bench.coco_anchors(), 80, (256, 256), 4, lr=1e-4)
        ts.enable_graph(mode)
        out = []
        for i in range(6):
            # a fresh tensor every second step exercises the copy into the graph's static inputs
            a, b = (img.clone(), bx.clone()) if i % 2 else (img, bx)
            out.append(float(ts.step(a, b)[7]))
        torch.cuda.synchronize()
        losses[mode] = out
        assert ts.step_count == 6
    a, b = np.array(losses[False]), np.array(losses[True])
    assert np.all(np.isfinite(b))
    np.testing.assert_allclose(b, a, rtol=2e-2)
    assert a[-1] < a[0]                      # it trains


@pytest.mark.gpu
@pytest.mark.parametrize("freeze_bn", [True, False])
def test_launch_plan_replay_matches_eager_step(freeze_bn):
    """TrainStep.enable_plan(): the step recorded as a launch plan (csrc/plan.cpp: every C-ABI call with its arguments and
    stream, the cross-stream waits, the buffer clears) and replayed by mgd_plan_run issues the SAME launches as the eager
    Python path.  Three networks: A steps eagerly, B from its plan, C eagerly again; before every compared step B and C are
    given A's exact state (weights, BatchNorm moving statistics, Adam moments, step count), so all three compute the same step
    on fresh input tensors at new addresses (plan parameters).
    freeze_bn=True (BatchNorm on its moving statistics): the step is deterministic up to the order of the weight gradient's
    fp32 atomics - loss and full gradient of B must equal A's to 1e-5 / cosine 0.99999.
    freeze_bn=False (batch statistics): 69 bf16 layers amplify the last-bit noise of the statistics' atomics at random
    initialisation: two EAGER runs from one state already differ by ~1e-3 in the loss and have a
    gradient cosine of 0.86 - 0.89 (measured, four networks side by side); B must agree with A within that band, like C."""
    import torch
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    import bench
    dev = torch.device("cuda:0")
    batches = []
    for seed in (0, 1):
        img, bx = bench.synth_batch(seed, 4, 256)
        batches.append((torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)))
    nets = [Network(80, 3, dev, seed=0) for _ in range(3)]
    for n in nets:
        n.freeze_bn = freeze_bn
    ta, tb, tc = (TrainStep(n, bench.coco_anchors(), 80, (256, 256), 4, lr=1e-4) for n in nets)
    tb.enable_plan(True)
    na, nb, nc = nets
    for i in range(3):                               # the planned step: two eager steps, then the recording step
        for t in (ta, tb, tc):
            t.step(*batches[0])
    plans = [st["plan"] for st in tb._plans.values()]
    assert len(plans) == 1 and plans[0] not in (None, False) and plans[0].size > 300, [getattr(p, "size", p) for p in plans]
    keep = []

    def cosine(x, y):
        x, y = x.double(), y.double()
        return float((x * y).sum() / (x.norm() * y.norm()))
    for i in range(4):
        torch.cuda.synchronize()
        for n, t in ((nb, tb), (nc, tc)):
            n.params.copy_(na.params); n.moving.copy_(na.moving); t.m.copy_(ta.m); t.v.copy_(ta.v)
            n.refresh_packed()
            t.step_count = ta.step_count
        a, b = batches[i % 2]
        ins = [(a, b), (a.clone(), b.clone()), (a.clone(), b.clone())]      # new addresses every step
        keep.append(ins)
        la, lb, lc = (float(t.step(*x)[7]) for t, x in zip((ta, tb, tc), ins))
        torch.cuda.synchronize()
        cab, cac = cosine(na.grads, nb.grads), cosine(na.grads, nc.grads)
        if freeze_bn:
            assert abs(la - lb) <= 1e-5 * abs(la), (i, la, lb)
            assert cab > 0.99999, (i, cab, cac)
        else:
            assert np.isfinite(lb) and abs(la - lb) <= 5e-3 * abs(la) and abs(la - lc) <= 5e-3 * abs(la), (i, la, lb, lc)
            assert cab > 0.75 and abs(cab - cac) < 0.1, (i, cab, cac)
    assert tb.step_count == ta.step_count == 7


@pytest.mark.gpu
def test_multiscale_steps_share_one_network():
    """BASELINE config 3 (multi-scale {320..608}): one Network / TrainStep takes batches of different resolutions step by
    step - every arena, loss configuration and target grid is keyed by the input size, weights and Adam state are shared."""
    import torch
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    import bench
    dev = torch.device("cuda:0")
    net = Network(80, 3, dev, seed=0)
    ts = TrainStep(net, bench.coco_anchors(), 80, (320, 320), 4, lr=1e-4)
    sizes = [320, 352, 416, 320, 352, 416, 320]
    batches = {}
    for s in set(sizes):
        img, bx = bench.synth_batch(s, 4, s)
        batches[s] = (torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev))
    first, last = {}, {}
    for s in sizes:
        comp = ts.step(*batches[s])
        v = float(comp[7])
        assert np.isfinite(v)
        first.setdefault(s, v)
        last[s] = v
    torch.cuda.synchronize()
    assert ts.step_count == len(sizes)
    assert last[320] < first[320]            # the shared weights keep training across resolutions


@pytest.mark.gpu
def test_fold_bn_inference_matches_unfolded():
    """Network.fold_bn(): BatchNorm folded into the convs (scaled weights, shift as bias, LeakyReLU + residual in the conv
    epilogue) against the conv + BN/activation path, on non-trivial moving statistics.  bf16 storage on both sides, the
    folded path rounds the scaled weights instead of y, hence a relative-L2 tolerance (the statistics are damped as in the
    wiring test so that 69 layers do not amplify the rounding difference)."""
    import torch
    from multigriddet_amd.engine import Network
    dev = torch.device("cuda:0")
    net = Network(80, 3, dev, seed=3)
    g = torch.Generator(device="cpu").manual_seed(11)
    for cv in net.layers:
        if cv.bn:
            C = cv.cout
            cv.gamma.copy_((torch.rand(C, generator=g) * 0.5 + 0.75).to(dev))
            cv.beta.copy_((torch.randn(C, generator=g) * 0.1).to(dev))
            cv.mm.copy_((torch.randn(C, generator=g) * 0.05).to(dev))
            cv.mv.copy_((torch.rand(C, generator=g) * 1.0 + 3.5).to(dev))      # damped: per-layer gain ~0.5
    net.refresh_packed()
    net.training = False
    x = torch.rand(2, 128, 160, 3, generator=g).to(dev)
    ref = [o.clone() for o in net.forward(x)]
    acts_ref = [net._last["a"][i].clone() for i in range(8)]
    net.fold_bn(True)
    out = [o.clone() for o in net.forward(x)]
    acts = [net._last["a"][i].clone() for i in range(8)]
    net.fold_bn(False)
    again = net.forward(x)
    torch.cuda.synchronize()
    for i, (ar, af) in enumerate(zip(acts_ref, acts)):          # layer by layer at the start of the network
        rel = ((af.float() - ar.float()).norm() / ar.float().norm()).item()
        assert rel < 1.5e-2, (i, rel)
    for r, o, a2 in zip(ref, out, again):
        assert o.shape == r.shape and torch.isfinite(o).all()
        rel = ((o - r).norm() / r.norm()).item()
        assert rel < 5e-2, rel
        assert torch.equal(a2, r)              # switching it off restores the pinned path bit for bit


@pytest.mark.parametrize("fold", [False, True])
def test_inference_graph_replay_matches_eager(fold):
    """MultiGridDetModel.enable_graph(): the forward pass replayed from a captured hipGraph returns bit-identical head
    tensors to the eager launches (same kernels, no atomics in the inference pass), also on new input data and after the
    weights changed in place."""
    import torch
    from multigriddet_amd.models import build_multigriddet_darknet
    model, _ = build_multigriddet_darknet(input_shape=(96, 128, 3), num_classes=80)
    if fold:
        model.fold_bn(True)
    g = torch.Generator(device="cpu").manual_seed(21)
    xs = [torch.rand(1, 96, 128, 3, generator=g).cuda() for _ in range(5)]
    ref = [[o.clone() for o in model(x)] for x in xs]
    model.enable_graph(True)
    for x, r in zip(xs, ref):                       # calls 1-2 eager warm-up, 3 captures, 4-5 replay
        outs = model(x)
        torch.cuda.synchronize()
        for o, e in zip(outs, r):
            assert torch.equal(o, e)
    assert any(st["graph"] is not None for st in model._graphs.values())
    # another shape gets its own graph; the first one keeps working
    x2 = torch.rand(2, 64, 64, 3, generator=g).cuda()
    model.enable_graph(False)
    e2 = [o.clone() for o in model(x2)]
    model.enable_graph(True)
    for _ in range(4):
        o2 = model(x2)
    torch.cuda.synchronize()
    for o, e in zip(o2, e2):
        assert torch.equal(o, e)


def test_cabi_rccl_communicator_single_rank():
    """mgd_comm_* (the library's RCCL binding for the gradient sum): a one-rank communicator on this GPU - the sum over
    one rank is the identity, in place, stream-ordered; error paths return MGD_EINVAL with a message."""
    import ctypes as C
    import torch
    from multigriddet_amd import _lib as L
    from multigriddet_amd.dp import CabiComm
    comm = CabiComm(0, 1, torch.device("cuda:0"))
    g = torch.randn(3_000_001, device="cuda")
    ref = g.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.all_reduce_sum_(g[:2_000_000])
        comm.all_reduce_sum_(g[2_000_000:])
    side.synchronize()
    assert torch.equal(g, ref)
    lib = L.load()
    assert lib.mgd_comm_allreduce_bucket(None, L.ptr(g), C.c_int64(4), None) == -1
    assert b"comm_allreduce_bucket" in lib.mgd_last_error()
    assert lib.mgd_comm_init(None, 0, 1, None) == -1
    comm.destroy()
    comm.destroy()          # idempotent


def test_early_optimizer_matches_tail_optimizer():
    """TrainStep.early_optimizer: Adam + re-pack of stage 5 and the head on the side stream in the middle of backward
    against the single optimiser launch at the end of the step.  After ONE step from identical weights the two differ
    only by the last-bit noise of the weight-gradient atomics (first moments equal to 1e-3 relative; Adam's first step
    is lr * sign(g), so isolated near-zero gradients may flip by 2 lr); further steps are compared on the loss only.
    The packed images always equal a fresh pack of the final masters."""
    import torch
    from multigriddet_amd.engine import Network
    from multigriddet_amd.train_step import TrainStep
    import bench
    dev = torch.device("cuda:0")
    img, bx = bench.synth_batch(0, 4, 256)
    img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
    res = {}
    lr = 1e-4
    for mode in (False, True):
        net = Network(80, 3, dev, seed=0)
        ts = TrainStep(net, bench.coco_anchors(), 80, (256, 256), 4, lr=lr)
        ts.early_optimizer = mode
        assert ts._early_ok() == mode
        # fixed (moving) BatchNorm statistics: the net is then a fixed piecewise-linear map and two runs differ by the
        # atomics' last bits only - with batch statistics the same comparison drowns in chaotic amplification (after one
        # step two runs of the SAME mode already differ by 0.4 lr on average); the stream choreography is identical
        net.freeze_bn = True
        losses = [float(ts.step(img, bx)[7])]
        torch.cuda.synchronize()
        p1, m1 = net.params.clone(), ts.m.clone()
        losses += [float(ts.step(img, bx)[7]) for _ in range(3)]
        torch.cuda.synchronize()
        with_pk = [cv for cv in net.layers if cv.pk is not None]
        packed = [cv.pk.fwd.clone() for cv in with_pk]
        net.refresh_packed()
        torch.cuda.synchronize()
        for a, cv in zip(packed, with_pk):
            assert torch.equal(a.view(torch.int16), cv.pk.fwd.view(torch.int16))
        res[mode] = (np.array(losses), p1, m1)
    la, pa, ma = res[False]
    lb, pb, mb = res[True]
    np.testing.assert_allclose(lb, la, rtol=2e-2)
    assert abs(la[0] - lb[0]) <= 2e-3 * abs(la[0])  # same weights; BatchNorm statistics are summed with float atomics
    d = (pa - pb).abs()
    assert d.max().item() <= 2.5 * lr
    assert d.mean().item() <= 0.02 * lr
    assert (ma - mb).abs().max().item() <= 1e-3 * ma.abs().max().item() + 1e-7


def test_two_models_on_two_streams_share_no_latency_state():
    """"no global mutable state; safe to call concurrently on different streams".  The batch-1 forward runs the
    latency-form convolutions (K ranges met inside the kernel through tickets + partial tiles in uncached memory); that
    workspace belongs to the CALLER - one ops.LatencyWorkspace per (Network, stream), the library keeps none.  Two folded
    models at 608 x 608, batch 1, driven from two streams with their launches interleaved and no synchronisation in between,
    must each return exactly what they return alone, run after run, and leave every ticket at zero."""
    import torch
    from multigriddet_amd import ops
    from multigriddet_amd.models import build_multigriddet_darknet
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    models, xs = [], []
    for seed in (1, 2):
        torch.manual_seed(seed)
        m, _ = build_multigriddet_darknet(input_shape=(608, 608, 3), num_classes=80)
        for cv in m.net.layers:                      # distinct, non-trivial moving statistics per model
            if cv.bn:
                cv.mm.copy_((torch.randn(cv.cout, generator=g) * 0.05).to(dev))
                cv.mv.copy_((torch.rand(cv.cout, generator=g) + 3.0).to(dev))
                cv.w.copy_((torch.randn(cv.w.shape, generator=g) * (1.5 / (cv.T * cv.cin) ** 0.5)).to(dev))
        m.net.refresh_packed()
        m.fold_bn(True)
        models.append(m)
        xs.append(torch.rand(1, 608, 608, 3, generator=g).to(dev))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    serial = []
    for m, x, s in zip(models, xs, streams):         # each alone on its stream
        with torch.cuda.stream(s):
            serial.append([o.clone() for o in m(x)])
        torch.cuda.synchronize()
    assert not all(torch.equal(a, b) for a, b in zip(*serial))            # two different networks
    fam = set()
    for rnd in range(4):                             # together: launches interleaved, nothing waits for anything
        outs = [None, None]
        for i in (0, 1) if rnd % 2 == 0 else (1, 0):
            with torch.cuda.stream(streams[i]):
                outs[i] = [o.clone() for o in models[i](xs[i])]
                fam.add(ops.L.load().mgd_last_kernel())
        torch.cuda.synchronize()
        for i in (0, 1):
            for o, e in zip(outs[i], serial[i]):
                assert torch.equal(o, e), (rnd, i)
    wss = [ws for m in models for ws in m.net._lat_ws.values()]
    assert len(wss) == 2 and wss[0].ptr != wss[1].ptr                     # one workspace per (network, stream)
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            for ws in models[i].net._lat_ws.values():
                assert not ws.tickets().any()


def test_streams_are_one_per_role_and_device():
    """multigriddet_amd/streams.py: every Network of the process runs its weight gradients on the same side stream and every
    TrainStep on the same high-priority main stream (hardware queues are few: a stream per object makes streams share queues,
    and the two-stream backward serialises)."""
    from multigriddet_amd.engine import Network
    from multigriddet_amd.streams import shared_stream
    from multigriddet_amd.train_step import TrainStep
    from conftest import coco_anchors
    dev = torch.device("cuda:0")
    n1, n2 = Network(4, 3, dev, seed=0), Network(4, 3, dev, seed=1)
    assert n1.wg_stream is n2.wg_stream is shared_stream("wgrad", dev)
    t1 = TrainStep(n1, coco_anchors(), 4, (128, 128), 2, lr=1e-4)
    t2 = TrainStep(n2, coco_anchors(), 4, (128, 128), 2, lr=1e-4)
    assert t1.main_stream is t2.main_stream and t1.main_stream is not n1.wg_stream
    assert t1.main_stream.priority < n1.wg_stream.priority or t1.main_stream.priority == -1


def test_prediction_branches_on_the_side_stream_change_nothing():
    """Network.forward runs the prediction branches of the first two scales on the side stream beside the next scale's trunk
    (batches of at least parallel_heads_min_pixels).  Wherever the forward pass is deterministic - inference, and training
    with frozen BatchNorm statistics (no statistics atomics) - the outputs are bit-identical to the one-stream order.  (With
    batch statistics two runs of the SAME order already differ: the atomics' order moves a statistic by an ulp and 75 bf16
    layers of a random-init network amplify it.)"""
    from multigriddet_amd.engine import Network
    dev = torch.device("cuda:0")
    net = Network(8, 3, dev, seed=3)
    net.parallel_heads_min_pixels = 0
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 160, 192, 3, generator=g).to(dev)
    res = {}
    for ph in (False, True):
        net.parallel_heads = ph
        net.training, net.freeze_bn = False, False
        inf = [o.clone() for o in net.forward(x)]
        net.training, net.freeze_bn = True, True
        outs = [o.clone() for o in net.forward(x)]
        torch.cuda.synchronize()
        res[ph] = inf + outs
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
