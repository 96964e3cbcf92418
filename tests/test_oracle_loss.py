"""CPU: properties of the loss oracle's GIoU / DIoU / CIoU / focal branches.

The reference's IoU losses multiply a [B,H,W] loss by the [B,H,W,1] object mask (losses/iou_losses.py:70-93,
140-158); the oracle restates that literally and lets torch broadcast as TensorFlow does.  The device kernel uses the
closed form of that 4-D product's sum; these tests pin the closed form against the literal broadcast on the CPU, the
shapes for which the broadcast is defined, and textbook values of the per-box formulas.
"""
import numpy as np
import pytest
import torch

from oracle import loss as ol


def _rand(B, H, seed):
    g = torch.Generator().manual_seed(seed)
    txy, twh = torch.rand(B, H, H, 2, generator=g, dtype=torch.float64), torch.randn(B, H, H, 2, generator=g, dtype=torch.float64)
    pxy, pwh = torch.randn(B, H, H, 2, generator=g, dtype=torch.float64), torch.randn(B, H, H, 2, generator=g, dtype=torch.float64).abs() + 0.2
    mask = (torch.rand(B, H, H, 1, generator=g, dtype=torch.float64) < 0.2).double()
    return txy, twh, pxy, pwh, mask


@pytest.mark.parametrize("B,H", [(1, 7), (7, 7), (1, 19), (19, 19)])
@pytest.mark.parametrize("kind", ["giou", "diou", "ciou"])
def test_broadcast_sum_closed_form(B, H, kind):
    """sum([B,H,W] * [B,H,W,1]) == sum_{b,h,w} cell[b,h,w] * Wt[b,h] + W * sum_{positives} dist, with
    Wt[b,h] = sum_i mask[i,b,h] (B == H) or sum_j mask[0,j,h] (B == 1) - the form csrc/loss.hip computes."""
    txy, twh, pxy, pwh, mask = _rand(B, H, 3 + B + H)
    fn = dict(giou=ol.giou_loss, diou=ol.diou_loss, ciou=ol.ciou_loss)[kind]
    literal = float(fn(txy, twh, pxy, pwh, mask))
    iou, union, ewh = ol._box_terms(txy, twh, pxy, pwh)
    m = mask[..., 0]
    Wt = m.sum(0) if B == H else m[0].sum(0)[None, :]          # [B,H] indexed [b,h]
    if kind == "giou":
        earea = ewh[..., 0] * ewh[..., 1]
        cell = 1.0 - (iou - (earea - union) / (earea + ol.EPS))
        dist = torch.zeros_like(iou)
    else:
        cell = 1.0 - iou
        dist = ((txy - pxy) ** 2).sum(-1) / ((ewh ** 2).sum(-1) + ol.EPS)
        if kind == "ciou":
            v = 4.0 * (torch.atan2(twh[..., 0], twh[..., 1]) - torch.atan2(pwh[..., 0], pwh[..., 1])) ** 2 / np.pi ** 2
            cell = cell + v * v / (1.0 - iou + v + ol.EPS)
    closed = float((cell * Wt[:, :, None]).sum() + H * (dist * m).sum())
    assert abs(literal - closed) <= 1e-9 * max(1.0, abs(literal))


@pytest.mark.parametrize("B,H", [(2, 7), (16, 19), (4, 4 + 1)])
def test_broadcast_undefined_shapes_raise(B, H):
    txy, twh, pxy, pwh, mask = _rand(B, H, 1)
    if B == H:
        pytest.skip("defined")
    with pytest.raises(RuntimeError):
        ol.giou_loss(txy, twh, pxy, pwh, mask)


def test_textbook_values_per_cell():
    """Identical boxes: IoU 1, all three losses 0.  Disjoint unit boxes two apart: IoU 0, GIoU = -1/3, DIoU = -4/10."""
    one = torch.ones(1, 1, 1, dtype=torch.float64)
    t_xy = torch.tensor([[[[0.5, 0.5]]]], dtype=torch.float64)
    wh = torch.tensor([[[[1.0, 1.0]]]], dtype=torch.float64)
    for fn in (ol.giou_loss, ol.diou_loss, ol.ciou_loss):
        assert abs(float(fn(t_xy, wh, t_xy, wh, one))) < 1e-6
    p_xy = torch.tensor([[[[2.5, 0.5]]]], dtype=torch.float64)
    assert abs(float(ol.giou_loss(t_xy, wh, p_xy, wh, one)) - (1.0 + 1.0 / 3.0)) < 1e-6
    assert abs(float(ol.diou_loss(t_xy, wh, p_xy, wh, one)) - (1.0 + 4.0 / 10.0)) < 1e-6
    assert abs(float(ol.ciou_loss(t_xy, wh, p_xy, wh, one)) - (1.0 + 4.0 / 10.0)) < 1e-6      # equal aspect: v = 0


def test_focal_reduces_to_scaled_bce_at_gamma_zero():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 5, generator=g, dtype=torch.float64)
    y = (torch.rand(4, 5, generator=g) < 0.3).double()
    fl = ol.sigmoid_focal(y, x, 0.25, 0.0)
    ref = (y * 0.25 + (1 - y) * 0.75) * ol.bce_logits(y, x)
    assert torch.allclose(fl, ref)
    onehot = torch.nn.functional.one_hot(torch.tensor([0, 2, 1, 4]), 5).double()
    sf = ol.softmax_focal(onehot, x, 0.0)
    assert torch.allclose(sf, -(onehot * torch.log_softmax(x, -1)).sum(-1))
    # one class: softmax == 1, cross entropy == 0 - the only C for which the reference's three-scale softmax branch runs
    assert float(ol.softmax_focal(torch.ones(3, 1, dtype=torch.float64), torch.randn(3, 1, dtype=torch.float64), 2.0).abs().max()) == 0.0


def test_oracle_three_scales_tf_ref_needs_batch_one():
    from conftest import coco_anchors
    rng = np.random.default_rng(0)
    grids = [(4, 4), (8, 8), (16, 16)]

    def mk(B):
        yt = [np.zeros((B, g[0], g[1], 88), np.float32) for g in grids]
        for y in yt:
            y[:, 1, 2, 4] = 1.0
            y[:, 1, 2, 5] = 1.0
            y[:, 1, 2, 8 + 3] = 1.0
        yp = [rng.standard_normal((B, g[0], g[1], 88)).astype(np.float32) for g in grids]
        return yt, yp
    o = ol.MultiGridLossOracle(coco_anchors(), 80, (128, 128), loss_option=3, use_giou_loss=True, dtype=torch.float64)
    tot, comp, grads = o.value_and_grad(*mk(1))
    assert np.isfinite(tot) and comp["loc"] != 0.0
    with pytest.raises(RuntimeError):
        o.value_and_grad(*mk(4))
    o2 = ol.MultiGridLossOracle(coco_anchors(), 80, (128, 128), loss_option=3, use_giou_loss=True, compat="fixed",
                                dtype=torch.float64)
    tot2, _, _ = o2.value_and_grad(*mk(4))
    assert np.isfinite(tot2)
