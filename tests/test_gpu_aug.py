"""GPU: on-device Mosaic / MixUp / GridMask against the numpy oracle with the same host draws
(images bit-exact for the copies, MixUp within 1 ulp-level fp32 tolerance; box lists exact)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(seed, B, S, M=20, n=8):
    rng = np.random.default_rng(seed)
    img = (rng.random((B, S, S, 3), dtype=np.float32) * 255).astype(np.float32)
    bx = np.zeros((B, M, 5), np.float32)
    for b in range(B):
        for t in range(int(rng.integers(0, n + 1))):
            w, h = rng.uniform(6, S / 2, 2)
            cx, cy = rng.uniform(w / 2, S - w / 2), rng.uniform(h / 2, S - h / 2)
            bx[b, t] = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.integers(0, 80)]
    return rng, img, bx


def test_mosaic_vs_oracle():
    from multigriddet_amd.data import augment as aug
    from oracle import aug as oa
    rng, img, bx = _batch(1, 6, 160)
    bx4 = np.concatenate([bx, np.zeros((6, 60, 5), np.float32)], 1)      # 4x capacity
    src, crop = aug.draw_mosaic(rng, 6, 160)
    gi, gb = aug.mosaic(torch.from_numpy(img).cuda(), torch.from_numpy(bx4).cuda(), src, crop)
    ri, rb = oa.mosaic(img, bx4, src, crop)
    torch.cuda.synchronize()
    assert np.array_equal(gi.cpu().numpy(), ri)
    assert np.array_equal(gb.cpu().numpy(), rb)
    assert (rb[..., 2] > rb[..., 0]).sum() > 0


def test_mosaic_overflow_raises():
    from multigriddet_amd.data import augment as aug
    rng, img, bx = _batch(2, 4, 128, M=6, n=6)
    bx[:, :, :] = [10, 10, 120, 120, 1]                                    # every box survives in every quadrant
    src = np.zeros((4, 4), np.int32)
    crop = np.full((4, 2), 64, np.int32)
    with pytest.raises(RuntimeError):
        aug.mosaic(torch.from_numpy(img).cuda(), torch.from_numpy(bx).cuda(), src, crop)


def test_mixup_vs_oracle():
    from multigriddet_amd.data import augment as aug
    from oracle import aug as oa
    rng, img, bx = _batch(3, 5, 96)
    bx2 = np.concatenate([bx, np.zeros_like(bx)], 1)
    partner, lam = aug.draw_mixup(rng, 5)
    assert (partner != np.arange(5)).all() and 0.2 <= lam[0] <= 0.8
    gi, gb = aug.mixup(torch.from_numpy(img).cuda(), torch.from_numpy(bx2).cuda(), partner, lam)
    ri, rb = oa.mixup(img, bx2, partner, lam)
    torch.cuda.synchronize()
    np.testing.assert_allclose(gi.cpu().numpy(), ri, rtol=1e-6, atol=1e-4)
    assert np.array_equal(gb.cpu().numpy(), rb)


def test_gridmask_vs_oracle():
    from multigriddet_amd.data import augment as aug
    from oracle import aug as oa
    rng, img, bx = _batch(4, 6, 224)
    apply, par = aug.draw_gridmask(rng, 6, 224, prob=0.7)
    apply[0], apply[1] = 1, 0
    gi, gb = torch.from_numpy(img).cuda(), torch.from_numpy(bx).cuda()
    aug.gridmask(gi, gb, apply, par)
    ri, rb = oa.gridmask(img, bx, apply, par)
    torch.cuda.synchronize()
    assert np.array_equal(gi.cpu().numpy(), ri)
    assert np.array_equal(gb.cpu().numpy(), rb)
    assert np.array_equal(ri[1], img[1])
