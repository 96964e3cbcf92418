"""CPU oracle for MultiGridLoss (value and gradient).  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, as the checker.  The product path never imports `oracle`.

PARITY UNPINNED: the reference's loss is TensorFlow code (multigriddet/losses/multigrid_loss.py)
that cannot run in this container and the reference's tests hold no loss values (SURVEY.md §4,
§8c).  This file restates the algorithm line by line in torch-CPU (so autograd supplies the
reference gradient); every quirk of the TF code is kept:
  * ignore-mask grid built with meshgrid(indexing='ij') => coords[row,col] = (row, col)
    (multigrid_loss.py:545-551), i.e. transposed relative to the decoder;
  * GT boxes for the ignore mask = every positive cell, wh = exp(t)*anchor*scale with pixel
    anchors multiplied by the stride again (:562-572);
  * anchor_scale applied twice (:349/:390 and :433); object_scale applied twice on positives
    (:908 and :432);
  * normalisation = product over `loss_normalization` entries, floored at 1 (:194-231).
  * loss_option 3 with use_giou_loss / use_diou_loss / use_ciou_loss (losses/iou_losses.py:36-237 via
    multigrid_loss.py:353-364) and use_softmax_loss (losses/focal_loss.py:80-114 via multigrid_loss.py:815-828):
    compat="tf_ref" restates the code LITERALLY, including the product of a [B,H,W] loss with the [B,H,W,1] object mask -
    torch broadcasts exactly as TensorFlow/numpy do (trailing dimensions aligned), so the 4-D result, and the
    RuntimeError for shapes that do not broadcast (TF: InvalidArgumentError), come out of the same arithmetic; the boxes
    are the raw tensors as given.  compat="fixed" is this build's repaired form (per-cell mask, boxes decoded to
    grid-cell units with the assigned anchor) and has no reference counterpart.
  * use_focal_loss (SigmoidFocalLoss, focal_loss.py:40-77 via multigrid_loss.py:800-813); use_softmax_loss takes
    precedence when both are set (multigrid_loss.py:400-407).
Keras semantics used: K.epsilon() = 1e-7; K.binary_crossentropy(from_logits=True) =
max(x,0) - x*z + log1p(exp(-|x|)); K.categorical_crossentropy(from_logits=True) = -sum(y * log_softmax(x)).
"""
import math

import torch

EPS = 1e-7


def bce_logits(z, x):
    return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x)))


def xy_act(p):
    return torch.tanh(0.15 * p) + torch.sigmoid(0.15 * p)


def _box_terms(txy, twh, pxy, pwh):
    """Shared head of GIoULoss/DIoULoss/CIoULoss.compute_loss (iou_losses.py:58-79, 121-141, 186-206)."""
    tmin, tmax = txy - twh / 2.0, txy + twh / 2.0
    pmin, pmax = pxy - pwh / 2.0, pxy + pwh / 2.0
    iwh = torch.clamp(torch.minimum(tmax, pmax) - torch.maximum(tmin, pmin), min=0.0)
    inter = iwh[..., 0] * iwh[..., 1]
    union = twh[..., 0] * twh[..., 1] + pwh[..., 0] * pwh[..., 1] - inter
    iou = inter / (union + EPS)
    ewh = torch.clamp(torch.maximum(tmax, pmax) - torch.minimum(tmin, pmin), min=0.0)
    return iou, union, ewh


def giou_loss(txy, twh, pxy, pwh, mask):
    """iou_losses.py:58-95.  mask is [B,H,W,1]; the [B,H,W] * [B,H,W,1] product broadcasts as in the reference."""
    iou, union, ewh = _box_terms(txy, twh, pxy, pwh)
    earea = ewh[..., 0] * ewh[..., 1]
    giou = iou - (earea - union) / (earea + EPS)
    return ((1.0 - giou) * mask).sum()


def diou_loss(txy, twh, pxy, pwh, mask, keepdim=True):
    """iou_losses.py:121-160 (centre distance and enclosing diagonal are keepdims tensors in the reference;
    keepdim=False is the repaired per-cell form, mask [B,H,W])."""
    iou, _, ewh = _box_terms(txy, twh, pxy, pwh)
    cd = ((txy - pxy) ** 2).sum(-1, keepdim=keepdim)
    ed = (ewh ** 2).sum(-1, keepdim=keepdim)
    diou = iou - cd / (ed + EPS)
    return ((1.0 - diou) * mask).sum()


def ciou_loss(txy, twh, pxy, pwh, mask, keepdim=True):
    """iou_losses.py:186-237 (alpha is NOT a stop-gradient in the reference)."""
    iou, _, ewh = _box_terms(txy, twh, pxy, pwh)
    cd = ((txy - pxy) ** 2).sum(-1, keepdim=keepdim)
    ed = (ewh ** 2).sum(-1, keepdim=keepdim)
    diou = iou - cd / (ed + EPS)
    v = 4.0 * (torch.atan2(twh[..., 0], twh[..., 1]) - torch.atan2(pwh[..., 0], pwh[..., 1])) ** 2 / (math.pi * math.pi)
    alpha = v / (1.0 - iou + v + EPS)
    ciou = diou - alpha * v
    return ((1.0 - ciou) * mask).sum()


def sigmoid_focal(y, x, alpha, gamma):
    """SigmoidFocalLoss.compute_loss, focal_loss.py:49-77."""
    p = torch.sigmoid(x)
    pt = y * p + (1 - y) * (1 - p)
    return torch.pow(1.0 - pt, gamma) * (y * alpha + (1 - y) * (1 - alpha)) * bce_logits(y, x)


def softmax_focal(y, x, gamma):
    """SoftmaxFocalLoss.compute_loss, focal_loss.py:89-114 -> [B,H,W]."""
    ce = -(y * torch.log_softmax(x, -1)).sum(-1)
    pt = (y * torch.softmax(x, -1)).sum(-1)
    return torch.pow(1.0 - pt, gamma) * ce


def _patches(t, k=3):
    """tf.image.extract_patches(sizes=k, strides=1, padding='SAME') -> [B,H,W,k*k,C], zero padded,
    patch index row-major (multigrid_loss.py:949-964)."""
    B, H, W, C = t.shape
    r = k // 2
    p = torch.zeros(B, H + 2 * r, W + 2 * r, C, dtype=t.dtype)
    p[:, r:r + H, r:r + W] = t
    out = [p[:, di:di + H, dj:dj + W] for di in range(k) for dj in range(k)]
    return torch.stack(out, dim=3)


class MultiGridLossOracle:
    """Mirror of MultiGridLoss.__init__/compute_loss (multigrid_loss.py:37-443)."""

    def __init__(self, anchors, num_classes, input_shape=(608, 608), ignore_thresh=0.5, label_smoothing=0.0,
                 loss_option=2, coord_scale=1.0, object_scale=1.0, no_object_scale=1.0, class_scale=1.0,
                 anchor_scale=1.0, class_weights=None, loss_normalization=None, use_iou_aware_objectness=False,
                 iou_objectness_power=1.0, iou_objectness_ratio=1.0, trainable_nms_weight=0.0,
                 trainable_nms_power=2.0, use_consensus_loss=False, consensus_kernel_size=3,
                 consensus_iou_power=1.5, consensus_min_iou=1e-3, consensus_coord_scale=0.5,
                 consensus_obj_scale=0.5, consensus_class_scale=0.3, consensus_stop_gradient=True,
                 consensus_center_tolerance=1e-4, use_focal_loss=False, use_softmax_loss=False, use_giou_loss=False,
                 use_diou_loss=False, use_ciou_loss=False, focal_alpha=0.25, focal_gamma=2.0, compat="tf_ref",
                 dtype=torch.float32):
        self.anchors = [torch.as_tensor(a, dtype=dtype) for a in anchors]
        self.C = num_classes
        self.input_shape = input_shape
        self.dtype = dtype
        self.ignore_thresh = ignore_thresh
        self.label_smoothing = label_smoothing
        self.loss_option = loss_option
        self.coord_scale, self.object_scale, self.no_object_scale = coord_scale, object_scale, no_object_scale
        self.class_scale, self.anchor_scale = class_scale, anchor_scale
        if class_weights is not None and len(class_weights) != num_classes:
            raise ValueError(f"class_weights length ({len(class_weights)}) must match num_classes ({num_classes})")
        self.class_weights = torch.ones(num_classes, dtype=dtype) if class_weights is None \
            else torch.as_tensor(class_weights, dtype=dtype)
        self.norm = ["batch"] if loss_normalization is None else (
            loss_normalization if isinstance(loss_normalization, list) else [loss_normalization])
        self.iou_aware = use_iou_aware_objectness
        self.iou_pow = iou_objectness_power
        self.iou_ratio = float(min(max(iou_objectness_ratio, 0.0), 1.0))
        self.nms_w, self.nms_pow = float(trainable_nms_weight), trainable_nms_power
        self.consensus = use_consensus_loss
        if use_consensus_loss and (consensus_kernel_size % 2 == 0 or consensus_kernel_size < 1):
            raise ValueError("consensus_kernel_size must be an odd positive integer")
        self.ck, self.cpow, self.cmin = consensus_kernel_size, consensus_iou_power, consensus_min_iou
        self.ccs, self.cos, self.ccls = consensus_coord_scale, consensus_obj_scale, consensus_class_scale
        self.cstop, self.ctol = consensus_stop_gradient, consensus_center_tolerance
        self.focal, self.softmax = use_focal_loss, use_softmax_loss
        self.giou, self.diou, self.ciou = use_giou_loss, use_diou_loss, use_ciou_loss
        self.falpha, self.fgamma = focal_alpha, focal_gamma
        assert compat in ("tf_ref", "fixed")
        self.compat = compat

    # multigrid_loss.py:194-231
    def _norm(self, B, gh, gw, obj):
        f = 1.0
        for n in self.norm:
            if n == "positives":
                f = f * max(float(obj.sum()), 1.0)
            elif n == "batch":
                f = f * B
            elif n == "grid":
                f = f * (B * gh * gw)
        return max(f, 1.0)

    # multigrid_loss.py:494-703
    def _ignore(self, pxy, pwh, txy, twh, anchors, obj, ytl):
        B, gh, gw, _ = pxy.shape
        A = anchors.shape[0]
        dt = self.dtype
        gx = torch.arange(gw, dtype=dt)
        gy = torch.arange(gh, dtype=dt)
        gxm, gym = torch.meshgrid(gx, gy, indexing="ij")          # [gw, gh]; [i,j] -> (i, j)
        grid = torch.stack([gxm, gym], -1).unsqueeze(0)           # indexed as [row, col] (square grids)
        scale = torch.tensor([self.input_shape[1] / gw, self.input_shape[0] / gh], dtype=dt)
        t_xy = (txy + grid) * scale
        aidx = torch.argmax(ytl[..., 5:5 + A], -1)
        onehot = torch.nn.functional.one_hot(aidx, A).to(dt)
        t_wh = torch.exp(twh) * (onehot @ anchors) * scale
        p_xy = (xy_act(pxy) + grid) * scale                       # [B,gh,gw,2]
        p_wh = torch.exp(pwh).unsqueeze(-2) * anchors.view(1, 1, 1, A, 2) * scale   # [B,gh,gw,A,2]
        iou_all = torch.zeros(B, gh, gw, A, dtype=dt)
        for b in range(B):
            m = obj[b, ..., 0] > 0.5
            if not bool(m.any()):
                continue
            gxy, gwh = t_xy[b][m], t_wh[b][m]                     # [n,2]
            pmin = (p_xy[b].unsqueeze(-2) - p_wh[b] / 2.0).unsqueeze(-2)   # [gh,gw,A,1,2]
            pmax = (p_xy[b].unsqueeze(-2) + p_wh[b] / 2.0).unsqueeze(-2)
            gmin, gmax = gxy - gwh / 2.0, gxy + gwh / 2.0
            iwh = torch.clamp(torch.minimum(pmax, gmax) - torch.maximum(pmin, gmin), min=0.0)
            inter = iwh[..., 0] * iwh[..., 1]
            parea = (p_wh[b][..., 0] * p_wh[b][..., 1]).unsqueeze(-1)
            garea = gwh[:, 0] * gwh[:, 1]
            iou = inter / (parea + garea - inter + EPS)
            iou_all[b] = iou.max(-1).values
        max_iou = iou_all.max(-1).values
        ignore = ((max_iou > self.ignore_thresh) & (obj[..., 0] < 0.5)).to(dt).unsqueeze(-1)
        assigned = (iou_all * onehot).sum(-1, keepdim=True) * obj
        return ignore.detach(), assigned.detach(), max_iou.unsqueeze(-1).detach()

    def components(self, y_true, y_pred):
        dt = self.dtype
        B = y_pred[0].shape[0]
        tot = dict(loc=0.0, obj=0.0, anchor=0.0, cls=0.0, ccoord=0.0, cobj=0.0, ccls=0.0)
        for l in range(len(self.anchors)):
            yp, yt = y_pred[l].to(dt), torch.as_tensor(y_true[l], dtype=dt)
            anc = self.anchors[l]
            A = anc.shape[0]
            pxy, pwh, pobj, panc, pcls = yp[..., 0:2], yp[..., 2:4], yp[..., 4:5], yp[..., 5:5 + A], yp[..., 5 + A:]
            txy, twh, tobj, tanc, tcls = yt[..., 0:2], yt[..., 2:4], yt[..., 4:5], yt[..., 5:5 + A], yt[..., 5 + A:]
            obj = (tobj > 0.5).to(dt)
            gh, gw = yp.shape[1], yp.shape[2]
            ignore, assigned, maxiou = self._ignore(pxy, pwh, txy, twh, anc, obj, yt)
            nf = self._norm(B, gh, gw, obj)
            # localisation (:729-757); options 1, 2 and 3-without-flags are all MSE
            fn = None
            if self.loss_option == 3:      # first flag set wins (:353-364)
                fn = giou_loss if self.giou else diou_loss if self.diou else ciou_loss if self.ciou else None
            if fn is None:
                loc = ((((txy - xy_act(pxy)) ** 2).sum(-1, keepdim=True) +
                        ((twh - pwh) ** 2).sum(-1, keepdim=True)) * obj).sum() / nf
            elif self.compat == "tf_ref":
                loc = fn(txy, twh, pxy, pwh, obj) / nf          # raw tensors, [B,H,W] * [B,H,W,1] broadcast
            else:
                # repaired form: per-cell mask, boxes in grid-cell units (wh = exp(t) * assigned anchor / stride)
                aidx = torch.argmax(tanc, -1)
                stride = torch.tensor([self.input_shape[1] / gw, self.input_shape[0] / gh], dtype=dt)
                awh = anc[aidx] / stride
                kw = {} if fn is giou_loss else {"keepdim": False}
                loc = fn(txy, torch.exp(twh) * awh, xy_act(pxy), torch.exp(pwh) * awh, obj[..., 0], **kw) / nf
            tot["loc"] = tot["loc"] + loc
            # anchor (:759-799), pre-multiplied by anchor_scale (:349 / :390)
            al = (bce_logits(tanc, panc) * obj * (1.0 - ignore)).sum() / nf
            tot["anchor"] = tot["anchor"] + self.anchor_scale * al
            # objectness (:861-928)
            tgt = tobj
            if self.iou_aware:
                piou = torch.clamp(assigned, 0.0, 1.0)
                blended = self.iou_ratio * torch.pow(piou + EPS, self.iou_pow) + (1.0 - self.iou_ratio) * tobj
                tgt = obj * blended + (1.0 - obj) * tgt
            w = obj * self.object_scale + (1.0 - obj) * (1.0 - ignore) * self.no_object_scale
            if self.nms_w > 0.0:
                w = w + (1.0 - obj) * ignore * self.nms_w * torch.pow(torch.clamp(maxiou, 0.0, 1.0) + EPS, self.nms_pow)
            tot["obj"] = tot["obj"] + (bce_logits(tgt, pobj) * w).sum() / nf
            # classification (:829-859)
            if self.softmax:                 # :815-828 ([B,H,W] * [B,H,W,1] * [1,1,1,C] in the reference)
                fl = softmax_focal(tcls, pcls, self.fgamma)
                if self.compat == "tf_ref":
                    cl = (fl * obj * self.class_weights.view(1, 1, 1, -1)).sum() / nf
                else:
                    cl = (fl * obj[..., 0] * (tcls * self.class_weights).sum(-1)).sum() / nf
            elif self.focal:                 # :800-813
                cl = (sigmoid_focal(tcls, pcls, self.falpha, self.fgamma) * obj * self.class_weights).sum() / nf
            else:
                ts = tcls * (1.0 - self.label_smoothing) + self.label_smoothing / self.C if self.label_smoothing > 0 else tcls
                cl = (bce_logits(ts, pcls) * self.class_weights * obj).sum() / nf
            tot["cls"] = tot["cls"] + cl
            if self.consensus:
                cc, co, ccl = self._consensus(pxy, pwh, pobj, pcls, txy, obj, assigned)
                tot["ccoord"], tot["cobj"], tot["ccls"] = tot["ccoord"] + cc, tot["cobj"] + co, tot["ccls"] + ccl
        return tot

    # multigrid_loss.py:930-1043
    def _consensus(self, pxy, pwh, pobj, pcls, txy, obj, assigned):
        dt = self.dtype
        B, gh, gw, _ = pxy.shape
        cmask = ((txy[..., 0] >= 0) & (txy[..., 0] < 1) & (txy[..., 1] >= 0) & (txy[..., 1] < 1)).to(dt).unsqueeze(-1) * obj
        gxm, gym = torch.meshgrid(torch.arange(gw, dtype=dt), torch.arange(gh, dtype=dt), indexing="ij")
        centers = txy + torch.stack([gxm, gym], -1).unsqueeze(0)
        k = self.ck
        mp, ip, cp = _patches(obj, k), _patches(assigned, k), _patches(centers, k)
        same = ((cp - centers.unsqueeze(3)).abs().max(-1, keepdim=True).values < self.ctol).to(dt)
        gm = mp * same * cmask.unsqueeze(3)
        vw = torch.where(gm > 0, torch.clamp(ip, min=self.cmin), torch.zeros_like(ip))
        raw = torch.pow(vw, self.cpow) * gm
        wts = raw / (raw.sum(3, keepdim=True) + EPS)
        ws = wts.squeeze(-1)
        normalizer = max(float(cmask.sum()), 1.0)

        def var(t):
            pt = _patches(t, k)
            cons = (wts * pt).sum(3)
            if self.cstop:
                cons = cons.detach()
            return pt - cons.unsqueeze(3)

        bd = var(torch.cat([pxy, pwh], -1))
        coord = (ws * (bd ** 2).sum(-1)).sum() / normalizer
        od = var(torch.sigmoid(pobj))
        objv = (ws * (od ** 2).squeeze(-1)).sum() / normalizer
        cd = var(torch.sigmoid(pcls))
        clsv = (ws.unsqueeze(-1) * cd ** 2).sum() / (normalizer * float(self.C))
        return coord, objv, clsv

    def total(self, tot):
        t = (self.coord_scale * tot["loc"] + self.object_scale * tot["obj"] +
             self.anchor_scale * tot["anchor"] + self.class_scale * tot["cls"])
        if self.consensus:
            t = t + self.ccs * tot["ccoord"] + self.cos * tot["cobj"] + self.ccls * tot["ccls"]
        return t

    def __call__(self, y_true, y_pred):
        return self.total(self.components(y_true, y_pred))

    def value_and_grad(self, y_true, y_pred):
        """Returns (total, components dict of floats, [d total / d y_pred[l]])."""
        yp = [torch.as_tensor(p, dtype=self.dtype).clone().requires_grad_(True) for p in y_pred]
        comp = self.components(y_true, yp)
        total = self.total(comp)
        total.backward()
        return float(total.detach()), {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in comp.items()}, \
            [p.grad.detach().numpy() for p in yp]
