"""MultiGridLoss with the reference's constructor signature, computed by mgd_loss_fwd_bwd on gfx950.

Mirrors `MultiGridLoss(anchors, num_classes, input_shape, ...35 kwargs...)(y_true, y_pred) -> scalar`
(reference multigriddet/losses/multigrid_loss.py:37-73, 190).  Same argument names, defaults and
ValueErrors; tensors are torch CUDA tensors (numpy inputs are uploaded).  Differences, stated:
 * one extra keyword, `compat` ("tf_ref" default | "fixed"), for the two branches whose reference code is
   shape-broken: GIoU/DIoU/CIoU localisation (loss_option=3 + use_giou_loss / use_diou_loss / use_ciou_loss,
   losses/iou_losses.py:36-237) and use_softmax_loss (losses/focal_loss.py:80-114).  Both multiply a [B,H,W]
   per-cell loss by the [B,H,W,1] object mask (iou_losses.py:70-93, multigrid_loss.py:815-828).  "tf_ref" computes
   exactly what TensorFlow's broadcasting makes of that - defined for square grids and B == 1 or B == H, a
   ValueError otherwise (TensorFlow raises InvalidArgumentError there; with three scales only B == 1 passes) - on
   the raw offset / log-ratio tensors, as the reference feeds them.  "fixed" applies the mask per cell to boxes
   decoded to grid-cell units.  With loss_option=3 and no IoU flag the reference falls back to MSE
   (multigrid_loss.py:365-368) and so does this.
 * consensus_kernel_size must be 3 on the device path.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops


class MultiGridLoss:
    __name__ = "MultiGridLoss"

    def __init__(self, anchors: List[np.ndarray], num_classes: int, input_shape: Tuple[int, int] = (608, 608),
                 ignore_thresh: float = 0.5, label_smoothing: float = 0.0, elim_grid_sense: bool = False,
                 use_focal_loss: bool = False, use_softmax_loss: bool = False, use_iol: bool = True,
                 use_giou_loss: bool = False, use_diou_loss: bool = False, use_ciou_loss: bool = False,
                 loss_option: int = 2, focal_alpha: float = 0.25, focal_gamma: float = 2.0,
                 coord_scale: float = 1.0, object_scale: float = 1.0, no_object_scale: float = 1.0,
                 class_scale: float = 1.0, anchor_scale: float = 1.0, class_weights: Optional[np.ndarray] = None,
                 loss_normalization: Optional[List[str]] = None, use_iou_aware_objectness: bool = False,
                 iou_objectness_power: float = 1.0, iou_objectness_ratio: float = 1.0,
                 trainable_nms_weight: float = 0.0, trainable_nms_power: float = 2.0,
                 use_consensus_loss: bool = False, consensus_kernel_size: int = 3,
                 consensus_iou_power: float = 1.5, consensus_min_iou: float = 1e-3,
                 consensus_coord_scale: float = 0.5, consensus_obj_scale: float = 0.5,
                 consensus_class_scale: float = 0.3, consensus_stop_gradient: bool = True,
                 consensus_center_tolerance: float = 1e-4, compat: str = "tf_ref"):
        self.anchors = [np.asarray(a, np.float32) for a in anchors]
        self.num_classes = num_classes
        self.input_shape = tuple(input_shape)
        self.num_layers = len(anchors)
        if class_weights is not None and len(class_weights) != num_classes:
            raise ValueError(f"class_weights length ({len(class_weights)}) must match num_classes ({num_classes})")
        if use_consensus_loss and (consensus_kernel_size % 2 == 0 or consensus_kernel_size < 1):
            raise ValueError("consensus_kernel_size must be an odd positive integer")
        if use_consensus_loss and consensus_kernel_size != 3:
            raise NotImplementedError("device consensus loss supports consensus_kernel_size=3 only")
        if compat not in ("tf_ref", "fixed"):
            raise ValueError(f"compat must be 'tf_ref' or 'fixed', got {compat!r}")
        self.compat = compat
        self._broadcast = compat == "tf_ref" and (use_softmax_loss or (
            loss_option == 3 and (use_giou_loss or use_diou_loss or use_ciou_loss)))
        self._softmax_ref = compat == "tf_ref" and use_softmax_loss
        self.class_weights = None if class_weights is None else np.asarray(class_weights, np.float32)
        if loss_normalization is None:
            loss_normalization = ["batch"]
        if not isinstance(loss_normalization, list):
            loss_normalization = [loss_normalization]
        self.kw = dict(ignore_thresh=ignore_thresh, label_smoothing=label_smoothing, loss_option=loss_option,
                       coord_scale=coord_scale, object_scale=object_scale, no_object_scale=no_object_scale,
                       class_scale=class_scale, anchor_scale=anchor_scale, loss_normalization=loss_normalization,
                       use_iou_aware_objectness=use_iou_aware_objectness, iou_objectness_power=iou_objectness_power,
                       iou_objectness_ratio=iou_objectness_ratio, trainable_nms_weight=trainable_nms_weight,
                       trainable_nms_power=trainable_nms_power, use_consensus_loss=use_consensus_loss,
                       consensus_iou_power=consensus_iou_power, consensus_min_iou=consensus_min_iou,
                       consensus_coord_scale=consensus_coord_scale, consensus_obj_scale=consensus_obj_scale,
                       consensus_class_scale=consensus_class_scale, consensus_stop_gradient=consensus_stop_gradient,
                       consensus_center_tolerance=consensus_center_tolerance, use_focal_loss=use_focal_loss,
                       focal_alpha=focal_alpha, focal_gamma=focal_gamma, use_softmax_loss=use_softmax_loss,
                       use_giou_loss=use_giou_loss, use_diou_loss=use_diou_loss, use_ciou_loss=use_ciou_loss,
                       compat=compat)
        self.loss_option, self.loss_normalization = loss_option, loss_normalization
        self._runners = {}

    def _dev(self, t):
        if isinstance(t, torch.Tensor):
            return t.to("cuda", torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(t, np.float32)).cuda()

    def _runner(self, y_pred):
        B = y_pred[0].shape[0]
        grids = tuple((int(p.shape[1]), int(p.shape[2])) for p in y_pred)
        key = (B, grids)
        if self._broadcast:
            for gh, gw in grids:        # what TensorFlow's broadcast of [B,H,W] * [B,H,W,1] accepts
                if gh != gw or B not in (1, gh):
                    raise ValueError(f"Incompatible shapes: [{B},{gh},{gw}] vs. [{B},{gh},{gw},1] (the reference's "
                                     f"IoU / softmax-focal branch broadcasts only for square grids and B == 1 or "
                                     f"B == H; pass compat='fixed')")
                if self._softmax_ref and self.num_classes not in (1, gw):
                    raise ValueError(f"Incompatible shapes: [{B},{gh},{gw},{gw}] vs. [1,1,1,{self.num_classes}] "
                                     f"(reference class_weights broadcast; pass compat='fixed')")
        if key not in self._runners:
            cfg = ops.make_loss_cfg(self.anchors, self.num_classes, self.input_shape, B, grids, **self.kw)
            self._runners[key] = ops.LossRunner(cfg, y_pred[0].device, class_weights=self.class_weights)
        return self._runners[key]

    def compute_loss(self, y_true: Sequence, y_pred: Sequence) -> torch.Tensor:
        return self.components(y_true, y_pred)[7]

    def components(self, y_true, y_pred, grad_f32=None, grad_bf16=None) -> torch.Tensor:
        """Device tensor [loc, obj, anchor, cls, consensus_coord, consensus_obj, consensus_cls, total]."""
        if len(y_pred) != self.num_layers or len(y_true) != self.num_layers:
            raise ValueError(f"Expected {self.num_layers} scales, got {len(y_pred)} / {len(y_true)}")
        yp = [self._dev(p) for p in y_pred]
        yt = [self._dev(t) for t in y_true]
        return self._runner(yp).run(yt, yp, grad_f32=grad_f32, grad_bf16=grad_bf16).clone()

    def value_and_grad(self, y_true, y_pred):
        yp = [self._dev(p) for p in y_pred]
        g = [torch.empty_like(p) for p in yp]
        c = self.components(y_true, yp, grad_f32=g)
        return c[7], g

    def __call__(self, y_true, y_pred) -> torch.Tensor:
        return self.compute_loss(y_true, y_pred)


def multigriddet_loss(args, anchors, num_classes, **kwargs):
    """Keras-Lambda style wrapper (reference multigrid_loss.py:1046-1064): args = outputs + y_true."""
    n = len(anchors)
    return MultiGridLoss(anchors=anchors, num_classes=num_classes, **kwargs).compute_loss(args[n:], args[:n])
