"""MultiGridDecoder with the reference's interface, running decode + NMS on gfx950.

Mirrors `MultiGridDecoder(anchors, num_classes, input_shape, rescore_confidence, use_softmax)` and
`.postprocess(outputs, image_shape, model_image_size, max_boxes, confidence, nms_threshold, use_iol,
nms_method, use_wbf, return_xyxy) -> (boxes int32 (N,4) xyxy, classes int32 (N,), scores (N,))`
(reference multigriddet/postprocess/multigrid_decode.py:25-30, 347-395); empty results are three
empty arrays (:274).  `postprocess_batch` is the batched device entry the inference/eval loops use
(no per-image device->host copy before NMS, unlike evaluator.py:257).
nms_method 'diou' / 'cluster' / 'soft' and use_wbf=True all run on the device (mgd_nms methods 1 / 0 / 2, mgd_wbf);
any other nms_method raises NotImplementedError exactly as the reference's abstract NMS does
(multigrid_decode.py:297).
"""
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .. import ops


class MultiGridDecoder:
    def __init__(self, anchors: List[np.ndarray], num_classes: int, input_shape: Tuple[int, int] = (608, 608),
                 rescore_confidence: bool = True, use_softmax: bool = True):
        self.anchors = [np.asarray(a, np.float32) for a in anchors]
        self.num_classes = num_classes
        self.input_shape = tuple(input_shape)
        self.rescore_confidence = rescore_confidence
        self.use_softmax = use_softmax
        self.num_layers = len(anchors)

    def _dev(self, t):
        if isinstance(t, torch.Tensor):
            return t.to("cuda", torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(t, np.float32)).cuda()

    def postprocess_batch(self, outputs: Sequence, image_shapes, max_boxes=100, confidence=0.1, nms_threshold=0.5,
                          nms_method="diou", return_xyxy=True, use_wbf=False, per_scale_nms=False):
        """outputs: L tensors [B,g,g,F]; image_shapes: [B,2] (h,w).  Returns device tensors
        (boxes [B,max_boxes,4], scores [B,max_boxes], classes [B,max_boxes], count [B])."""
        if len(outputs) != self.num_layers:
            raise ValueError(f"Expected {self.num_layers} predictions, got {len(outputs)}")
        if not use_wbf and (nms_method not in ops.NMS_METHODS or nms_method in ("standard", "iou")):
            # the reference maps only 'diou' / 'soft' / 'cluster'; anything else reaches the abstract NMS.apply_nms
            raise NotImplementedError("Subclasses must implement apply_nms method")
        outs = [self._dev(o) for o in outputs]
        B = outs[0].shape[0]
        grids = [(int(o.shape[1]), int(o.shape[2])) for o in outs]
        ihw = self._image_hw(image_shapes, B)
        cfg = ops.make_decode_cfg(self.anchors, self.num_classes, self.input_shape, B, grids, confidence,
                                  use_softmax=self.use_softmax, rescore=self.rescore_confidence,
                                  tag_scale=per_scale_nms)
        if per_scale_nms and (use_wbf or nms_method == "soft"):
            raise ValueError("per_scale_nms applies to the greedy NMS methods ('diou', 'cluster')")
        b, s, c, n = ops.decode(cfg, outs, ihw)
        return ops.nms(b, s, c, n, ihw, method="wbf" if use_wbf else nms_method, threshold=nms_threshold,
                       max_boxes=max_boxes, return_xyxy=return_xyxy, per_scale=per_scale_nms)

    def _image_hw(self, image_shapes, B):
        """[B,2] fp32 device tensor of the original image sizes; the upload of a shape set seen before is re-used (a stream of
        frames from one camera pays the host-to-device copy once)."""
        arr = np.asarray(image_shapes, np.float32).reshape(B, 2)
        key = arr.tobytes()
        cache = self.__dict__.setdefault("_ihw_cache", {})
        t = cache.get(key)
        if t is None:
            if len(cache) >= 64:
                cache.clear()
            t = cache[key] = torch.as_tensor(arr).cuda()
        return t

    def postprocess(self, multigriddet_outputs, image_shape, model_image_size, max_boxes: int = 100,
                    confidence: float = 0.1, nms_threshold: float = 0.5, use_iol: bool = True,
                    nms_method: str = "diou", use_wbf: bool = False, return_xyxy: bool = True):
        if tuple(model_image_size) != tuple(self.input_shape):
            self.input_shape = tuple(model_image_size)
        ob, osc, ocl, ocn = self.postprocess_batch(multigriddet_outputs, [image_shape], max_boxes, confidence,
                                                   nms_threshold, nms_method, return_xyxy, use_wbf=use_wbf)
        k = int(ocn[0])
        if k == 0:
            return np.array([]), np.array([]), np.array([])
        return ob[0, :k].cpu().numpy(), ocl[0, :k].cpu().numpy().astype("int32"), osc[0, :k].cpu().numpy()
