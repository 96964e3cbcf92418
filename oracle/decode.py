"""CPU oracle for decode -> correct_boxes -> confidence filter -> NMS -> xyxy.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, as the checker.  The product path never imports `oracle`.

Restates (numpy float32, same operation order):
  P1  MultiGridDecoder._decode_single_scale   multigriddet/postprocess/multigrid_decode.py:100-183
  P2  correct_boxes                           :185-235
  P3  handle_predictions / _filter_boxes      :237-345
  P4  StandardNMS / DIoUNMS / ClusterNMS      multigriddet/postprocess/nms.py:83-231, 320-385
      SoftNMS                                 :234-317
  P5  _convert_to_xyxy                        multigrid_decode.py:397-422
Pinned by tests/golden/decode_*.npz and tests/golden/nms.npz (outputs of the reference's own
numpy code on seeded inputs; generator: tests/golden/make_golden.py).
"""
import numpy as np


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32) if x.dtype == np.float32 \
        else 1.0 / (1.0 + np.exp(-x))


def _softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def decode_single_scale(pred, anchors, num_classes, input_shape, rescore=True, use_softmax=True):
    """P1.  Grid is (col,row) (:119-129); x is divided by grid_h and y by grid_w, w by input_h and
    h by input_w (:116,:155,:163) -- identical for square inputs, reproduced as written."""
    pred = np.asarray(pred, np.float32)
    B, gh, gw = pred.shape[0], pred.shape[1], pred.shape[2]
    A = len(anchors)
    anchors = np.asarray(anchors, np.float32)
    xo, yo = np.meshgrid(np.arange(gw), np.arange(gh))
    cell = np.stack([xo, yo], -1).reshape(1, gh, gw, 2)
    raw_xy, raw_wh = pred[..., 0:2], pred[..., 2:4]
    obj = _sigmoid(pred[..., 4:5])
    ap, cp = pred[..., 5:5 + A], pred[..., 5 + A:]
    if use_softmax:
        ap, cp = _softmax(ap), _softmax(cp)
    else:
        ap, cp = _sigmoid(ap), _sigmoid(cp)
    act = np.tanh(np.float32(0.15) * raw_xy) + _sigmoid(np.float32(0.15) * raw_xy)
    box_xy = (act + cell) / np.array([gh, gw])
    aidx = np.argmax(ap, -1)
    box_wh = anchors[aidx] * np.exp(raw_wh) / np.array(input_shape)
    if rescore:
        obj = obj * ap.max(-1, keepdims=True) * cp.max(-1, keepdims=True)
    out = np.concatenate([box_xy, box_wh, obj, cp], -1)
    return out.reshape(B, gh * gw, num_classes + 5)


def decode_predictions(preds, anchors, num_classes, input_shape, **kw):
    if len(preds) != len(anchors):
        raise ValueError(f"Expected {len(anchors)} predictions, got {len(preds)}")
    return np.concatenate([decode_single_scale(p, a, num_classes, input_shape, **kw)
                           for p, a in zip(preds, anchors)], 1)


def correct_boxes(pred, image_shape, model_image_size):
    """P2 (:185-235): undo the letterbox; output top-left xywh in image pixels."""
    box_xy, box_wh = pred[..., 0:2].copy(), pred[..., 2:4].copy()
    msize = np.array(model_image_size, "float32")
    ishape = np.array(image_shape, "float32")
    new_shape = np.round(ishape * np.min(msize / ishape))
    offset = ((msize - new_shape) / 2.0 / msize)[::-1]
    scale = (msize / new_shape)[::-1]
    box_xy = (box_xy - offset) * scale
    box_wh = box_wh * scale
    box_xy = box_xy - box_wh / 2.0
    iwh = ishape[::-1]
    return np.concatenate([box_xy * iwh, box_wh * iwh, pred[..., 4:5], pred[..., 5:]], -1)


def _iou_1vN(b, bs):
    x1, y1, w1, h1 = b
    x2, y2, w2, h2 = bs[:, 0], bs[:, 1], bs[:, 2], bs[:, 3]
    iw = np.maximum(0.0, np.minimum(x1 + w1, x2 + w2) - np.maximum(x1, x2))
    ih = np.maximum(0.0, np.minimum(y1 + h1, y2 + h2) - np.maximum(y1, y2))
    inter = iw * ih
    union = w1 * h1 + w2 * h2 - inter
    return inter / (union + 1e-8), (x1, y1, w1, h1, x2, y2, w2, h2)


def _diou_1vN(b, bs):
    """nms.py:189-231."""
    iou, (x1, y1, w1, h1, x2, y2, w2, h2) = _iou_1vN(b, bs)
    cd = ((x1 + w1 / 2) - (x2 + w2 / 2)) ** 2 + ((y1 + h1 / 2) - (y2 + h2 / 2)) ** 2
    ed = (np.maximum(x1 + w1, x2 + w2) - np.minimum(x1, x2)) ** 2 + \
         (np.maximum(y1 + h1, y2 + h2) - np.minimum(y1, y2)) ** 2
    return iou - cd / (ed + 1e-8)


def greedy_nms(boxes, scores, thr, method="diou"):
    """Class-agnostic greedy NMS (nms.py:154-187 / :86-118 / :323-355): order =
    argsort(scores)[::-1]; suppress metric >= thr.  Returns kept indices in selection order."""
    order = np.argsort(scores)[::-1]
    keep = []
    metric = _diou_1vN if method == "diou" else (lambda b, bs: _iou_1vN(b, bs)[0])
    while len(order) > 0:
        cur = order[0]
        keep.append(cur)
        if len(order) == 1:
            break
        m = metric(boxes[cur], boxes[order[1:]])
        order = order[1:][m < thr]
    return np.array(keep, dtype=np.int64)


def soft_nms(boxes, scores, sigma=0.5, score_threshold=0.001):
    """SoftNMS.apply_nms (nms.py:248-287): fixed order argsort(scores)[::-1] (never re-sorted);
    box i decays every later box by exp(-iou^2/sigma) unless its own decayed score has fallen
    below `score_threshold` (then it is zeroed and skipped); survivors are returned in ORIGINAL
    index order with their decayed scores.  Returns (keep_mask, soft_scores)."""
    order = np.argsort(scores)[::-1]
    soft = scores.copy()
    for i in range(len(order)):
        cur = order[i]
        if soft[cur] < score_threshold:
            soft[cur] = 0
            continue
        rest = order[i + 1:]
        if len(rest) == 0:
            break
        iou = _iou_1vN(boxes[cur], boxes[rest])[0]
        soft[rest] *= np.exp(-iou ** 2 / sigma)
    return soft >= score_threshold, soft


def convert_to_xyxy(boxes, image_shape):
    """P5 (:397-422)."""
    out = boxes.copy()
    out[:, 2] = boxes[:, 0] + boxes[:, 2]
    out[:, 3] = boxes[:, 1] + boxes[:, 3]
    h, w = image_shape[0], image_shape[1]
    out[:, 0] = np.clip(out[:, 0], 0, w)
    out[:, 1] = np.clip(out[:, 1], 0, h)
    out[:, 2] = np.clip(out[:, 2], 0, w)
    out[:, 3] = np.clip(out[:, 3], 0, h)
    return np.floor(out + 0.5).astype("int32")


def greedy_nms_per_scale(boxes, scores, seg, thr, method="diou"):
    """The build's per-scale option (no reference counterpart; multigrid_decode.py:98 concatenates the scales first):
    global score order, a kept box suppresses only later boxes of its own scale."""
    order = np.argsort(scores)[::-1]
    keep = []
    metric = _diou_1vN if method == "diou" else (lambda b, bs: _iou_1vN(b, bs)[0])
    while len(order) > 0:
        cur = order[0]
        keep.append(cur)
        if len(order) == 1:
            break
        rest = order[1:]
        m = metric(boxes[cur], boxes[rest])
        order = rest[(m < thr) | (seg[rest] != seg[cur])]
    return np.array(keep, dtype=np.int64)


def postprocess(outputs, anchors, num_classes, input_shape, image_shape, model_image_size, max_boxes=100,
                confidence=0.1, nms_threshold=0.5, nms_method="diou", return_xyxy=True, per_scale=False):
    """MultiGridDecoder.postprocess (:347-395) for nms_method in {'diou','cluster','soft'} (cluster
    == standard IoU greedy in the reference, nms.py:320-385)."""
    pred = decode_predictions(outputs, anchors, num_classes, input_shape)
    pred = correct_boxes(pred, image_shape, model_image_size)
    boxes, conf, cprob = pred[..., 0:4], pred[..., 4], pred[..., 5:]
    classes = np.argmax(cprob, -1)
    pos = np.where(conf >= confidence)
    if len(pos[0]) == 0:
        return np.array([]), np.array([]), np.array([])
    boxes, classes, scores = boxes[pos], classes[pos], conf[pos]
    if per_scale:
        cells = [int(np.prod(o.shape[1:3])) for o in outputs]
        seg_all = np.concatenate([np.full(n, l) for l, n in enumerate(cells)])[None, :]
        keep = greedy_nms_per_scale(boxes, scores, seg_all[pos], nms_threshold, "diou" if nms_method == "diou" else "iou")
        boxes, classes, scores = boxes[keep], classes[keep].astype("int32"), scores[keep]
    elif nms_method == "soft":
        mask, soft = soft_nms(boxes, scores)
        boxes, classes, scores = boxes[mask], classes[mask].astype("int32"), soft[mask]
    else:
        keep = greedy_nms(boxes, scores, nms_threshold, "diou" if nms_method == "diou" else "iou")
        boxes, classes, scores = boxes[keep], classes[keep].astype("int32"), scores[keep]
    if len(boxes) > max_boxes:
        top = np.argsort(scores)[::-1][:max_boxes]
        boxes, classes, scores = boxes[top], classes[top], scores[top]
    if return_xyxy and len(boxes) > 0:
        boxes = convert_to_xyxy(boxes, image_shape)
    return boxes, classes, scores
