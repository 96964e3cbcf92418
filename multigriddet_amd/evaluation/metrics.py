"""mAP metrics with the reference's interface (reference multigriddet/evaluation/metrics.py:28-864); the IoU matrix
and the greedy prediction -> ground-truth matching run on gfx950 (mgd_iou_matrix, mgd_eval_match: one block per
(image, class) group, one wavefront per IoU threshold, all classes and thresholds in ONE launch instead of the
reference's process pool), the precision/recall integration is a few numpy lines per class on the host.

Faithful to the reference including its two IoU conventions: the cached path (calculate_iou_matrix) treats boxes as
xyxy; the un-cached path (match_predictions_to_gt -> BoxUtils.box_iou, utils/boxes.py:16-57) reads the same numbers
as (cx, cy, w, h).  calculate_map takes the un-cached path when cache_ious is false, when use_parallel meets more
than 10 000 predictions (:592-595), and ALWAYS for the per-scale APS/APM/APL sub-runs (:756-812).  Pass
strict_xyxy=True (not in the reference) to use the xyxy IoU everywhere."""
import ctypes as C
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

from .. import _lib as L

COCO_IOU_THRESHOLDS = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]


def _dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype)).cuda()


def calculate_iou_matrix(boxes1: np.ndarray, boxes2: np.ndarray) -> np.ndarray:
    """(N,4) x (M,4) xyxy -> (N,M) IoU, float64 (metrics.py:28-71)."""
    boxes1, boxes2 = np.asarray(boxes1, np.float64), np.asarray(boxes2, np.float64)
    if len(boxes1) == 0 or len(boxes2) == 0:
        return np.zeros((len(boxes1), len(boxes2)))
    if boxes1.shape[1] != 4 or boxes2.shape[1] != 4:
        raise ValueError("Boxes must have 4 coordinates")
    L.require_gpu()
    b1, b2 = _dev(boxes1, np.float64), _dev(boxes2, np.float64)
    out = torch.empty(len(boxes1), len(boxes2), dtype=torch.float64, device="cuda")
    L.check(L.load().mgd_iou_matrix(L.ptr(b1), L.ptr(b2), L.ptr(out), len(boxes1), len(boxes2), 0, L.stream_ptr()),
            "iou_matrix")
    return out.cpu().numpy()


def match_all(predictions: List[Dict], ground_truths: List[Dict], iou_thresholds: List[float], quirk: bool):
    """Device matching of every class at every threshold.  Returns (order, tp) with `order` the prediction indices
    in the device layout (grouped by class, image; descending score inside a group) and tp[t][k] the flag of
    predictions[order[k]]."""
    P = len(predictions)
    if P == 0:
        return np.zeros(0, np.int64), np.zeros((len(iou_thresholds), 0), bool)
    L.require_gpu()
    pc = np.array([p["class"] for p in predictions], np.int64)
    pi = np.array([p["image_id"] for p in predictions], np.int64)
    ps = np.array([p["score"] for p in predictions], np.float64)
    order = np.lexsort((-ps, pi, pc))                       # class, image, score desc
    gc = np.array([g["class"] for g in ground_truths], np.int64) if ground_truths else np.zeros(0, np.int64)
    gi = np.array([g["image_id"] for g in ground_truths], np.int64) if ground_truths else np.zeros(0, np.int64)
    gorder = np.lexsort((np.arange(len(gc)), gi, gc))       # class, image, original order (first-maximum ties)
    pkey = np.stack([pc[order], pi[order]], 1)
    gkey = np.stack([gc[gorder], gi[gorder]], 1) if len(gc) else np.zeros((0, 2), np.int64)
    groups, pstart = np.unique(pkey, axis=0, return_index=True)
    pstart = np.append(pstart, P).astype(np.int32)
    gstart = np.zeros(len(groups) + 1, np.int32)
    gsel = []
    gdict = {}
    for j, k in enumerate(map(tuple, gkey)):
        gdict.setdefault(k, []).append(gorder[j])
    for gidx, k in enumerate(map(tuple, groups)):
        lst = gdict.get(k, [])
        gsel.extend(lst)
        gstart[gidx + 1] = gstart[gidx] + len(lst)
    pb = np.array([predictions[i]["bbox"] for i in order], np.float64).reshape(-1, 4)
    gb = (np.array([ground_truths[i]["bbox"] for i in gsel], np.float64).reshape(-1, 4) if gsel
          else np.zeros((1, 4), np.float64))
    NT = len(iou_thresholds)
    tp_all = np.zeros((NT, P), bool)
    lib = L.load()
    d_pb, d_ps, d_gb, d_gs = _dev(pb, np.float64), _dev(pstart, np.int32), _dev(gb, np.float64), _dev(gstart, np.int32)
    maxg = int(np.diff(gstart).max()) if len(groups) else 0
    for t0 in range(0, NT, 16):                             # 16 thresholds (wavefronts) per launch
        ths = np.asarray(iou_thresholds[t0:t0 + 16], np.float64)
        d_th = _dev(ths, np.float64)
        tp = torch.zeros(len(ths), P, dtype=torch.uint8, device="cuda")
        L.check(lib.mgd_eval_match(L.ptr(d_pb), L.ptr(d_ps), L.ptr(d_gb), L.ptr(d_gs), len(groups), maxg, L.ptr(d_th),
                                   len(ths), int(bool(quirk)), L.ptr(tp), C.c_longlong(P), L.stream_ptr()), "eval_match")
        tp_all[t0:t0 + len(ths)] = tp.cpu().numpy().astype(bool)
    return order, tp_all


def compute_precision_recall(tp_flags: np.ndarray, fp_flags: np.ndarray, num_gt: int) -> Tuple[np.ndarray, np.ndarray]:
    """metrics.py:221-248."""
    if len(tp_flags) == 0:
        return np.array([0.0]), np.array([0.0])
    cum_tp, cum_fp = np.cumsum(tp_flags), np.cumsum(fp_flags)
    return cum_tp / (cum_tp + cum_fp + 1e-8), cum_tp / (num_gt + 1e-8)


def compute_average_precision(precisions: np.ndarray, recalls: np.ndarray, method: str = "coco") -> float:
    """metrics.py:251-304: 'voc' = 11-point, 'coco' = all-point interpolation + trapezoid."""
    if len(precisions) == 0 or len(recalls) == 0:
        return 0.0
    if method == "voc":
        vals = []
        for r in np.arange(0, 1.1, 0.1):
            v = precisions[recalls >= r]
            vals.append(np.max(v) if len(v) > 0 else 0.0)
        return float(np.mean(vals))
    if method == "coco":
        o = np.argsort(recalls)
        rs, ps = recalls[o], precisions[o]
        interp = np.maximum.accumulate(ps[::-1])[::-1]
        if len(rs) > 1:
            trapz = getattr(np, "trapezoid", None) or np.trapz
            return float(trapz(interp, rs))
        return float(interp[0] * rs[0])
    raise ValueError(f"Unknown method: {method}")


def get_active_classes(predictions: List[Dict], ground_truths: List[Dict], num_classes: int) -> List[int]:
    return sorted({p["class"] for p in predictions} | {g["class"] for g in ground_truths})


def calculate_box_area(bbox) -> float:
    x1, y1, x2, y2 = bbox
    return (x2 - x1) * (y2 - y1)


def filter_by_area(predictions, ground_truths, min_area=None, max_area=None):
    """metrics.py:426-453 (min inclusive, max exclusive)."""
    keep = lambda b: (min_area is None or calculate_box_area(b) >= min_area) and \
                     (max_area is None or calculate_box_area(b) < max_area)
    return [p for p in predictions if keep(p["bbox"])], [g for g in ground_truths if keep(g["bbox"])]


def calculate_map(predictions: List[Dict], ground_truths: List[Dict], num_classes: int,
                  iou_thresholds: List[float] = None, class_names: List[str] = None, method: str = "coco",
                  use_parallel: bool = True, optimize_classes: bool = True, cache_ious: bool = True,
                  compute_per_scale: bool = True, strict_xyxy: bool = False) -> Dict[str, Any]:
    """metrics.py:529-815.  Same result dictionary: mAP, mAP50, mAP75, per_class, per_iou, num_predictions,
    num_ground_truths, APS/APM/APL(+50)."""
    if iou_thresholds is None:
        iou_thresholds = list(COCO_IOU_THRESHOLDS)
    if class_names is None:
        class_names = [f"class_{i}" for i in range(num_classes)]
    results = {"mAP": 0.0, "mAP50": 0.0, "mAP75": 0.0, "per_class": {}, "per_iou": {},
               "num_predictions": len(predictions), "num_ground_truths": len(ground_truths)}
    active = get_active_classes(predictions, ground_truths, num_classes) if optimize_classes else list(range(num_classes))
    # which IoU convention the reference would have used for this call (see the module docstring)
    parallel = use_parallel and len(active) > 1
    cached = cache_ious and not (parallel and len(predictions) > 10000)
    quirk = (not cached) and not strict_xyxy

    order, tp = match_all(predictions, ground_truths, iou_thresholds, quirk)
    pcls = np.array([predictions[i]["class"] for i in order], np.int64) if len(order) else np.zeros(0, np.int64)
    pscore = np.array([predictions[i]["score"] for i in order], np.float64) if len(order) else np.zeros(0)
    gcount = {}
    for g in ground_truths:
        gcount[g["class"]] = gcount.get(g["class"], 0) + 1
    iou_aps = {t: [] for t in iou_thresholds}
    for c in active:
        name = class_names[c] if c < len(class_names) else f"class_{c}"
        sel = np.nonzero(pcls == c)[0]
        ngt = gcount.get(c, 0)
        r = {}
        if len(sel):
            so = sel[np.argsort(pscore[sel])[::-1]]
        for ti, t in enumerate(iou_thresholds):
            if len(sel) == 0:
                ap = 0.0 if ngt > 0 else 1.0
            elif ngt == 0:
                ap = 0.0
            else:
                f = tp[ti][so]
                prec, rec = compute_precision_recall(f, ~f, ngt)
                ap = compute_average_precision(prec, rec, method)
            r[f"AP{t:.2f}"] = ap
            iou_aps[t].append(ap)
        r["AP"] = float(np.mean(list(r.values())))
        results["per_class"][name] = r
    for t in iou_thresholds:
        if len(iou_aps[t]) > 0:
            results["per_iou"][f"mAP{t:.2f}"] = float(np.mean(iou_aps[t]))
    if 0.5 in iou_thresholds:
        results["mAP50"] = results["per_iou"].get("mAP0.50", 0.0)
    if 0.75 in iou_thresholds:
        results["mAP75"] = results["per_iou"].get("mAP0.75", 0.0)
    if len(iou_thresholds) > 0:
        results["mAP"] = float(np.mean([results["per_iou"].get(f"mAP{t:.2f}", 0.0) for t in iou_thresholds]))
    if compute_per_scale:
        for key, lo, hi in (("APS", None, 1024.0), ("APM", 1024.0, 9216.0), ("APL", 9216.0, None)):
            sp, sg = filter_by_area(predictions, ground_truths, lo, hi)
            if len(sg) > 0:
                sub = calculate_map(sp, sg, num_classes, iou_thresholds, class_names, method, use_parallel=False,
                                    optimize_classes=optimize_classes, cache_ious=False, compute_per_scale=False,
                                    strict_xyxy=strict_xyxy)
                results[key], results[key + "50"] = sub["mAP"], sub.get("mAP50", 0.0)
            else:
                results[key], results[key + "50"] = 0.0, 0.0
    return results


def print_map_results(results: Dict[str, Any], top_k: int = 10):
    """metrics.py:817-864."""
    print("\n" + "=" * 80 + "\nmAP Evaluation Results\n" + "=" * 80)
    print(f"\nOverall Metrics:\n   mAP@0.5:0.95: {results['mAP']:.4f}\n   mAP@0.5:      {results['mAP50']:.4f}"
          f"\n   mAP@0.75:     {results['mAP75']:.4f}")
    if "APS" in results:
        print("\nPer-Scale Metrics (COCO style):")
        print(f"   APS (small,  area < 32^2):  {results['APS']:.4f} (AP50: {results.get('APS50', 0.0):.4f})")
        print(f"   APM (medium, 32^2 <= area < 96^2): {results['APM']:.4f} (AP50: {results.get('APM50', 0.0):.4f})")
        print(f"   APL (large,  area >= 96^2): {results['APL']:.4f} (AP50: {results.get('APL50', 0.0):.4f})")
    print("\nPer-IoU mAP:")
    for k, v in results["per_iou"].items():
        print(f"   {k}: {v:.4f}")
    print(f"\nPer-Class Results (Top {top_k} by AP@0.5):")
    ranked = sorted(((n, r.get("AP0.50", 0.0), r) for n, r in results["per_class"].items()), key=lambda x: x[1],
                    reverse=True)
    for i, (n, ap50, r) in enumerate(ranked[:top_k]):
        print(f"   {i + 1:2d}. {n:15s}: AP@0.5={ap50:.4f}, AP={r.get('AP', 0.0):.4f}")
    print(f"\nDataset Statistics:\n   Total Predictions: {results['num_predictions']:,}"
          f"\n   Total Ground Truths: {results['num_ground_truths']:,}\n\n" + "=" * 80)
