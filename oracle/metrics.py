"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's mAP computation (multigriddet/evaluation/metrics.py).
Pinned by tests/golden/map.json, generated from the reference's own calculate_map (tests/golden/make_golden_map.py).

Follows: calculate_iou_matrix :28-71, match_predictions_to_gt :73-145 (un-cached: BoxUtils.box_iou reads the xyxy
numbers as cx,cy,w,h - utils/boxes.py:16-57), match_predictions_to_gt_cached :148-219, compute_precision_recall
:221-248, compute_average_precision :251-304, calculate_ap_for_class :307-346, filter_by_area :426-453,
calculate_map :529-815 (sequential branch; per-scale sub-runs are always un-cached)."""
import numpy as np


def iou_xyxy(a, b):
    x1, y1 = max(a[0], b[0]), max(a[1], b[1])
    x2, y2 = min(a[2], b[2]), min(a[3], b[3])
    inter = max(0.0, x2 - x1) * max(0.0, y2 - y1)
    union = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / union if union > 0 else 0.0


def iou_center_quirk(a, b):
    ax0, ay0, ax1, ay1 = a[0] - a[2] / 2, a[1] - a[3] / 2, a[0] + a[2] / 2, a[1] + a[3] / 2
    bx0, by0, bx1, by1 = b[0] - b[2] / 2, b[1] - b[3] / 2, b[0] + b[2] / 2, b[1] + b[3] / 2
    ixmin, iymin, ixmax, iymax = max(ax0, bx0), max(ay0, by0), min(ax1, bx1), min(ay1, by1)
    if ixmax <= ixmin or iymax <= iymin:
        return 0.0
    inter = (ixmax - ixmin) * (iymax - iymin)
    union = a[2] * a[3] + b[2] * b[3] - inter
    return inter / union if union > 0 else 0.0


def calculate_iou_matrix(b1, b2):
    b1, b2 = np.asarray(b1, np.float64).reshape(-1, 4), np.asarray(b2, np.float64).reshape(-1, 4)
    return np.array([[iou_xyxy(p, q) for q in b2] for p in b1], np.float64).reshape(len(b1), len(b2))


def match(preds, gts, thr, quirk):
    """preds / gts of ONE class.  Returns tp flags in descending-score order and the sorted scores."""
    scores = np.array([p["score"] for p in preds])
    order = np.argsort(scores)[::-1]
    tp = np.zeros(len(preds), bool)
    used = set()
    f = iou_center_quirk if quirk else iou_xyxy
    for i, pi in enumerate(order):
        p = preds[pi]
        best, bj = 0.0, None
        for j, g in enumerate(gts):
            if g["image_id"] != p["image_id"] or j in used:
                continue
            v = f(p["bbox"], g["bbox"])
            if v > best:
                best, bj = v, j
        if bj is not None and best >= thr:
            tp[i] = True
            used.add(bj)
    return tp, scores[order]


def average_precision(tp, num_gt, method="coco"):
    if len(tp) == 0:
        prec, rec = np.array([0.0]), np.array([0.0])
    else:
        ctp, cfp = np.cumsum(tp), np.cumsum(~tp)
        prec, rec = ctp / (ctp + cfp + 1e-8), ctp / (num_gt + 1e-8)
    if method == "voc":
        return float(np.mean([prec[rec >= r].max() if (rec >= r).any() else 0.0 for r in np.arange(0, 1.1, 0.1)]))
    o = np.argsort(rec)
    rs, ps = rec[o], prec[o]
    pi = np.array([ps[i:].max() for i in range(len(ps))])
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return float(trapz(pi, rs)) if len(rs) > 1 else float(pi[0] * rs[0])


def ap_for_class(preds, gts, c, thr, method, quirk):
    cp = [p for p in preds if p["class"] == c]
    cg = [g for g in gts if g["class"] == c]
    if len(cp) == 0:
        return 0.0 if len(cg) > 0 else 1.0
    if len(cg) == 0:
        return 0.0
    tp, _ = match(cp, cg, thr, quirk)
    return average_precision(tp, len(cg), method)


def _area(b):
    return (b[2] - b[0]) * (b[3] - b[1])


def calculate_map(preds, gts, num_classes, iou_thresholds=None, class_names=None, method="coco", quirk=False,
                  per_scale=True):
    ths = iou_thresholds or [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]
    names = class_names or [f"class_{i}" for i in range(num_classes)]
    active = sorted({p["class"] for p in preds} | {g["class"] for g in gts})
    res = {"per_class": {}, "per_iou": {}, "num_predictions": len(preds), "num_ground_truths": len(gts)}
    per_iou = {t: [] for t in ths}
    for c in active:
        r = {}
        for t in ths:
            ap = ap_for_class(preds, gts, c, t, method, quirk)
            r[f"AP{t:.2f}"] = ap
            per_iou[t].append(ap)
        r["AP"] = float(np.mean(list(r.values())))
        res["per_class"][names[c] if c < len(names) else f"class_{c}"] = r
    for t in ths:
        if per_iou[t]:
            res["per_iou"][f"mAP{t:.2f}"] = float(np.mean(per_iou[t]))
    res["mAP50"] = res["per_iou"].get("mAP0.50", 0.0) if 0.5 in ths else 0.0
    res["mAP75"] = res["per_iou"].get("mAP0.75", 0.0) if 0.75 in ths else 0.0
    res["mAP"] = float(np.mean([res["per_iou"].get(f"mAP{t:.2f}", 0.0) for t in ths])) if ths else 0.0
    if per_scale:
        for key, lo, hi in (("APS", None, 1024.0), ("APM", 1024.0, 9216.0), ("APL", 9216.0, None)):
            sel = lambda L: [x for x in L if (lo is None or _area(x["bbox"]) >= lo) and (hi is None or _area(x["bbox"]) < hi)]
            sp, sg = sel(preds), sel(gts)
            if sg:
                sub = calculate_map(sp, sg, num_classes, ths, names, method, quirk=True, per_scale=False)
                res[key], res[key + "50"] = sub["mAP"], sub.get("mAP50", 0.0)
            else:
                res[key], res[key + "50"] = 0.0, 0.0
    return res
