#!/usr/bin/env python3
"""Golden vectors for the evaluator's mAP.  Run in the build container only
(`python tests/golden/make_golden_map.py`): loads the reference's pure-numpy evaluation/metrics.py by file path
(same recipe and placeholders as make_golden.py - utils/boxes.py imports tensorflow at module level but the functions
called here never touch it) and runs `calculate_map` (sequential, cached-IoU path = the one evaluator.py takes for
<= 10 000 predictions) on seeded synthetic detections.  Writes tests/golden/map.json (inputs are regenerated from
the seed by `synth_detections`, which the tests import from here)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def synth_detections(seed=0, n_images=40, n_classes=6):
    """Ground truths with COCO-like sizes + predictions = jittered GTs (some dropped) + random false positives."""
    rng = np.random.default_rng(seed)
    preds, gts = [], []
    for img in range(n_images):
        for _ in range(int(rng.integers(1, 9))):
            w, h = np.exp(rng.uniform(np.log(8), np.log(300), 2))
            x1, y1 = rng.uniform(0, 608 - w), rng.uniform(0, 608 - h)
            c = int(rng.integers(0, n_classes))
            gts.append({"bbox": [float(x1), float(y1), float(x1 + w), float(y1 + h)], "class": c, "image_id": img})
            if rng.random() < 0.8:
                j = rng.normal(0, 0.08, 4) * np.array([w, h, w, h])
                b = np.floor(np.array([x1, y1, x1 + w, y1 + h]) + j + 0.5)
                cc = c if rng.random() < 0.9 else int(rng.integers(0, n_classes))
                preds.append({"bbox": [float(v) for v in b], "class": cc, "score": float(rng.uniform(0.1, 1.0)),
                              "image_id": img})
                if rng.random() < 0.2:     # duplicate detection of the same object
                    preds.append({"bbox": [float(v) for v in b + rng.integers(-3, 4, 4)], "class": cc,
                                  "score": float(rng.uniform(0.1, 0.6)), "image_id": img})
        for _ in range(int(rng.integers(0, 4))):
            w, h = np.exp(rng.uniform(np.log(8), np.log(300), 2))
            x1, y1 = rng.uniform(0, 608 - w), rng.uniform(0, 608 - h)
            preds.append({"bbox": [float(np.floor(x1)), float(np.floor(y1)), float(np.floor(x1 + w)), float(np.floor(y1 + h))],
                          "class": int(rng.integers(0, n_classes)), "score": float(rng.uniform(0.1, 0.7)), "image_id": img})
    return preds, gts


def main():
    import make_golden as mg
    mg._install_placeholders()
    mg._pkg("refmg", mg.REF)
    for sub in ("utils", "evaluation"):
        mg._pkg(f"refmg.{sub}", f"{mg.REF}/{sub}")
    mg._load("refmg.utils.boxes", f"{mg.REF}/utils/boxes.py")
    metrics = mg._load("refmg.evaluation.metrics", f"{mg.REF}/evaluation/metrics.py")
    out = {}
    for tag, seed, method in (("coco_s0", 0, "coco"), ("coco_s1", 1, "coco"), ("voc_s0", 0, "voc")):
        preds, gts = synth_detections(seed)
        res = metrics.calculate_map(preds, gts, num_classes=6, class_names=[f"c{i}" for i in range(6)], method=method,
                                    use_parallel=False, optimize_classes=True, cache_ious=True, compute_per_scale=True)
        out[tag] = json.loads(json.dumps(res, default=float))
    # the IoU matrix itself
    preds, gts = synth_detections(3, n_images=1)
    pb, gb = np.array([p["bbox"] for p in preds]), np.array([g["bbox"] for g in gts])
    out["iou_matrix"] = {"boxes1": pb.tolist(), "boxes2": gb.tolist(), "iou": metrics.calculate_iou_matrix(pb, gb).tolist()}
    with open(os.path.join(HERE, "map.json"), "w") as f:
        json.dump(out, f)
    print({k: (v.get("mAP"), v.get("mAP50"), v.get("APS")) for k, v in out.items() if "mAP" in v})


if __name__ == "__main__":
    main()
