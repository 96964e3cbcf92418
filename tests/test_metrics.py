"""mAP metrics: the CPU oracle against the reference's own calculate_map output
(tests/golden/map.json), the device matcher against both, and the evaluator end to end."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_map import synth_detections  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "map.json")))
CASES = [("coco_s0", 0, "coco"), ("coco_s1", 1, "coco"), ("voc_s0", 0, "voc")]


def _check(res, gold, tol=1e-12):
    for k in ("mAP", "mAP50", "mAP75", "APS", "APS50", "APM", "APM50", "APL", "APL50"):
        assert abs(res[k] - gold[k]) <= tol, (k, res[k], gold[k])
    assert res["num_predictions"] == gold["num_predictions"] and res["num_ground_truths"] == gold["num_ground_truths"]
    assert set(res["per_class"]) == set(gold["per_class"])
    for c, r in gold["per_class"].items():
        for k, v in r.items():
            assert abs(res["per_class"][c][k] - v) <= tol, (c, k)
    for k, v in gold["per_iou"].items():
        assert abs(res["per_iou"][k] - v) <= tol, k


@pytest.mark.parametrize("tag,seed,method", CASES)
def test_oracle_map_matches_reference_output(tag, seed, method):
    from oracle import metrics as om
    preds, gts = synth_detections(seed)
    res = om.calculate_map(preds, gts, 6, class_names=[f"c{i}" for i in range(6)], method=method)
    _check(res, GOLD[tag])


def test_oracle_iou_matrix_matches_reference_output():
    from oracle import metrics as om
    g = GOLD["iou_matrix"]
    np.testing.assert_allclose(om.calculate_iou_matrix(g["boxes1"], g["boxes2"]), np.array(g["iou"]), rtol=0, atol=1e-15)


def test_oracle_edge_cases():
    from oracle import metrics as om
    gts = [{"bbox": [0, 0, 10, 10], "class": 0, "image_id": 0}]
    assert om.calculate_map([], gts, 2, per_scale=False)["mAP"] == 0.0          # GT but no predictions -> AP 0
    preds = [{"bbox": [0, 0, 10, 10], "class": 1, "score": 0.9, "image_id": 0}]
    r = om.calculate_map(preds, gts, 2, per_scale=False)                        # class 1: predictions, no GT -> 0
    assert r["per_class"]["class_1"]["AP"] == 0.0 and r["per_class"]["class_0"]["AP"] == 0.0
    perfect = [{"bbox": [0, 0, 10, 10], "class": 0, "score": 0.9, "image_id": 0}]
    assert abs(om.calculate_map(perfect, gts, 2, per_scale=False)["mAP"] - 1.0) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag,seed,method", CASES)
def test_device_map_matches_reference_output(tag, seed, method):
    from multigriddet_amd.evaluation import calculate_map
    preds, gts = synth_detections(seed)
    res = calculate_map(preds, gts, 6, class_names=[f"c{i}" for i in range(6)], method=method, use_parallel=False)
    _check(res, GOLD[tag])


@pytest.mark.gpu
def test_device_iou_matrix_and_modes():
    from multigriddet_amd.evaluation import calculate_iou_matrix, calculate_map
    from oracle import metrics as om
    g = GOLD["iou_matrix"]
    np.testing.assert_array_equal(calculate_iou_matrix(np.array(g["boxes1"]), np.array(g["boxes2"])), np.array(g["iou"]))
    assert calculate_iou_matrix(np.zeros((0, 4)), np.zeros((3, 4))).shape == (0, 3)
    with pytest.raises(ValueError):
        calculate_iou_matrix(np.zeros((2, 3)), np.zeros((2, 4)))
    # un-cached path = the (cx,cy,w,h) reading of BoxUtils.box_iou, cached = xyxy: both against the oracle, larger set
    preds, gts = synth_detections(5, n_images=120, n_classes=11)
    for cache in (True, False):
        res = calculate_map(preds, gts, 11, use_parallel=False, cache_ious=cache, compute_per_scale=False)
        ref = om.calculate_map(preds, gts, 11, quirk=not cache, per_scale=False)
        assert abs(res["mAP"] - ref["mAP"]) < 1e-12 and abs(res["mAP50"] - ref["mAP50"]) < 1e-12
    # empty inputs
    r = calculate_map([], [], 3)
    assert r["mAP"] == 0.0 and r["per_class"] == {}


@pytest.mark.gpu
def test_evaluator_end_to_end(tmp_path):
    """Random-init weights on synthetic images: exercises annotation parsing, batched forward + decode + NMS and
    the metric path; the numbers themselves are meaningless, the dictionary layout is the reference's."""
    import torch
    import yaml
    from PIL import Image
    from multigriddet_amd.models import build_multigriddet_darknet
    from multigriddet_amd.evaluation import MultiGridEvaluator
    root = os.path.dirname(HERE)
    rng = np.random.default_rng(0)
    lines = []
    for i in range(5):
        w, h = (320, 240) if i % 2 else (200, 300)
        p = tmp_path / f"im{i}.png"
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(p)
        lines.append(f"{p} 10,20,110,150,3 50,60,90,100,7")
    (tmp_path / "ann.txt").write_text("\n".join(lines) + "\n" + f"{tmp_path}/missing.png 1,2,3,4,0\n")
    model, _ = build_multigriddet_darknet(input_shape=(416, 416, 3), num_classes=80)
    wpath = str(tmp_path / "w.npz")
    model.save_weights(wpath)
    cfg = {"model_config": os.path.join(root, "configs/models/multigriddet_darknet.yaml"), "weights_path": wpath,
           "data": {"annotation": str(tmp_path / "ann.txt"), "classes_path": os.path.join(root, "configs/coco_classes.txt")},
           "evaluation": {"batch_size": 4, "input_shape": [416, 416, 3], "confidence_threshold": 0.01, "nms_threshold": 0.45,
                          "results_dir": str(tmp_path / "res"), "save_results": True}}
    ev = MultiGridEvaluator(cfg)
    res = ev.evaluate()
    for k in ("mAP", "mAP50", "mAP75", "APS", "APM", "APL", "per_class", "per_iou", "evaluation_info"):
        assert k in res
    assert res["evaluation_info"]["num_ground_truths"] == 10          # the unreadable image is skipped
    assert os.path.exists(tmp_path / "res" / "evaluation_results.json")
    ev.print_results(res)
