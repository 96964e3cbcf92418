"""MultiGridEvaluator with the reference's interface (reference multigriddet/evaluation/evaluator.py:37-654):
`MultiGridEvaluator(config).evaluate() -> dict`, `.print_results(results)`.  Same config keys (`model_config`,
`weights_path`, `data.annotation`, `data.classes_path`, `evaluation.{batch_size,input_shape,confidence_threshold,
nms_threshold,nms_method,use_wbf,max_images,iou_thresholds,interpolation_method,optimize_classes,cache_ious,
use_parallel,save_results,results_dir}`), same annotation line format (`path x1,y1,x2,y2,cls ...`, :101-118),
max_boxes = 500 per image (:276, :571), same prediction / ground-truth dictionaries fed to calculate_map.

What differs by construction: images are letterboxed on the host (PIL, utils/preprocessing.py = the reference's
fallback path `_process_single_image`, :558-593), the whole batch then stays on the GPU through forward, decode and
NMS (`MultiGridDecoder.postprocess_batch`) - the reference copies every image's heads to the host and runs NMS in a
thread pool (:238-300) - and the matching for every class and IoU threshold is one kernel launch (metrics.py here)."""
import json
import os
import time
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

from ..config.config_loader import ConfigLoader
from ..config.model_builder import build_model_for_inference
from ..postprocess import MultiGridDecoder
from ..utils.anchors import load_anchors, load_classes
from ..utils.preprocessing import preprocess_image
from .metrics import calculate_map, print_map_results


class MultiGridEvaluator:
    def __init__(self, config: Dict[str, Any]):
        self.config = config
        self.model = None
        self.class_names = None
        self.anchors = None
        self.model_config = ConfigLoader.load_config(config["model_config"])
        self.full_config = ConfigLoader.merge_configs(self.model_config, config)
        print("=" * 80 + "\nMultiGridDet Evaluator Initialized (MI355X / gfx950)\n" + "=" * 80)
        self._load_model()

    def _load_model(self):
        weights_path = self.config.get("weights_path")
        if not weights_path:
            raise ValueError("weights_path not specified in config")
        classes_path = self.config.get("data", {}).get("classes_path") or \
            self.full_config["model"]["preset"].get("classes_path")
        if not classes_path:
            raise ValueError("classes_path not found in config")
        self.class_names = load_classes(classes_path)
        self.anchors = load_anchors(self.model_config["model"]["preset"]["anchors_path"])
        self.model = build_model_for_inference(self.full_config, weights_path)
        # BatchNorm folded into the convs (one launch per DarknetConv2D_BN_Leaky): the default for inference - 4 534 against
        # 3 661 images/s at batch 16; `fold_bn: false` (not a reference key) keeps the separate BatchNorm launches
        if self.config.get("fold_bn", True):
            self.model.fold_bn(True)

    def _load_annotations(self, annotation_file: str) -> List[Dict]:
        annotations = []
        with open(annotation_file, "r") as f:
            for line in f:
                parts = line.strip().split()
                if len(parts) < 2:
                    continue
                boxes = []
                for box_str in parts[1:]:
                    bp = box_str.split(",")
                    if len(bp) == 5:
                        x1, y1, x2, y2, cls = map(float, bp)
                        boxes.append({"bbox": [x1, y1, x2, y2], "class": int(cls)})
                annotations.append({"image_path": parts[0], "boxes": boxes})
        print(f"Loaded {len(annotations)} annotations from {annotation_file}")
        return annotations

    def collect_detections(self, annotations: List[Dict], input_shape: Tuple[int, int], batch_size: int,
                           confidence: float, nms_threshold: float, nms_method: str, use_wbf: bool):
        """Forward + decode + NMS over the annotation list; returns (predictions, ground_truths) dictionaries."""
        from PIL import Image
        decoder = MultiGridDecoder(self.anchors, len(self.class_names), input_shape, rescore_confidence=True)
        predictions, ground_truths = [], []
        for b0 in range(0, len(annotations), batch_size):
            chunk = annotations[b0:b0 + batch_size]
            imgs, shapes, ids = [], [], []
            for k, a in enumerate(chunk):
                try:
                    im = Image.open(a["image_path"]).convert("RGB")
                except Exception as e:            # the reference warns and skips unreadable images (:591-592)
                    print(f"[WARNING] Error processing {a['image_path']}: {e}")
                    continue
                imgs.append(preprocess_image(im, input_shape))
                shapes.append(tuple(reversed(im.size)))
                ids.append(b0 + k)
            if not imgs:
                continue
            outs = self.model(torch.from_numpy(np.concatenate(imgs, 0)).cuda(), training=False)
            ob, osc, ocl, ocn = decoder.postprocess_batch(outs, shapes, max_boxes=500, confidence=confidence,
                                                          nms_threshold=nms_threshold, nms_method=nms_method,
                                                          return_xyxy=True, use_wbf=use_wbf)
            ob, osc, ocl, ocn = ob.cpu().numpy(), osc.cpu().numpy(), ocl.cpu().numpy(), ocn.cpu().numpy()
            for j, image_id in enumerate(ids):
                for q in range(int(ocn[j])):
                    predictions.append({"image_id": image_id, "class": int(ocl[j, q]), "bbox": ob[j, q].tolist(),
                                        "score": float(osc[j, q])})
                for gt in annotations[image_id]["boxes"]:
                    ground_truths.append({"image_id": image_id, "class": gt["class"], "bbox": gt["bbox"]})
        return predictions, ground_truths

    def evaluate(self) -> Dict[str, Any]:
        print("=" * 80 + "\nStarting Evaluation\n" + "=" * 80)
        t0 = time.time()
        annotations = self._load_annotations(self.config["data"]["annotation"])
        ec = self.config["evaluation"]
        input_shape = tuple(ec["input_shape"][:2])
        max_images = ec.get("max_images", None)
        if max_images:
            annotations = annotations[:max_images]
        preds, gts = self.collect_detections(annotations, input_shape, int(ec.get("batch_size", 1)),
                                             ec["confidence_threshold"], ec["nms_threshold"],
                                             ec.get("nms_method", "diou"), ec.get("use_wbf", False))
        t_inf = time.time() - t0
        print(f"Inference: {len(annotations)} images in {t_inf:.2f} s, {len(preds)} predictions, {len(gts)} ground truths")
        results = self._calculate_metrics(preds, gts, ec)
        results["timing"] = {"inference_s": t_inf, "total_s": time.time() - t0,
                             "images_per_s": len(annotations) / max(t_inf, 1e-9)}
        if ec.get("save_results", True):
            self._save_results(results, ec)
        return results

    def _calculate_metrics(self, predictions: List, ground_truths: List, eval_config: Dict) -> Dict:
        iou_thresholds = eval_config.get("iou_thresholds", None) or \
            [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]
        method = eval_config.get("interpolation_method", "coco")
        results = calculate_map(predictions=predictions, ground_truths=ground_truths, num_classes=len(self.class_names),
                                iou_thresholds=iou_thresholds, class_names=self.class_names, method=method,
                                optimize_classes=eval_config.get("optimize_classes", True),
                                cache_ious=eval_config.get("cache_ious", True),
                                use_parallel=eval_config.get("use_parallel", True))
        results["evaluation_info"] = {
            "num_predictions": len(predictions), "num_ground_truths": len(ground_truths),
            "classes_predicted": len({p["class"] for p in predictions}),
            "classes_in_gt": len({g["class"] for g in ground_truths}),
            "iou_thresholds": iou_thresholds, "interpolation_method": method}
        return results

    def _save_results(self, results: Dict, eval_config: Dict):
        results_dir = eval_config.get("results_dir", "results/evaluation")
        os.makedirs(results_dir, exist_ok=True)
        path = os.path.join(results_dir, "evaluation_results.json")
        with open(path, "w") as f:
            json.dump(results, f, indent=2, default=float)
        print(f"\nResults saved to: {path}")

    def print_results(self, results: Dict):
        print_map_results(results, top_k=10)
        if "evaluation_info" in results:
            info = results["evaluation_info"]
            print(f"\n[INFO] Evaluation Details:\n   Total Predictions: {info['num_predictions']:,}"
                  f"\n   Total Ground Truths: {info['num_ground_truths']:,}\n   Classes Predicted: {info['classes_predicted']}"
                  f"\n   Classes in GT: {info['classes_in_gt']}\n   IoU Thresholds: {len(info['iou_thresholds'])} points"
                  f"\n   Interpolation: {info['interpolation_method'].upper()}")
        print("\n" + "=" * 80)
