#!/usr/bin/env python3
"""eval.py - same flags as the reference CLI (reference eval.py:25-135): --config --weights --data --batch-size
--conf --max-images; exit code 0/1."""
import argparse
import sys
import traceback

from multigriddet_amd.config import ConfigLoader
from multigriddet_amd.evaluation import MultiGridEvaluator


def parse_args():
    p = argparse.ArgumentParser(description="Evaluate MultiGridDet model", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--config", type=str, default="configs/eval_config.yaml", help="Path to evaluation config file")
    p.add_argument("--weights", type=str, default=None, help="Model weights path (overrides config)")
    p.add_argument("--data", type=str, default=None, help="Annotation file path (overrides config)")
    p.add_argument("--batch-size", type=int, default=None, help="Batch size (overrides config)")
    p.add_argument("--conf", type=float, default=None, help="Confidence threshold (overrides config)")
    p.add_argument("--max-images", type=int, default=None, help="Maximum number of images to evaluate (for testing)")
    return p.parse_args()


def main():
    args = parse_args()
    try:
        config = ConfigLoader.load_config(args.config)
    except FileNotFoundError as e:
        print(f"[ERROR] {e}")
        return 1
    config.setdefault("data", {}); config.setdefault("evaluation", {})
    if args.weights:
        config["weights_path"] = args.weights
    if args.data:
        config["data"]["annotation"] = args.data
    if args.batch_size is not None:
        config["evaluation"]["batch_size"] = args.batch_size
    if args.conf is not None:
        config["evaluation"]["confidence_threshold"] = args.conf
    if args.max_images is not None:
        config["evaluation"]["max_images"] = args.max_images
    try:
        evaluator = MultiGridEvaluator(config)
        results = evaluator.evaluate()
        evaluator.print_results(results)
        return 0
    except KeyboardInterrupt:
        return 1
    except Exception as e:
        print(f"\n[ERROR] Evaluation error: {e}")
        traceback.print_exc()
        return 1


if __name__ == "__main__":
    sys.exit(main())
