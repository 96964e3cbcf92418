#!/usr/bin/env python3
"""infer.py - same flags as the reference CLI (reference infer.py:33-170): --config --input --output --weights
--type --conf --nms --no-save --no-show; exit code 0/1."""
import argparse
import sys
import traceback

from multigriddet_amd.config import ConfigLoader
from multigriddet_amd.inference import MultiGridInference


def parse_args():
    p = argparse.ArgumentParser(description="Run MultiGridDet inference", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--config", type=str, default="configs/infer_config.yaml")
    p.add_argument("--input", type=str, default=None)
    p.add_argument("--output", type=str, default=None)
    p.add_argument("--weights", type=str, default=None)
    p.add_argument("--type", type=str, default=None, choices=["image", "video", "camera", "directory"])
    p.add_argument("--conf", type=float, default=None)
    p.add_argument("--nms", type=float, default=None)
    p.add_argument("--no-save", action="store_true")
    p.add_argument("--no-show", action="store_true")
    return p.parse_args()


def main():
    args = parse_args()
    try:
        config = ConfigLoader.load_config(args.config)
    except FileNotFoundError as e:
        print(f"[ERROR] {e}")
        return 1
    config.setdefault("input", {}); config.setdefault("output", {}); config.setdefault("detection", {})
    if args.input:
        config["input"]["source"] = args.input
    if args.output:
        config["output"]["output_dir"] = args.output
    if args.weights:
        config["weights_path"] = args.weights
    if args.type:
        config["input"]["type"] = args.type
    if args.conf is not None:
        config["detection"]["confidence_threshold"] = args.conf
    if args.nms is not None:
        config["detection"]["nms_threshold"] = args.nms
    if args.no_save:
        config["output"]["save_result"] = False
    if args.no_show:
        config["output"]["show_result"] = False
    try:
        MultiGridInference(config).run()
        return 0
    except KeyboardInterrupt:
        return 1
    except Exception as e:
        print(f"\n[ERROR] Inference error: {e}")
        traceback.print_exc()
        return 1


if __name__ == "__main__":
    sys.exit(main())
