#!/usr/bin/env python3
"""train.py - same flags as the reference CLI (reference train.py:26-138): --config --weights --backbone-weights
--resume --epochs --batch-size; exit code 0/1; exceptions are caught and printed."""
import argparse
import sys
import traceback

from multigriddet_amd.config import ConfigLoader
from multigriddet_amd.trainers import MultiGridTrainer


def parse_args():
    p = argparse.ArgumentParser(description="Train MultiGridDet model", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--config", type=str, default="configs/train_config.yaml", help="Path to training config file")
    p.add_argument("--weights", type=str, default=None, help="Path to pretrained full model weights (overrides config)")
    p.add_argument("--backbone-weights", type=str, default=None, help="Path to pretrained backbone weights")
    p.add_argument("--resume", action="store_true", help="Resume training from checkpoint")
    p.add_argument("--epochs", type=int, default=None, help="Number of epochs (overrides config)")
    p.add_argument("--batch-size", type=int, default=None, help="Batch size (overrides config)")
    return p.parse_args()


def main():
    args = parse_args()
    print("=" * 80 + "\nMultiGridDet Training\n" + "=" * 80 + f"\nConfig file: {args.config}")
    try:
        config = ConfigLoader.load_config(args.config)
    except FileNotFoundError as e:
        print(f"[ERROR] {e}")
        return 1
    config.setdefault("resume", {})
    if args.weights:
        config["resume"]["weights_path"] = args.weights
    if args.backbone_weights:
        config["resume"]["backbone_weights_path"] = args.backbone_weights
    if args.resume:
        config["resume"]["enabled"] = True
    if args.epochs:
        config["training"]["epochs"] = args.epochs
    if args.batch_size:
        config["training"]["batch_size"] = args.batch_size
    try:
        MultiGridTrainer(config).train()
        return 0
    except KeyboardInterrupt:
        print("\n[WARNING] Training interrupted by user")
        return 1
    except Exception as e:
        print(f"\n[ERROR] Training error: {e}")
        traceback.print_exc()
        return 1


if __name__ == "__main__":
    sys.exit(main())
