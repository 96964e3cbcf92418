"""MultiGridTrainer with the reference's flow (reference multigriddet/trainers/trainer.py:106-594):
setup_data -> build_model -> setup_callbacks -> optional frozen stage -> unfreeze + recompile -> fit ->
save 'final_model.weights'.  The step loop itself is multigriddet_amd.train_step.TrainStep.
Under torch.distributed (WORLD_SIZE>1) every rank builds the same model, shards the annotation list by
rank and averages gradients over RCCL - the reference has no such mode."""
import math
import os
from typing import Any, Dict

import numpy as np

from ..config.config_loader import ConfigLoader
from ..config.model_builder import build_model_for_training, create_optimizer_from_config
from ..data.generators import MultiGridDataGenerator, load_annotation_lines
from ..utils.anchors import load_anchors, load_classes


class Callback:
    def set_model(self, model):
        self.model = model


class CosineAnnealingWithWarmup(Callback):
    """Per-epoch schedule (reference trainer.py:23-100): linear warm-up from warmup_lr_factor*lr over
    `warmup_epochs`, then cosine to `min_lr`."""

    def __init__(self, initial_lr, min_lr=1e-7, warmup_epochs=3, total_epochs=100, warmup_lr_factor=0.01, verbose=1):
        self.initial_lr, self.min_lr = initial_lr, min_lr
        self.warmup_epochs, self.total_epochs = warmup_epochs, total_epochs
        self.warmup_lr = initial_lr * warmup_lr_factor
        self.verbose = verbose

    def lr_at(self, epoch):
        n = epoch + 1
        if n <= self.warmup_epochs:
            return self.warmup_lr + (self.initial_lr - self.warmup_lr) * (n / self.warmup_epochs)
        progress = (n - self.warmup_epochs) / (self.total_epochs - self.warmup_epochs)
        return self.min_lr + (self.initial_lr - self.min_lr) * 0.5 * (1 + np.cos(np.pi * progress))

    def on_epoch_begin(self, epoch, logs=None):
        lr = float(self.lr_at(epoch))
        self.model.optimizer.learning_rate.assign(lr)
        if self.verbose:
            print(f"\nEpoch {epoch + 1}/{self.total_epochs} - Learning rate: {lr:.2e}")


class ModelCheckpoint(Callback):
    """save_weights_only checkpoints named ep{epoch:03d}-loss{loss:.3f}-val_loss{val_loss:.3f} (reference :356-368)."""

    def __init__(self, directory, monitor="val_loss", save_best_only=True, period=1):
        self.directory, self.monitor, self.best_only, self.period = directory, monitor, save_best_only, period
        self.best = math.inf

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        cur = logs.get(self.monitor, logs.get("loss", math.inf))
        if (epoch + 1) % self.period or (self.best_only and not cur < self.best):
            return
        self.best = min(self.best, cur)
        os.makedirs(self.directory, exist_ok=True)
        name = f"ep{epoch + 1:03d}-loss{logs.get('loss', 0):.3f}-val_loss{logs.get('val_loss', 0):.3f}.weights"
        self.model.save_weights(os.path.join(self.directory, name))


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", patience=50, min_delta=0.0):
        self.monitor, self.patience, self.min_delta = monitor, patience, min_delta
        self.best, self.wait = math.inf, 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if cur < self.best - self.min_delta:
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.model.stop_training = True


class MultiGridTrainer:
    def __init__(self, config: Dict[str, Any]):
        self.config = config
        self.model = None
        self.train_generator = self.val_generator = None
        self.callbacks = []
        self.model_config = ConfigLoader.load_config(config["model_config"])
        self.full_config = ConfigLoader.merge_configs(self.model_config, config)
        # one process per GPU: bind to cuda:LOCAL_RANK and create the RCCL process group before anything touches the GPU
        from ..dp import init_distributed
        self.rank, self.world, self.device = init_distributed()
        print("=" * 80 + "\nMultiGridDet Trainer Initialized (MI355X / gfx950)\n" + "=" * 80)

    def setup_data(self):
        dc, tc = self.config["data"], self.config["training"]
        preset = self.model_config["model"]["preset"]
        self.class_names = load_classes(dc["classes_path"])
        self.num_classes = len(self.class_names)
        self.anchors = load_anchors(preset["anchors_path"])
        train_lines = load_annotation_lines(dc["train_annotation"], shuffle=True)
        val_lines = load_annotation_lines(dc["val_annotation"], shuffle=False)
        if self.world > 1:
            from ..dp import shard_lines
            import torch.distributed as dist
            box = [train_lines]                       # rank 0's shuffle for everyone, then equal-length shards
            dist.broadcast_object_list(box, src=0)
            train_lines = shard_lines(box[0], self.rank, self.world)
        self.input_shape = tuple(preset["input_shape"][:2])
        ac = tc.get("augmentation", {})
        dl = self.config.get("data_loader", {})
        pb = dl.get("prefetch_buffer", "auto")        # batches; 'auto' / None = 6 (reference trainer.py:215-221)
        prefetch = int(pb) if isinstance(pb, (int, float)) and not isinstance(pb, bool) else 6
        base_seed = int(tc.get("seed", 0))
        # every rank draws its own permutation and augmentation (seed + rank) but the SAME multi-scale shapes (shape_seed)
        common = dict(batch_size=tc["batch_size"], input_shape=self.input_shape, anchors=self.anchors,
                      num_classes=self.num_classes, multi_anchor_assign=tc.get("multi_anchor_assign", False),
                      num_workers=dl.get("num_workers", 8), prefetch_factor=prefetch,
                      seed=base_seed + self.rank, shape_seed=base_seed)
        self.train_generator = MultiGridDataGenerator(
            annotation_lines=train_lines, augment=ac.get("enabled", True), enhance_augment=ac.get("enhance_type"),
            rescale_interval=ac.get("rescale_interval", -1), shuffle=True, mosaic_prob=ac.get("mosaic_prob", 0.3),
            mixup_prob=ac.get("mixup_prob", 0.1), max_boxes_per_image=ac.get("max_boxes_per_image", 100), **common)
        self.val_generator = MultiGridDataGenerator(annotation_lines=val_lines, augment=False, shuffle=False, **common)
        print(f"[INFO] classes {self.num_classes}, train {len(train_lines)}, val {len(val_lines)}")

    def build_model(self):
        rc = self.config.get("resume", {})
        self.model = build_model_for_training(self.full_config, anchors=self.anchors,
                                              weights_path=rc.get("weights_path"),
                                              backbone_weights_path=rc.get("backbone_weights_path"),
                                              world_size=self.world)
        self.model.base.summary()

    def setup_callbacks(self):
        tc, cc = self.config["training"], self.config.get("callbacks", {})
        out = self.config.get("output", {}).get("model_dir", "trained_models")
        ls = self.config.get("lr_schedule", {})
        lr = float(self.model.optimizer.learning_rate)
        self.callbacks = []
        if ls.get("type", "cosine") in ("cosine", "cosine_warmup", "cosine_annealing"):
            self.callbacks.append(CosineAnnealingWithWarmup(lr, ls.get("min_lr", 1e-7), ls.get("warmup_epochs", 3),
                                                            tc.get("epochs", 100), ls.get("warmup_lr_factor", 0.01)))
        if self.rank == 0:
            ck = cc.get("checkpoint", {})
            self.callbacks.append(ModelCheckpoint(out, ck.get("monitor", "val_loss"), ck.get("save_best_only", True),
                                                  ck.get("period", 1)))
        es = cc.get("early_stopping", {})
        if es.get("enabled", True):
            self.callbacks.append(EarlyStopping(es.get("monitor", "val_loss"), es.get("patience", 50),
                                                es.get("min_delta", 0.0)))

    def train(self):
        self.setup_data()
        self.build_model()
        self.setup_callbacks()
        tc = self.config["training"]
        epochs, initial_epoch = tc.get("epochs", 100), tc.get("initial_epoch", 0)
        transfer_epochs = tc.get("transfer_epochs", 0)
        spe, vs = max(1, len(self.train_generator)), max(1, len(self.val_generator))
        if transfer_epochs > 0 and initial_epoch < transfer_epochs:
            self.model.fit(self.train_generator, steps_per_epoch=spe, validation_data=self.val_generator,
                           validation_steps=vs, epochs=transfer_epochs, initial_epoch=initial_epoch,
                           callbacks=self.callbacks)
            self.model.base.set_freeze_level(1 if tc.get("next_freeze_level", 0) == 1 else 0)
            self.model.compile(optimizer=create_optimizer_from_config(self.config))
            initial_epoch = transfer_epochs
        history = self.model.fit(self.train_generator, steps_per_epoch=spe, validation_data=self.val_generator,
                                 validation_steps=vs, epochs=epochs, initial_epoch=initial_epoch,
                                 callbacks=self.callbacks)
        if self.rank == 0:
            out = self.config.get("output", {}).get("model_dir", "trained_models")
            os.makedirs(out, exist_ok=True)
            self.model.save_weights(os.path.join(out, "final_model.weights"))
        return history
