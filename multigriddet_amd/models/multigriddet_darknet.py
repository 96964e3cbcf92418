"""Model builders with the reference's signatures on top of the gfx950 graph executor.

 * build_multigriddet_darknet(input_shape=(416,416,3), num_anchors_per_head=[3,3,3], num_classes=80,
   weights_path=None, clear_session=False, **kw) -> (model, 185)
   (reference multigriddet/models/multigriddet_darknet.py:488-548): model(x) on NHWC fp32 in [0,1],
   .predict(x, verbose=0), .load_weights / .save_weights, .count_params(), .layers[i].trainable
 * build_multigriddet_darknet_train(anchors, num_classes=80, input_shape, weights_path,
   backbone_weights_path, freeze_level=1, optimizer, ..., **loss_kwargs) -> (training_model, 185)
   (reference :551-751): .fit(data, steps_per_epoch, validation_data, validation_steps, epochs,
   initial_epoch, callbacks), .compile(optimizer=...), .optimizer.learning_rate.assign(v)
Freeze levels as in the reference (:635-645): 1 = backbone (first 185 Keras layers = the 52 backbone
convs) frozen, 2 = everything but the three prediction convs frozen, 0 = all trainable.  Frozen
BatchNorm runs on its moving statistics, as Keras does for trainable=False.
Weights are stored as .npz with Keras-style names (conv2d[_N]/kernel:0, batch_normalization[_N]/...),
HWIO kernels; Keras .h5 needs h5py, which this image lacks.
"""
import os
from typing import List, Optional

import numpy as np
import torch

from ..engine import Network, BACKBONE_CONVS
from ..train_step import TrainStep


class _Layer:
    """Stand-in for a Keras layer handle: only .name and .trainable are used by the trainer."""

    def __init__(self, name, model, idx):
        self.name, self._m, self._idx = name, model, idx
        self._trainable = True

    @property
    def trainable(self):
        return self._trainable

    @trainable.setter
    def trainable(self, v):
        self._trainable = bool(v)
        self._m._sync_freeze()


def _keras_names(net):
    names, ci, bi = [], 0, 0
    for cv in net.layers:
        c = "conv2d" if ci == 0 else f"conv2d_{ci}"
        if cv.role == "pred":
            c = f"predict_conv_{sum(1 for x in net.layers[:cv.idx + 1] if x.role == 'pred')}"
        else:
            ci += 1
        b = None
        if cv.bn:
            b = "batch_normalization" if bi == 0 else f"batch_normalization_{bi}"
            bi += 1
        names.append((c, b))
    return names


class MultiGridDetModel:
    name = "multigriddet_darknet"

    def __init__(self, input_shape=(416, 416, 3), num_anchors_per_head=(3, 3, 3), num_classes=80, device="cuda:0",
                 seed=0, precision="bf16"):
        self.input_shape = tuple(input_shape)
        self.net = Network(num_classes, int(num_anchors_per_head[0]), device, seed=seed, precision=precision)
        self.net.training = False
        names = _keras_names(self.net)
        self.layers = [_Layer(n[0], self, i) for i, n in enumerate(names)]
        self.backbone_len = 185

    # -- freeze bookkeeping: map per-conv flags onto the executor's three modes
    def _sync_freeze(self):
        fl = [l.trainable for l in self.layers]
        self.net.freeze_backbone = not any(fl[:BACKBONE_CONVS])
        head_non_pred = [f for f, cv in zip(fl, self.net.layers) if cv.idx >= BACKBONE_CONVS and cv.role != "pred"]
        self.net.freeze_all_but_pred = self.net.freeze_backbone and not any(head_non_pred)

    def set_freeze_level(self, level):
        for l, cv in zip(self.layers, self.net.layers):
            l._trainable = (level == 0) or (level == 1 and cv.idx >= BACKBONE_CONVS) or (level == 2 and cv.role == "pred")
        self._sync_freeze()

    def __call__(self, x, training=False):
        prev = self.net.training
        self.net.training = bool(training)
        xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, np.float32))
        xt = xt.to(self.net.device, torch.float32).contiguous()
        try:
            if not training and getattr(self, "_use_graph", False):
                return self._forward_graph(xt)
            return self.net.forward(xt)
        finally:
            self.net.training = prev

    def enable_graph(self, on=True):
        """Opt-in: inference forwards are replayed from a captured hipGraph, one per input shape (the ~130 launches of
        a forward pass become one; matters for small batches, where the pass is launch-bound).  The returned head
        tensors are static buffers, overwritten by the next call - as in the eager path.  Weights may change in
        place (load_weights); call fold_bn / enable_graph again after anything that re-allocates."""
        self._use_graph = bool(on)
        self._graphs = {}
        return self

    def _forward_graph(self, xt):
        key = (tuple(xt.shape), bool(getattr(self.net, "folded", False)))
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"eager": 0, "graph": None}
        if st["graph"] is None:
            if st["eager"] < 2:                       # allocations (arena, workspaces) happen outside the capture
                st["eager"] += 1
                return self.net.forward(xt)
            st["x"] = xt.clone()
            # the capture runs on a stream of this model's own; its latency workspace (uncached memory: an allocation and a
            # device synchronisation) must exist before the capture starts.  Replays use that workspace: replay one model's
            # graphs from one stream at a time.
            cs = getattr(self, "_capture_stream", None)
            if cs is None:
                from ..streams import shared_stream
                cs = self._capture_stream = shared_stream("capture", self.net.device)
            with torch.cuda.stream(cs):
                self.net.latency_workspace()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cs):
                st["outs"] = self.net.forward(st["x"])
            st["graph"] = g
        if xt.data_ptr() != st["x"].data_ptr():
            st["x"].copy_(xt)
        st["graph"].replay()
        return st["outs"]

    def fold_bn(self, on=True):
        """Opt-in inference mode: every BatchNorm folded into its conv (one launch per conv block); see
        engine.Network.fold_bn.  Call after loading weights."""
        self.net.fold_bn(on)
        self._graphs = {}
        return self

    def predict(self, x, verbose=0, batch_size=None):
        return [o.cpu().numpy() for o in self(x, training=False)]

    def count_params(self):
        return self.net.count_params()

    def summary(self):
        print(f"multigriddet_darknet: {len(self.net.layers)} convs, {self.count_params():,} params "
              f"({self.net.n_params:,} trainable)")

    def save_weights(self, path, extra=None):
        out = dict(extra or {})
        for (cn, bn), p in zip(_keras_names(self.net), self.net.export_keras_style()):
            out[f"{cn}/kernel:0"] = p["kernel"]
            if bn:
                out[f"{bn}/gamma:0"], out[f"{bn}/beta:0"] = p["gamma"], p["beta"]
                out[f"{bn}/moving_mean:0"], out[f"{bn}/moving_variance:0"] = p["moving_mean"], p["moving_var"]
            else:
                out[f"{cn}/bias:0"] = p["bias"]
        if not path.endswith(".npz"):
            path = path + ".npz"
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        np.savez(path, **out)
        return path

    def load_weights(self, path, by_name=False, skip_mismatch=False, backbone_only=False):
        if path.endswith(".h5"):
            raise NotImplementedError("Keras .h5 weights need h5py, which is not available in this image; "
                                      "use the .npz written by save_weights")
        if not os.path.exists(path) and os.path.exists(path + ".npz"):
            path = path + ".npz"
        z = np.load(path, allow_pickle=False)
        plist = self.net.export_keras_style()
        for (cn, bn), p, cv in zip(_keras_names(self.net), plist, self.net.layers):
            if backbone_only and cv.idx >= BACKBONE_CONVS:
                continue
            key = f"{cn}/kernel:0"
            if key not in z:
                if by_name or skip_mismatch:
                    continue
                raise KeyError(key)
            if z[key].shape != p["kernel"].shape:
                if skip_mismatch:
                    continue
                raise ValueError(f"{key}: shape {z[key].shape} != {p['kernel'].shape}")
            p["kernel"] = z[key]
            if bn:
                p["gamma"], p["beta"] = z[f"{bn}/gamma:0"], z[f"{bn}/beta:0"]
                p["moving_mean"], p["moving_var"] = z[f"{bn}/moving_mean:0"], z[f"{bn}/moving_variance:0"]
            else:
                p["bias"] = z[f"{cn}/bias:0"]
        self.net.load_keras_style(plist)
        if getattr(self.net, "folded", False):
            if self.net.training:
                self.net.folded = False
            else:
                self.net.fold_bn(True)          # re-fold from the new weights
        return z


def build_multigriddet_darknet(input_shape=(416, 416, 3), num_anchors_per_head=(3, 3, 3), num_classes=80,
                               weights_path=None, clear_session=False, **kwargs):
    model = MultiGridDetModel(input_shape, num_anchors_per_head, num_classes, device=kwargs.get("device", "cuda:0"),
                              seed=kwargs.get("seed", 0), precision=kwargs.get("precision", "bf16"))
    if weights_path and (os.path.exists(weights_path) or os.path.exists(weights_path + ".npz")):
        model.load_weights(weights_path, backbone_only=True)
        print(f"Loaded backbone weights from {weights_path}")
    return model, 185


class _LR:
    def __init__(self, v):
        self.v = float(v)

    def assign(self, v):
        self.v = float(v)

    def numpy(self):
        return self.v

    def __float__(self):
        return self.v


class Optimizer:
    """What create_optimizer_from_config returns (reference config/model_builder.py:22-99)."""

    def __init__(self, kind="adam", learning_rate=1e-3, **kw):
        self.kind = kind
        self.learning_rate = _LR(learning_rate)
        self.kwargs = kw


class MultiGridDetTrainModel:
    """Training wrapper: inputs [image, y_true_0..2] (or image + boxes) -> scalar loss, like the Keras
    model with the Lambda loss layer (reference :648-726)."""

    def __init__(self, base, anchors, num_classes, input_shape, optimizer, loss_kwargs, world_size=1):
        self.base = base
        self.anchors = [np.asarray(a, np.float32) for a in anchors]
        self.num_classes = num_classes
        self.input_shape = tuple(input_shape[:2])
        self.loss_kwargs = loss_kwargs
        self.world_size = world_size
        self.layers = base.layers
        self.history = {"loss": [], "val_loss": [], "lr": []}
        self.stop_training = False
        self.compile(optimizer=optimizer)

    def compile(self, optimizer=None, loss=None, **kw):
        self.optimizer = optimizer if optimizer is not None else Optimizer("adam", 1e-3)
        cw = self.loss_kwargs.get("class_weights")
        lk = {k: v for k, v in self.loss_kwargs.items() if k != "class_weights"}
        self.ts = TrainStep(self.base.net, self.anchors, self.num_classes, self.input_shape, 0, loss_kwargs=lk,
                            lr=float(self.optimizer.learning_rate), optimizer=self.optimizer.kind,
                            opt_kwargs=self.optimizer.kwargs, world_size=self.world_size, class_weights=cw)
        self.sync_replicas()

    def sync_replicas(self):
        """Data-parallel replicas start from rank 0's state: master weights, BatchNorm moving statistics and the
        optimiser moments are broadcast and the packed bf16 images rebuilt from them (no-op for one process).  Without
        it the ranks would only agree while every process happens to draw the same initial weights."""
        if self.world_size <= 1:
            return
        from ..dp import broadcast_tensors
        net = self.base.net
        broadcast_tensors([net.params, net.moving, self.ts.m, self.ts.v], self.world_size)
        if not net.fp32:
            net.refresh_packed(0)

    def count_params(self):
        return self.base.count_params()

    def save_weights(self, path, include_optimizer=True):
        """Weights in the Keras-named .npz plus, by default, the optimiser state (`optimizer/m`, `optimizer/v`,
        `optimizer/step`, `optimizer/kind`, `optimizer/lr`) so that training resumes exactly; the reference's checkpoints
        hold weights only (trainers/trainer.py:356-368, save_weights_only)."""
        extra = None
        if include_optimizer:
            extra = {"optimizer/m": self.ts.m.cpu().numpy(), "optimizer/step": np.int64(self.ts.step_count),
                     "optimizer/kind": np.array(self.optimizer.kind), "optimizer/lr": np.float64(float(self.optimizer.learning_rate))}
            if self.ts.v is not None:
                extra["optimizer/v"] = self.ts.v.cpu().numpy()
        return self.base.save_weights(path, extra=extra)

    def load_weights(self, path, load_optimizer=True, **kw):
        z = self.base.load_weights(path, **kw)
        if load_optimizer and z is not None and "optimizer/m" in z and str(z["optimizer/kind"]) == self.optimizer.kind \
                and z["optimizer/m"].shape[0] == self.ts.m.numel():
            self.ts.m.copy_(torch.from_numpy(z["optimizer/m"]))
            if self.ts.v is not None and "optimizer/v" in z:
                self.ts.v.copy_(torch.from_numpy(z["optimizer/v"]))
            self.ts.step_count = int(z["optimizer/step"])
        self.sync_replicas()
        return z

    def train_on_batch(self, inputs):
        """inputs: (images, y0, y1, y2) device tensors.  Returns the loss (python float)."""
        if getattr(self.base.net, "folded", False):
            self.base.net.fold_bn(False)       # folded images are an inference-only view of the weights
        self.base.net.training = True
        self.ts.lr = float(self.optimizer.learning_rate)
        comp = self.ts.step(inputs[0], y_true=list(inputs[1:]))
        return float(comp[7])

    def test_on_batch(self, inputs):
        from ..losses import MultiGridLoss
        outs = self.base(inputs[0], training=False)
        if not hasattr(self, "_eval_loss"):
            lk = {k: v for k, v in self.loss_kwargs.items()}
            self._eval_loss = MultiGridLoss(self.anchors, self.num_classes, self.input_shape, **lk)
        return float(self._eval_loss(list(inputs[1:]), outs))

    def fit(self, x, steps_per_epoch=None, validation_data=None, validation_steps=None, epochs=1, initial_epoch=0,
            callbacks=None, verbose=1, **kw):
        callbacks = callbacks or []
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
        steps_per_epoch = steps_per_epoch or len(x)
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                if hasattr(cb, "on_epoch_begin"):
                    cb.on_epoch_begin(epoch, {})
            run = 0.0
            it = iter(x)
            for step in range(steps_per_epoch):
                try:
                    inputs, _ = next(it)
                except StopIteration:
                    it = iter(x)
                    inputs, _ = next(it)
                run += self.train_on_batch(inputs)
            logs = {"loss": run / max(steps_per_epoch, 1), "lr": float(self.optimizer.learning_rate)}
            if validation_data is not None:
                vs = validation_steps or len(validation_data)
                vit = iter(validation_data)
                logs["val_loss"] = float(np.mean([self.test_on_batch(next(vit)[0]) for _ in range(vs)]))
            if self.world_size > 1:
                # every rank's callbacks must see the same numbers (per-replica BN statistics make val_loss differ)
                from ..dp import all_reduce_mean_scalar
                for k in ("loss", "val_loss"):
                    if k in logs:
                        logs[k] = all_reduce_mean_scalar(logs[k], self.world_size)
            for k, v in logs.items():
                self.history.setdefault(k, []).append(v)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.6g}" for k, v in logs.items()))
            if hasattr(x, "on_epoch_end"):
                x.on_epoch_end()
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            if self.world_size > 1:
                from ..dp import broadcast_flag
                self.stop_training = broadcast_flag(self.stop_training, self.world_size)
            if self.stop_training:
                break
        return self


def build_multigriddet_darknet_train(anchors, num_classes=80, input_shape=(416, 416, 3), weights_path=None,
                                     backbone_weights_path=None, freeze_level=1, optimizer=None, label_smoothing=0,
                                     elim_grid_sense=False, loss_option=3, coord_scale=1.0, object_scale=1.0,
                                     no_object_scale=1.0, class_scale=1.0, anchor_scale=1.0, class_weights=None,
                                     clear_session=True, world_size=1, **loss_kwargs):
    num_anchors_per_head = [len(a) for a in anchors]
    base, blen = build_multigriddet_darknet(input_shape, num_anchors_per_head, num_classes,
                                            weights_path=backbone_weights_path)
    if weights_path:
        base.load_weights(weights_path, by_name=True, skip_mismatch=True)
        print(f"Loaded weights from {weights_path}")
    if freeze_level in (1, 2):
        base.set_freeze_level(freeze_level)
        print(f"Freeze level {freeze_level}: " + ("backbone frozen" if freeze_level == 1 else "all but prediction convs frozen"))
    elif freeze_level == 0:
        base.set_freeze_level(0)
    lk = dict(label_smoothing=label_smoothing, loss_option=loss_option, coord_scale=coord_scale,
              object_scale=object_scale, no_object_scale=no_object_scale, class_scale=class_scale,
              anchor_scale=anchor_scale, class_weights=class_weights)
    lk.update(loss_kwargs)
    lk.pop("consensus_kernel_size", None)
    lk.pop("elim_grid_sense", None)
    model = MultiGridDetTrainModel(base, anchors, num_classes, input_shape, optimizer, lk, world_size=world_size)
    return model, blen
