#!/usr/bin/env python3
"""Generate golden vectors by running the reference's *numpy* code in the build container.

This script is fixture tooling only: it is run once, here, where `/root/reference` is mounted,
and its outputs (`tests/golden/*.npz`) are committed.  Nothing under `tests/`, `bench.py` or
`__graft_entry__` imports it, and the GPU box never sees the reference.

Loading recipe: the reference's `multigriddet/__init__.py` eagerly imports
TensorFlow-dependent subpackages, so individual files are loaded by path under a private
package name.  Files whose top-level imports name TensorFlow / cv2 / imgaug (none of which the
numpy functions we call ever touch) get inert `types.ModuleType` entries in `sys.modules`
so that the `import` statements succeed.  Only pure numpy/scipy functions are called:

  postprocess/nms.py            StandardNMS, DIoUNMS, SoftNMS, ClusterNMS      (P4)
  postprocess/wbf.py            WeightedBoxesFusion                            (P6)
  postprocess/multigrid_decode  MultiGridDecoder.{decode_predictions,correct_boxes,postprocess}  (P1-P5)
  data/generators.py            preprocess_true_boxes, best_fit_and_layer, iol_common_center     (T2)
  utils/anchors.py              load_anchors

Everything TensorFlow-backed (MultiGridLoss, tf_preprocess_true_boxes, the Keras model) cannot
be executed here; for those the oracle is the CPU restatement in `oracle/` pinned by the two
known-answer cases in the reference's own tests (tests/test_target_consistency.py:29-51,
tests/test_9cell_alignment.py:21-50), reproduced below through the numpy builder.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/multigriddet"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Inert(types.ModuleType):
    """Placeholder module: attribute access yields another inert object; never a number."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = _InertObj(f"{self.__name__}.{name}")
        setattr(self, name, obj)
        return obj


class _InertObj:
    def __init__(self, name):
        self._n = name

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _InertObj(self._n + "." + name)

    def __call__(self, *a, **k):
        # used as a decorator (tf.function) -> identity; otherwise inert
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _InertObj(self._n + "()")

    def __mro_entries__(self, bases):
        return (object,)

    def __bool__(self):          # `if gpus:` at import time (data/generators.py:52-53) -> skipped
        return False

    def __iter__(self):
        return iter(())


def _install_placeholders():
    names = [
        "tensorflow", "tensorflow.keras", "tensorflow.keras.backend", "tensorflow.keras.utils",
        "tensorflow.keras.layers", "tensorflow.keras.models", "tensorflow.keras.optimizers",
        "tensorflow.keras.regularizers", "tensorflow.keras.callbacks", "cv2", "h5py", "imgaug",
        "imgaug.augmenters",
    ]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = _Inert(n)
    seq = type("Sequence", (), {})
    sys.modules["tensorflow.keras.utils"].Sequence = seq
    sys.modules["tensorflow"].function = lambda f=None, **k: (f if f is not None else (lambda g: g))
    sys.modules["tensorflow"].keras = sys.modules["tensorflow.keras"]
    sys.modules["tensorflow.keras"].utils = sys.modules["tensorflow.keras.utils"]
    sys.modules["tensorflow.keras"].backend = sys.modules["tensorflow.keras.backend"]


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    _install_placeholders()
    _pkg("refmg", REF)
    for sub in ("postprocess", "data", "utils"):
        _pkg(f"refmg.{sub}", f"{REF}/{sub}")
    ref = types.SimpleNamespace()
    ref.nms = _load("refmg.postprocess.nms", f"{REF}/postprocess/nms.py")
    ref.wbf = _load("refmg.postprocess.wbf", f"{REF}/postprocess/wbf.py")
    ref.decode = _load("refmg.postprocess.multigrid_decode", f"{REF}/postprocess/multigrid_decode.py")
    ref.anchors = _load("refmg.utils.anchors", f"{REF}/utils/anchors.py")
    ref.dutils = _load("refmg.data.utils", f"{REF}/data/utils.py")
    ref.aug = _load("refmg.data.augmentation", f"{REF}/data/augmentation.py")
    ref.gen = _load("refmg.data.generators", f"{REF}/data/generators.py")
    return ref


# ----------------------------------------------------------------------------- inputs
def coco_anchors():
    """configs/yolov3_coco_anchor.txt, largest-anchor scale first (stride 32, 16, 8)."""
    return [
        np.array([[112, 74], [149, 190], [370, 328]], dtype=np.float32),
        np.array([[28, 17], [56, 112], [57, 35]], dtype=np.float32),
        np.array([[9, 10], [13, 28], [28, 55]], dtype=np.float32),
    ]


def synth_boxes(rng, batch, size, max_boxes=100, nmax=20, num_classes=80, integer=False):
    """SURVEY §8(d) config-2 box generator: n~U{1..nmax}, w,h=exp(U(ln 8, ln 400)) clipped to the
    image, centre uniform s.t. the box lies inside; zero-padded rows."""
    out = np.zeros((batch, max_boxes, 5), dtype=np.float32)
    for b in range(batch):
        n = int(rng.integers(1, nmax + 1))
        for t in range(n):
            w = min(float(np.exp(rng.uniform(np.log(8), np.log(400)))), size - 2)
            h = min(float(np.exp(rng.uniform(np.log(8), np.log(400)))), size - 2)
            cx = rng.uniform(w / 2, size - w / 2)
            cy = rng.uniform(h / 2, size - h / 2)
            x1, y1, x2, y2 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
            if integer:
                x1, y1, x2, y2 = np.floor(x1), np.floor(y1), np.ceil(x2), np.ceil(y2)
            out[b, t] = [x1, y1, x2, y2, rng.integers(0, num_classes)]
    return out


def main():
    ref = load_reference()
    anchors = coco_anchors()

    # ---- anchors file parse (utils/anchors.py:282-312)
    la = ref.anchors.load_anchors("/root/reference/configs/yolov3_coco_anchor.txt")
    np.savez_compressed(f"{OUT}/anchors.npz", a0=la[0], a1=la[1], a2=la[2])

    # ---- NMS variants (postprocess/nms.py) on clustered boxes, top-left xywh
    rng = np.random.default_rng(30)
    centres = rng.uniform(50, 550, size=(12, 2))
    boxes = []
    for c in centres:
        k = int(rng.integers(5, 40))
        wh = rng.uniform(20, 200, size=2)
        for _ in range(k):
            cc = c + rng.normal(0, 8, 2)
            ww = wh * np.exp(rng.normal(0, 0.15, 2))
            boxes.append([cc[0] - ww[0] / 2, cc[1] - ww[1] / 2, ww[0], ww[1]])
    boxes = np.array(boxes, dtype=np.float32)
    scores = rng.uniform(0.1, 1.0, size=len(boxes)).astype(np.float32)
    classes = rng.integers(0, 80, size=len(boxes)).astype(np.int64)
    gold = {"boxes": boxes, "scores": scores, "classes": classes}
    for name, cls in (("standard", ref.nms.StandardNMS), ("diou", ref.nms.DIoUNMS),
                      ("cluster", ref.nms.ClusterNMS), ("soft", ref.nms.SoftNMS)):
        for thr in (0.3, 0.45, 0.5):
            b, c, s = cls().apply_nms(boxes.copy(), classes.copy(), scores.copy(), thr, 0.1)
            b = np.concatenate(b) if isinstance(b, list) and len(b) else np.asarray(b)
            c = np.concatenate(c) if isinstance(c, list) and len(c) else np.asarray(c)
            s = np.concatenate(s) if isinstance(s, list) and len(s) else np.asarray(s)
            tag = f"{name}_{int(thr * 100)}"
            gold[f"{tag}_boxes"], gold[f"{tag}_classes"], gold[f"{tag}_scores"] = b, c, s
    np.savez_compressed(f"{OUT}/nms.npz", **gold)

    # ---- WBF (postprocess/wbf.py)
    wb, wc, ws = ref.wbf.WeightedBoxesFusion(iou_thr=0.5).fuse_boxes(
        [boxes.copy()], [classes.copy() % 3], [scores.copy()], (608, 608))
    np.savez_compressed(f"{OUT}/wbf.npz", boxes=boxes, scores=scores, classes=classes % 3,
                        out_boxes=np.asarray(wb), out_classes=np.asarray(wc), out_scores=np.asarray(ws))

    # ---- decode -> correct_boxes -> NMS -> xyxy (postprocess/multigrid_decode.py)
    for size, seed, img_shape in ((416, 20, (375, 500)), (608, 21, (480, 640)), (608, 22, (608, 608))):
        rng = np.random.default_rng(seed)
        grids = [size // 32, size // 16, size // 8]
        heads = [(2.0 * rng.standard_normal((1, g, g, 88))).astype(np.float32) for g in grids]
        dec = ref.decode.MultiGridDecoder(anchors, 80, input_shape=(size, size))
        raw = dec.decode_predictions([h.copy() for h in heads])
        cor = dec.correct_boxes(raw.copy(), img_shape, (size, size))
        # heads are regenerated from the seed by the tests (numpy Generator streams are stable);
        # a float64 checksum pins the regeneration.  Full decoded rows only for the 416 case, a
        # every-16th-row sample for 608, to keep the fixtures small.
        step = 1 if size == 416 else 16
        res = {"seed": np.array(seed), "size": np.array(size), "row_step": np.array(step),
               "head_sums": np.array([h.astype(np.float64).sum() for h in heads]),
               "image_shape": np.array(img_shape),
               "decoded": raw.astype(np.float32)[:, ::step], "corrected": cor.astype(np.float32)[:, ::step]}
        for method, thr, conf in (("diou", 0.45, 0.1), ("diou", 0.5, 0.3), ("cluster", 0.45, 0.1),
                                  ("soft", 0.45, 0.3)):
            b, c, s = dec.postprocess([h.copy() for h in heads], img_shape, (size, size), max_boxes=100,
                                      confidence=conf, nms_threshold=thr, nms_method=method)
            tag = f"{method}_{int(thr * 100)}_{int(conf * 100)}"
            res[f"{tag}_boxes"], res[f"{tag}_classes"], res[f"{tag}_scores"] = \
                np.asarray(b), np.asarray(c), np.asarray(s)
        np.savez_compressed(f"{OUT}/decode_{size}_{seed}.npz", **res)

    # ---- numpy target builder T2 (data/generators.py:3393-3473)
    for size, seed, batch, integer in ((608, 1, 4, False), (416, 2, 3, True), (608, 3, 2, False)):
        rng = np.random.default_rng(seed)
        nmax = 60 if seed == 3 else 20      # seed 3: crowded -> collisions and the >=3-cell rule
        tb = synth_boxes(rng, batch, size, nmax=nmax, integer=integer)
        grids = [(size // 32,) * 2, (size // 16,) * 2, (size // 8,) * 2]
        yt = ref.gen.preprocess_true_boxes(tb.copy(), (size, size), anchors, 80, False, grid_shapes=grids)
        np.savez_compressed(f"{OUT}/targets_np_{size}_{seed}.npz", boxes=tb, y0=yt[0], y1=yt[1], y2=yt[2])

    # ---- the reference's own known-answer cases (tests/test_target_consistency.py:29-51,
    #      tests/test_9cell_alignment.py:21-50): anchors as written there (small->large)
    kat_anchors = [
        np.array([[10, 13], [16, 30], [33, 23]], dtype=np.float32),
        np.array([[30, 61], [62, 45], [59, 119]], dtype=np.float32),
        np.array([[116, 90], [156, 198], [373, 326]], dtype=np.float32),
    ]
    grids = [(19, 19), (38, 38), (76, 76)]
    for tag, (cx, cy) in (("consistency", (304.0, 304.0)), ("9cell", (311.999, 311.999))):
        bw, bh = 100.0, 80.0
        tb = np.array([[[cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2, 0]]], dtype=np.float32)
        yt = ref.gen.preprocess_true_boxes(tb.copy(), (608, 608), kat_anchors, 1, False, grid_shapes=grids)
        np.savez_compressed(f"{OUT}/targets_kat_{tag}.npz", boxes=tb, y0=yt[0], y1=yt[1], y2=yt[2],
                            a0=kat_anchors[0], a1=kat_anchors[1], a2=kat_anchors[2])

    # ---- IoL / best-fit on a sweep of box sizes (data/generators.py:2486-2544)
    rng = np.random.default_rng(5)
    wh = np.exp(rng.uniform(np.log(4), np.log(600), size=(256, 2))).astype(np.float32)
    sel = np.array([ref.gen.best_fit_and_layer(b, anchors)[:2] for b in wh], dtype=np.int32)
    iols = ref.gen.iol_common_center(np.concatenate(anchors, 0), wh)
    np.savez_compressed(f"{OUT}/bestfit.npz", wh=wh, layer_anchor=sel, iols=iols.astype(np.float32))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
