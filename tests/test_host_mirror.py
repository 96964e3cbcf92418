"""CPU: host-side mirror of the reference API (no GPU): config semantics, anchors/classes parsing against the
reference-generated golden file, class weights, LR schedule, the `multigriddet` import shim, CLI flag sets."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_shim_and_anchor_parsing_vs_reference_output():
    import multigriddet
    from multigriddet.utils import load_anchors, load_classes
    a = load_anchors(os.path.join(ROOT, "configs", "yolov3_coco_anchor.txt"))
    g = np.load(os.path.join(GOLDEN, "anchors.npz"))
    assert len(a) == 3 and all(np.array_equal(x, g[f"a{i}"]) for i, x in enumerate(a))
    assert len(load_classes(os.path.join(ROOT, "configs", "coco_classes.txt"))) == 80
    assert multigriddet.list_available_models() == ["multigriddet_darknet"]


def test_config_merge_and_validation(tmp_path):
    from multigriddet_amd.config import ConfigLoader
    base = {"a": {"x": 1, "y": {"z": 2}}, "b": [1, 2]}
    over = {"a": {"y": {"z": 3, "w": 4}}, "b": [9]}
    m = ConfigLoader.merge_configs(base, over)
    assert m == {"a": {"x": 1, "y": {"z": 3, "w": 4}}, "b": [9]} and base["a"]["y"]["z"] == 2
    with pytest.raises(FileNotFoundError):
        ConfigLoader.load_config(str(tmp_path / "nope.yaml"))
    with pytest.raises(KeyError):
        ConfigLoader.validate_config({"data": {}}, "training")
    with pytest.raises(ValueError):
        ConfigLoader.validate_config({"model_config": "m", "data": {}, "training": {"loss_option": 7}}, "training")
    r = ConfigLoader.resolve_paths({"p": "x/y.yaml", "q": ["a.txt", 3], "r": "/abs/z.h5", "s": "keep.me"}, str(tmp_path))
    assert r["p"] == str(tmp_path / "x/y.yaml") and r["q"][0] == str(tmp_path / "a.txt") and r["r"] == "/abs/z.h5"
    assert r["s"] == "keep.me"


def test_optimizer_factory_precedence():
    from multigriddet_amd.config import create_optimizer_from_config
    o = create_optimizer_from_config({"optimizer": {"type": "adamw", "learning_rate": 0.5}, "training": {"learning_rate": 0.01}})
    assert o.kind == "adamw" and float(o.learning_rate) == 0.01 and o.kwargs["weight_decay"] == 0.0005
    o = create_optimizer_from_config({"optimizer": {"type": "sgd", "learning_rate": 0.2}})
    assert o.kind == "sgd" and float(o.learning_rate) == 0.2 and o.kwargs["momentum"] == 0.937
    assert float(create_optimizer_from_config({}).learning_rate) == 0.001


def test_class_weights(tmp_path):
    from multigriddet_amd.utils import compute_class_weights
    f = tmp_path / "ann.txt"
    f.write_text("a.jpg 1,2,3,4,0 1,2,3,4,0 1,2,3,4,1\nb.jpg 5,5,9,9,2\nc.jpg\n")
    w = compute_class_weights(str(f), 4, "balanced")
    # reference formula: an absent class gets total/(C*1e-8), the mean normalisation then floors the rest at 0.1
    assert w.dtype == np.float32 and np.allclose(w, [0.1, 0.1, 0.1, 4.0])
    assert np.array_equal(compute_class_weights(str(tmp_path / "ann.txt"), 4, "other"), np.ones(4, np.float32))


def test_cosine_warmup_schedule_values():
    from multigriddet_amd.trainers import CosineAnnealingWithWarmup
    s = CosineAnnealingWithWarmup(1e-3, min_lr=1e-7, warmup_epochs=3, total_epochs=100, verbose=0)
    assert abs(s.lr_at(0) - (1e-5 + (1e-3 - 1e-5) / 3)) < 1e-12          # epoch 1 of warm-up
    assert abs(s.lr_at(2) - 1e-3) < 1e-12
    assert abs(s.lr_at(99) - 1e-7) < 1e-10                                # cosine reaches min_lr at the end
    assert s.lr_at(50) < s.lr_at(10)


def test_expansion_factor_contract():
    """Capacity factors 1/2/4/8 (reference tests/test_augmentation_capacity.py:108-268)."""
    from multigriddet_amd.data.generators import MultiGridDataGenerator
    anchors = [np.ones((3, 2))] * 3
    mk = lambda **k: MultiGridDataGenerator([], 4, (608, 608), anchors, 80, **k)._calculate_expansion_factor()
    assert mk(enhance_augment="mosaic", mosaic_prob=0.3, mixup_prob=0.1) == 8
    assert mk(enhance_augment="mosaic", mosaic_prob=0.3, mixup_prob=0.0) == 4
    assert mk(enhance_augment=None, mixup_prob=0.1) == 2
    assert mk(enhance_augment=None, mixup_prob=0.0) == 1


@pytest.mark.parametrize("script,flags", [("train.py", ["--config", "--weights", "--backbone-weights", "--resume", "--epochs", "--batch-size"]),
                                          ("infer.py", ["--config", "--input", "--output", "--weights", "--type", "--conf", "--nms", "--no-save", "--no-show"])])
def test_cli_flags(script, flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0
    for f in flags:
        assert f in out.stdout


def test_cli_missing_config_returns_1():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--config", "/nonexistent.yaml"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 1 and "Config file not found" in out.stdout


def test_letterbox_matches_reference_fixture():
    """G4: utils/preprocessing.letterbox_resize / preprocess_image against the reference's own outputs
    (tests/golden/letterbox.npz, generated by tests/golden/make_golden_letterbox.py from
    /root/reference/multigriddet/utils/preprocessing.py:12-90)."""
    import os
    from PIL import Image
    from conftest import GOLDEN
    from multigriddet_amd.utils.preprocessing import letterbox_resize, preprocess_image, letterbox_geometry
    g = np.load(os.path.join(GOLDEN, "letterbox.npz"))
    for i in range(int(g["n"])):
        pil = Image.fromarray(g[f"img{i}"])
        mh, mw = (int(v) for v in g[f"model_hw{i}"])
        boxed, size, off = letterbox_resize(pil, (mw, mh), return_padding_info=True)
        assert np.array_equal(np.asarray(boxed, np.uint8), g[f"boxed{i}"])
        assert (size[0], size[1], off[0], off[1]) == tuple(int(v) for v in g[f"pad{i}"])
        data = preprocess_image(pil, (mh, mw))
        assert data.dtype == np.float32 and data.shape == (1, mh, mw, 3)
        assert np.array_equal(data[0], g[f"boxed{i}"].astype(np.float32) / np.float32(255.0))
        nh, nw, dy, dx = letterbox_geometry(g[f"img{i}"].shape[:2], (mh, mw))
        assert (nw, nh, dx, dy) == tuple(int(v) for v in g[f"pad{i}"])


def test_resample_tables_reproduce_pil_bicubic_exactly():
    """The coefficient tables the device letterbox consumes: driving PIL's two integer passes with them (oracle/preprocess.py)
    gives PIL's Image.resize(BICUBIC) bit for bit, for up- and down-scaling, and the reference's letterbox fixture."""
    import os
    from PIL import Image
    from conftest import GOLDEN
    from multigriddet_amd.utils.preprocessing import resample_tables, letterbox_geometry
    from oracle import preprocess as op
    rng = np.random.default_rng(5)
    for (h, w), (nh, nw) in [((37, 53), (45, 64)), ((80, 60), (64, 48)), ((33, 100), (10, 31)), ((20, 20), (20, 20)),
                             ((17, 9), (60, 32))]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC), np.uint8)
        assert np.array_equal(op.resize_u8(img, (nh, nw), resample_tables), ref), ((h, w), (nh, nw))
    g = np.load(os.path.join(GOLDEN, "letterbox.npz"))
    for i in range(int(g["n"])):
        mh, mw = (int(v) for v in g[f"model_hw{i}"])
        assert np.array_equal(op.letterbox(g[f"img{i}"], (mh, mw), resample_tables, letterbox_geometry), g[f"boxed{i}"])


def test_process_loader_matches_in_process_decode(tmp_path):
    """The loader processes of the prefetching generator (python -m multigriddet_amd.host_io, no torch inside) return,
    through the shared-memory batch ring, exactly what host_io.load_image computes in this process - with and without the
    per-image host augmentation (seeded) - and the worker module really imports without torch."""
    import subprocess
    import sys
    import threading
    import numpy as np
    from PIL import Image
    from multigriddet_amd.data.generators import _ProcessLoader
    from multigriddet_amd.host_io import load_image
    rng = np.random.default_rng(5)
    lines = []
    for i in range(6):
        h, w = (90, 120) if i % 2 else (128, 100)
        img = (rng.random((h, w, 3)) * 255).astype(np.uint8)
        path = str(tmp_path / f"im{i}.png")
        Image.fromarray(img).save(path)
        lines.append(f"{path} 10,12,{40 + i},50,{i} 5,6,30,31,7")
    out = subprocess.run([sys.executable, "-c", "import sys, multigriddet_amd.host_io; print('torch' in sys.modules)"],
                         capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "False", out.stdout + out.stderr
    S = 96
    for aug, dtype in ((False, np.uint8), (True, np.float32)):
        per = S * S * 3 * np.dtype(dtype).itemsize
        pl = _ProcessLoader(2, 1, len(lines) * per)
        try:
            res = [None] * len(lines)
            fin = threading.Semaphore(0)

            def done(r, j):
                res[j] = r
                fin.release()
            for j, line in enumerate(lines):
                pl.submit(pl.begin_epoch(), (j * per, line, (S, S), (S, S), 100 + j, aug), (lambda r, j=j: done(r, j)))
            for _ in lines:
                assert fin.acquire(timeout=120)
            for j, line in enumerate(lines):
                raw, cnt, err = res[j]
                assert err is None, err
                im_ref, bx_ref = load_image(line, (S, S), (S, S), 100 + j, aug)
                got = np.ndarray((S, S, 3), dtype, buffer=pl.shm.buf, offset=j * per).copy()
                assert im_ref.dtype == dtype and np.array_equal(got, im_ref)
                assert cnt == len(bx_ref) and np.array_equal(np.frombuffer(raw, np.float32).reshape(cnt, 5), bx_ref)
        finally:
            pl.close()


def test_prefetch_iterator_machinery_on_cpu():
    """The prefetching iterator (loader processes and loader threads, two epochs, multi-scale draws, host augmentation on and
    off) yields the same batches in both modes; torch.cuda is stubbed out inside the harness interpreter."""
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "prefetch_cpu_harness.py")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "aug True ok 6" in out.stdout and "aug False ok 6" in out.stdout, out.stdout + out.stderr
