"""Host-side image I/O of the data generator, importable WITHOUT torch: annotation parsing, decode + letterbox, the
per-image augmentation chain - and the worker process of the process-based loader (`python -m multigriddet_amd.host_io
<address> <authkey> <shm>`), which runs exactly these functions on another core, outside the trainer's GIL.

Reference: multigriddet/data/generators.py:1640-1700 (Sequence path: parse, decode, letterbox), :167-209 (zero-padded
letterbox), :1918-1943 (per-image augmentation, restated in data/host_aug.py)."""
import importlib.util
import os
import sys

import numpy as np

_ALLOC_TUNED = False
_HOST_AUG = None


def tune_host_allocators():
    """The loader threads / processes allocate and free megabyte-sized image buffers at a high rate.  glibc serves those
    with mmap/munmap (every buffer page-faults in again, and threads serialise on the process's address-space lock) and
    Pillow frees its image arenas at once: sixteen 608x608 PNGs took 78 ms on 8 threads, 28 ms with both caches on."""
    global _ALLOC_TUNED
    if _ALLOC_TUNED:
        return
    _ALLOC_TUNED = True
    try:
        from PIL import Image
        Image.core.set_blocks_max(256)                 # keep freed 16-MiB image blocks for reuse
    except Exception:
        pass
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)                      # M_MMAP_THRESHOLD: image-sized buffers from the heap
        libc.mallopt(-1, 1 << 30)                      # M_TRIM_THRESHOLD: and the heap keeps them
    except Exception:
        pass


def host_aug():
    """data/host_aug.py (numpy / PIL only), loaded by path so that a worker process does not import the package's
    torch-backed modules."""
    global _HOST_AUG
    if _HOST_AUG is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "host_aug.py")
        spec = importlib.util.spec_from_file_location("mgd_host_aug", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _HOST_AUG = mod                                # published only once complete: loader threads race here
    return _HOST_AUG


def parse_annotation_line(line):
    """'path x1,y1,x2,y2,cls ...' (reference generators.py:2425-2429) -> (path, boxes [n,5])."""
    parts = line.split()
    boxes = [list(map(float, p.split(",")))[:5] for p in parts[1:] if p]
    boxes = [b + [0.0] * (5 - len(b)) for b in boxes]
    boxes = np.array([b for b in boxes if any(v != 0 for v in b)], np.float32).reshape(-1, 5)
    return parts[0], boxes


def letterbox(image, boxes, target_hw, fill=0, dtype=np.float32):
    """Aspect-preserving resize (bicubic) + centred pad; boxes mapped along.  Training pads with zeros
    (tf.image.pad_to_bounding_box, reference generators.py:167-209); inference pads with 128
    (utils/preprocessing.py:46) - pass fill accordingly."""
    from PIL import Image
    th, tw = target_hw
    w, h = image.size
    r = min(tw / w, th / h)
    nw, nh = max(1, int(round(w * r))), max(1, int(round(h * r)))
    ox, oy = (tw - nw) // 2, (th - nh) // 2
    if (nw, nh) == (tw, th):
        canvas = image.resize((nw, nh), Image.BICUBIC)          # fills the frame: no pad to paste into
    else:
        canvas = Image.new("RGB", (tw, th), (fill, fill, fill))
        canvas.paste(image.resize((nw, nh), Image.BICUBIC), (ox, oy))
    out = boxes.copy()
    if len(out):
        out[:, [0, 2]] = out[:, [0, 2]] * r + ox
        out[:, [1, 3]] = out[:, [1, 3]] * r + oy
    return np.asarray(canvas, dtype), out


def load_image(line, target_shape, out_shape, seed, host_augment):
    """One image: decode -> letterbox to target_shape (-> bilinear resize to out_shape when they differ, the reference's
    cv2.resize at :1655) -> per-image augmentation chain.  8-bit until something needs fractions: without host
    augmentation the batch crosses PCIe as uint8 (a quarter of the bytes) and is widened on the device."""
    from PIL import Image
    target_shape, out_shape = tuple(target_shape), tuple(out_shape)
    path, boxes = parse_annotation_line(line)
    img = Image.open(path).convert("RGB")
    im, bx = letterbox(img, boxes, target_shape, fill=0, dtype=np.uint8)
    if out_shape != target_shape:
        oh, ow = out_shape
        im = np.asarray(Image.fromarray(im).resize((ow, oh), Image.BILINEAR))
        if len(bx):
            bx[:, [0, 2]] *= ow / target_shape[1]
            bx[:, [1, 3]] *= oh / target_shape[0]
    if host_augment:
        im, bx = host_aug().augment_image(np.random.default_rng(seed), im.astype(np.float32), bx, out_shape)
    return im, bx


def worker_main(address, authkey_hex, shm_name):
    """Loader worker: receives (slot_offset, line, target, out, seed, host_augment) over the connection, writes the image
    into the shared-memory batch buffer at slot_offset and answers (boxes bytes, count, error)."""
    from multiprocessing.connection import Client
    from multiprocessing import shared_memory
    tune_host_allocators()
    try:
        os.nice(5)      # the trainer's own threads (600 kernel launches per step) come first when the cores are oversubscribed
    except OSError:
        pass
    if authkey_hex == "-":                             # the key travels in the environment, not on the command line
        authkey_hex = os.environ.pop("MGD_LOADER_KEY")
    conn = Client(address, family="AF_UNIX", authkey=bytes.fromhex(authkey_hex))
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        # attaching registered the segment with THIS process's resource tracker, which would unlink it when the worker
        # exits (CPython < 3.13): the trainer owns it
        from multiprocessing import resource_tracker
        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:
        pass
    try:
        while True:
            try:
                task = conn.recv()
            except EOFError:
                break
            if task is None:
                break
            off, line, target, out, seed, host_augment = task
            try:
                im, bx = load_image(line, target, out, seed, host_augment)
                view = np.ndarray(im.shape, im.dtype, buffer=shm.buf, offset=off)
                view[...] = im
                del view
                conn.send((bx.astype(np.float32).tobytes(), len(bx), None))
            except Exception as e:                     # reported to the trainer, the worker lives on
                conn.send((b"", 0, f"{type(e).__name__}: {e}"))
    finally:
        shm.close()
        conn.close()


if __name__ == "__main__":
    worker_main(sys.argv[1], sys.argv[2], sys.argv[3])
