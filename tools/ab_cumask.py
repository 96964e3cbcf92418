"""Does confining the weight-gradient side stream to a subset of the CUs (hipExtStreamCreateWithCUMask) shorten the step?
The side stream's long-running blocks delay the short kernels of the critical chain (BatchNorm passes of the small
layers run 5-6x their solo time inside the step); a masked side stream always leaves CUs free for them."""
import ctypes, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigriddet_amd.engine import Network
from multigriddet_amd.train_step import TrainStep

dev = torch.device("cuda:0")
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    if rc != 0:
        print("hipExtStreamCreateWithCUMask failed rc", rc)
        return None
    return torch.cuda.ExternalStream(s.value, device=dev)


net = Network(80, 3, dev, seed=0)
ts = TrainStep(net, bench.coco_anchors(), 80, (608, 608), 16, lr=1e-4)
img, bx = bench.synth_batch(0, 16, 608)
img, bx = torch.from_numpy(img).to(dev), torch.from_numpy(bx).to(dev)
plain = net.wg_stream
MASKS = {"all256": [0xffffffff] * 8, "first128": [0xffffffff] * 4 + [0] * 4, "first96": [0xffffffff] * 3 + [0] * 5,
         "first160": [0xffffffff] * 5 + [0] * 3, "first64": [0xffffffff] * 2 + [0] * 6, "even128": [0x55555555] * 8,
         "lo16of32": [0x0000ffff] * 8, "lo24of32": [0x00ffffff] * 8, "lo8of32": [0x000000ff] * 8}
which = sys.argv[1] if len(sys.argv) > 1 else "plain"       # ONE extra stream per process: more HW queues slow everything
if which != "plain":
    st = masked_stream(MASKS[which])
    if st is None:
        sys.exit(1)
    net.wg_stream = st
res = []
for rnd in range(4):
    for _ in range(2):
        ts.step(img, bx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        ts.step(img, bx)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 8 * 1e3)
print(which, "ms/step median %.3f min %.3f" % (statistics.median(res), min(res)))
