"""CPU harness for MultiGridDataGenerator._iter_prefetch (run in its own interpreter by tests/test_host_mirror.py): the few
torch.cuda calls the iterator makes are replaced by inert stand-ins, so that the prefetch machinery itself - producer thread,
loader processes, shared-memory ring, per-batch completion, epoch restart - runs here without a GPU, and the batches of
the process-based, the thread-based and the synchronous-order draws can be compared."""
# exercise _iter_prefetch on CPU by faking the few torch.cuda calls it makes
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
import multigriddet_amd.data.generators as G
class FakeStream:
    def __init__(self, device=None, priority=0): pass
    def __enter__(self): return self
    def __exit__(self,*a): return False
class FakeEvent:
    def record(self, s=None): pass
    def synchronize(self): pass
torch.cuda.is_available = lambda: True
torch.cuda.Stream = FakeStream
torch.cuda.stream = lambda s: s
torch.cuda.Event = FakeEvent
class CS:
    def wait_event(self, e): pass
torch.cuda.current_stream = lambda: CS()
_dev = {"main": 3}
import threading
torch.cuda.current_device = lambda: _dev.get(threading.get_ident(), _dev["main"])
def _set_device(i): _dev[threading.get_ident()] = i
torch.cuda.set_device = _set_device
torch.Tensor.cuda = lambda self, non_blocking=False: self
torch.Tensor.record_stream = lambda self, s: None
G.MultiGridDataGenerator._host_buffer = staticmethod(lambda shape, dtype, pinned: np.zeros(shape, dtype))
G.MultiGridDataGenerator._device_part = lambda self, img, bx: (img, bx, [bx])
import tempfile; tmp = tempfile.mkdtemp(prefix='mgd_prefetch_')
rng=np.random.default_rng(0); lines=[]
for i in range(10):
    im=(rng.random((64,80,3))*255).astype(np.uint8); p=f'{tmp}/{i}.png'; Image.fromarray(im).save(p); lines.append(f'{p} 1,2,30,40,3')
anchors=[np.ones((3,2),np.float32)]*3
def run(mode, aug):
    g=G.MultiGridDataGenerator(lines, 4, (64,64), anchors, 80, augment=aug, shuffle=True, seed=1, num_workers=3, prefetch_factor=2, worker_mode=mode, rescale_interval=2)
    out=[]
    for ep in range(2):
        for (x, _z) in g:
            out.append((x[0].clone(), x[1].clone()))
        g.on_epoch_end()
    g.close()
    return out
for aug in (False, True):
    a=run('process', aug); b=run('thread', aug)
    assert len(a)==len(b)==6, (len(a), len(b))
    for (ia,ba),(ib,bb) in zip(a,b):
        assert torch.equal(ia.float(), ib.float()) and torch.equal(ba, bb)
    print('aug', aug, 'ok', len(a))

# abandoned epoch: leave an epoch after ONE batch; the next epoch's batches must still equal thread mode -
# a stale task of the abandoned epoch must neither run later nor overwrite a slot the new epoch has filled
def run_abandon(mode):
    g=G.MultiGridDataGenerator(lines, 2, (64,64), anchors, 80, augment=False, shuffle=True, seed=1, num_workers=3, prefetch_factor=3, worker_mode=mode)
    out=[]
    for rep in range(3):
        it = iter(g)
        x, _z = next(it)
        out.append((x[0].clone(), x[1].clone()))
        it.close()                                   # the generator's finally: runs the teardown
        g.on_epoch_end()
    for (x, _z) in g:
        out.append((x[0].clone(), x[1].clone()))
    devs = list(g._thread_devices)
    g.close()
    return out, devs
(a, da), (b, db) = run_abandon('process'), run_abandon('thread')
assert len(a) == len(b) == 3 + 5, (len(a), len(b))
for (ia,ba),(ib,bb) in zip(a,b):
    assert torch.equal(ia.float(), ib.float()) and torch.equal(ba, bb)
assert da == [3, 3] and db == [3, 3], (da, db)    # both helper threads bound to the creator's device, not to device 0
print('abandon ok', len(a))
