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
