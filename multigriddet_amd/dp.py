"""Data-parallel gradient exchange for the flat gradient buffer (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU path at all: this is new design.  Gradients live in
one flat fp32 buffer in layer order; backward finalises them from the last layer to the first, so the
buffer is cut into contiguous buckets from the END and each bucket's all-reduce (sum) is enqueued on
a side stream as soon as backward has passed its first layer - overlapping the collective with the
remaining dgrad/wgrad kernels.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), ring collectives
are per-link bound, hence few, large (~32 MB) buckets.  Averaging (1/world) is folded into the
optimiser kernel's grad_scale.  Works on CPU tensors with the gloo backend too (tests).
"""
import torch
import torch.distributed as dist


def make_buckets(layer_offsets, n_params, bucket_elems):
    """layer_offsets[i] = first flat index of layer i (ascending).  Returns [(first_layer, begin, end)],
    ordered from the last layers to the first, tiling [0, n_params) exactly."""
    buckets = []
    end = n_params
    for i in range(len(layer_offsets) - 1, -1, -1):
        begin = layer_offsets[i]
        if end - begin >= bucket_elems or i == 0:
            buckets.append((i, begin, end))
            end = begin
    return buckets


class GradBuckets:
    def __init__(self, grads, layer_offsets, world_size, bucket_mb=32.0, comm_stream=None, producer_streams=()):
        self.grads = grads
        self.producer_streams = tuple(producer_streams)     # side streams that also write gradients (wgrad)
        self.world = world_size
        self.buckets = make_buckets(layer_offsets, grads.numel(), int(bucket_mb * 1e6 / 4))
        self.comm_stream = comm_stream
        self.reset()

    def reset(self, lo=0):
        self._next = 0
        self._works = []
        self._lo = lo          # flat indices below `lo` are frozen: never exchanged

    def on_layer_done(self, i):
        """Every gradient of layers >= i is final: launch the buckets that start at or after layer i."""
        if self.world <= 1:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= i:
            _, b, e = self.buckets[self._next]
            self._next += 1
            b = max(b, self._lo)
            if e <= b:
                continue
            sl = self.grads[b:e]
            if self.comm_stream is not None:
                ev = torch.cuda.Event()
                ev.record()
                self.comm_stream.wait_event(ev)
                for ps in self.producer_streams:
                    self.comm_stream.wait_stream(ps)
                with torch.cuda.stream(self.comm_stream):
                    self._works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))
            else:
                self._works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        self.on_layer_done(0)
        for w in self._works:
            w.wait()
        self._works = []
