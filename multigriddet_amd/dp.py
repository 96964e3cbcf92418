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


def init_distributed(share_gpu=None):
    """One process per GPU under torch.distributed.run: bind this rank to cuda:LOCAL_RANK BEFORE anything touches the
    GPU, then create the default process group on RCCL (backend "nccl").  Returns (rank, world, device).  No-op for
    WORLD_SIZE <= 1.  share_gpu (default: MGD_BENCH_SHARE_GPU=1): rehearsal on a one-GPU box - every rank on cuda:0,
    gloo for the exchange (RCCL refuses two ranks on one device)."""
    import os
    # read by the HSA runtime when it initialises (the first torch.cuda call of the process): set it before that, or the
    # launcher has to export it (the host driver only supports dmabuf IPC)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if share_gpu is None:
        share_gpu = os.environ.get("MGD_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local = 0
    dev = torch.device("cuda", local)
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu or not torch.cuda.is_available():
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    return rank, world, dev


def all_reduce_mean_scalar(value, world):
    """Mean of a python float over the ranks (validation loss, so that every rank's callbacks see the same number)."""
    if world <= 1 or not dist.is_initialized():
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / world


def broadcast_flag(flag, world, src=0):
    """Rank `src`'s boolean for everyone (stop_training): ranks must leave the epoch loop together or the next bucket
    all-reduce hangs."""
    if world <= 1 or not dist.is_initialized():
        return bool(flag)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.broadcast(t, src=src)
    return bool(t.item())


def broadcast_tensors(tensors, world, src=0):
    """Rank `src`'s contents of every tensor for everyone, in place (initial weights, BatchNorm moving statistics,
    optimiser state).  RCCL moves device tensors directly; under gloo (CPU tests, the shared-GPU rehearsal) CUDA tensors
    are staged through the host."""
    if world <= 1 or not dist.is_initialized():
        return
    direct = dist.get_backend() == "nccl"
    for t in tensors:
        if t is None or t.numel() == 0:
            continue
        if direct or not t.is_cuda:
            dist.broadcast(t, src=src)
        else:
            h = t.detach().cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)


def shard_lines(lines, rank, world):
    """Rank r's share of the annotation list, truncated so that EVERY rank has the same number of lines
    (len // world): steps_per_epoch, and with it the sequence of collectives, is then identical across ranks."""
    if world <= 1:
        return list(lines)
    per = len(lines) // world
    if per == 0:
        raise ValueError(f"{len(lines)} annotation lines cannot be sharded over {world} ranks")
    return list(lines[rank:per * world:world])


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


class CabiComm:
    """The library's own RCCL binding (include/mgd_hip.h `mgd_comm_*`): one communicator per process on the current
    device.  The 128-byte id is drawn by rank 0 and travels through the already initialised torch.distributed group
    (any backend - it is host bytes)."""

    def __init__(self, rank, world, device):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        lib = L.load()
        idbuf = (C.c_ubyte * 128)()
        if rank == 0:
            L.check(lib.mgd_comm_unique_id(idbuf), "comm_unique_id")
        if world > 1:
            box = [bytes(idbuf)]
            dist.broadcast_object_list(box, src=0)
            idbuf = (C.c_ubyte * 128).from_buffer_copy(box[0])
        self.comm = C.c_void_p()
        with torch.cuda.device(device):
            L.check(lib.mgd_comm_init(C.byref(self.comm), int(rank), int(world), idbuf), "comm_init")

    def all_reduce_sum_(self, flat_fp32):
        """In place, asynchronous on the current stream."""
        L, C = self._L, self._C
        assert flat_fp32.dtype == torch.float32 and flat_fp32.is_cuda and flat_fp32.is_contiguous()
        L.check(L.load().mgd_comm_allreduce_bucket(self.comm, L.ptr(flat_fp32), C.c_int64(flat_fp32.numel()),
                                                   L.stream_ptr()), "comm_allreduce_bucket")

    def destroy(self):
        if self.comm:
            self._L.check(self._L.load().mgd_comm_destroy(self.comm), "comm_destroy")
            self.comm = self._C.c_void_p()


class _Done:
    def wait(self):
        pass


class GradBuckets:
    def __init__(self, grads, layer_offsets, world_size, bucket_mb=32.0, comm_stream=None, producer_streams=(),
                 cabi_comm=None):
        """cabi_comm: a CabiComm - buckets then go through mgd_comm_allreduce_bucket on the communication stream
        instead of torch.distributed (opt-in, MGD_DP_COMM=cabi in TrainStep; same bucket order, same overlap)."""
        self.cabi = cabi_comm
        self.grads = grads
        self.producer_streams = tuple(producer_streams)     # side streams that also write gradients (wgrad)
        self.world = world_size
        self.buckets = make_buckets(layer_offsets, grads.numel(), int(bucket_mb * 1e6 / 4))
        self.comm_stream = comm_stream
        self.after_bucket = None     # hook(bucket_index, begin, end): enqueued on the communication stream BEHIND the
        #                              bucket's all-reduce (TrainStep: optimiser + weight re-pack of that slice)
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
                    if self.cabi is not None:
                        self.cabi.all_reduce_sum_(sl)
                        work = _Done()
                    else:
                        work = dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True)
                    if self.after_bucket is not None:
                        work.wait()        # RCCL: a stream dependency (comm stream waits for the collective), no host block
                        self.after_bucket(self._next - 1, b, e)
                        work = _Done()
                    self._works.append(work)
            else:
                self._works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        self.on_layer_done(0)
        for w in self._works:
            w.wait()
        if self.cabi is not None and self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)     # the optimiser reads the summed gradients
        self._works = []
