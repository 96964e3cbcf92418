"""Process-wide HIP streams of the package, one per (role, device).

ROCm maps a process's streams onto a small number of hardware queues (four by default); every further stream that carries
work shares a queue with an earlier one.  A training process that gave every object a stream of its own - each Network a
weight-gradient stream, each TrainStep a main stream, each prefetching generator a copy stream - ran its steps 20 % slower as
soon as a second generator (validation data beside training data) existed: the weight-gradient stream shared a hardware queue
with the main stream and the two-stream backward serialised (measured: 11.9 -> 14.6 ms per step with three generators alive,
tests/test_gpu_host_loop.py).  So the roles are singletons: every object of this process that needs "the copy stream" of a
device gets the same one.  Objects that share a stream are serialised against each other by it, which is what a single Python
thread driving them does anyway.  No reference counterpart (TensorFlow owns its streams)."""
import threading

import torch

_LOCK = threading.Lock()
_STREAMS = {}


def shared_stream(role, device=None, priority=0):
    """The stream of `role` ("main", "wgrad", "copy", "comm", "capture") on `device` (default: the current device)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (role, idx)
    with _LOCK:
        s = _STREAMS.get(key)
        if s is None:
            s = _STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", idx), priority=priority)
        return s
