"""Timeline of ONE steady-state folded inference forward from a rocprofv3 kernel trace: per launch the kernel's duration and
the idle gap in front of it.  Two modes:
  trace_timeline.py run [batch] [graph]      the traced program (forwards only).  Run it as the profiler's direct child -
                                             cd /tmp && export TMPDIR=/tmp &&
                                             rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/trace_timeline.py run 1
                                             (never through env / bash -c / a #! script: the profiler has initialised the GPU)
  trace_timeline.py show <dir> [launches]    parse <dir>/**/*kernel_trace.csv, print the last forward's launches"""
import csv, glob, os, sys

if sys.argv[1] == "run":
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multigriddet_amd.models import build_multigriddet_darknet
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda:0")
    model, _ = build_multigriddet_darknet(input_shape=(608, 608, 3), num_classes=80)
    model.fold_bn(True)
    if len(sys.argv) > 3 and sys.argv[3] == "graph":
        model.enable_graph(True)
    img = torch.from_numpy(np.random.default_rng(0).random((B, 608, 608, 3), dtype=np.float32)).to(dev)
    for _ in range(8):
        model(img, training=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        model(img, training=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"forward, batch {B}: {e0.elapsed_time(e1) / 10:.3f} ms")
    sys.exit(0)

rows = []
for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# one forward = the launches between two stem kernels
stems = [i for i, n in enumerate(names) if "stem_fwd" in n or "stem_im2col" in n]
i0, i1 = stems[-2], stems[-1]
fw = rows[i0:i1]
busy = sum(e - s for s, e, _ in fw)
wall = rows[i1][0] - rows[i0][0]
print(f"{len(fw)} launches per forward; kernel time {busy / 1e3:.1f} us, wall {wall / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us")
prev_end = None
for s, e, n in fw:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {short}")
    prev_end = e
