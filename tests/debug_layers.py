"""Diagnostic (not a test): per-layer relative L2 of the product's activations vs the oracle."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigriddet_amd.engine import Network
from oracle import model as om

torch.set_num_threads(8)
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 128
net = Network(80, 3, "cuda:0", seed=0)
params = om.init_params(seed=3)
net.load_keras_style(params)
img = np.random.default_rng(0).random((B, S, S, 3), dtype=np.float32)
acts = []
tp = om.torch_params(params)
with torch.no_grad():
    outs_ref = om.forward(torch.from_numpy(img), tp, training=True, acts_out=acts, emulate_bf16=(len(sys.argv) > 3))
net.training = True
outs = net.forward(torch.from_numpy(img).cuda())
torch.cuda.synchronize()
A = net._last
for i, cv in enumerate(net.layers):
    ref = acts[i].permute(0, 2, 3, 1).numpy().astype(np.float64)
    got = (A["a"][i] if cv.bn else A["y"][i]).float().cpu().numpy().astype(np.float64)
    r = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
    print(f"{i:3d} {cv.role:5s} {cv.cin:5d}->{cv.cout:5d} k{cv.k} s{cv.s} hw{A['hw'][i]}  rel-L2 {r:.4f}  |ref| {np.abs(ref).mean():.4f}")
