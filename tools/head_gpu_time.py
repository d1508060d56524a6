"""GPU diagnostic: GPU-side time of the fused MLP head kernel (graph replay of 50 calls removes host enqueue cost)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsd_amd
from rsd_amd.classifier import FusedMLPHead
rng = np.random.default_rng(0)
dims = [240, 128, 64, 3]
h = FusedMLPHead(np.zeros(240), np.ones(240), [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)], [np.zeros(dims[i + 1]) for i in range(3)])
x = torch.randn(1024, 240, device="cuda")
for _ in range(5): h(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(50): y = h(x)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print("%-50s %.2f us per call (GPU, back to back)" % (os.environ.get("DYSF_LIB", "default").split("/")[-1], e0.elapsed_time(e1) * 1000 / 500))
