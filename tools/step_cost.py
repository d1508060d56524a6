"""GPU diagnostic: wall time per bench step for different step counts / event usage (is the loop host bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.classifier import FusedMLPHead
from rsd_amd.synth import synth_bank
NB = int(os.environ.get("STEP_BANK", "64"))
bank = synth_bank(NB, workers=8)
pcm = torch.from_numpy(bank[np.arange(1024) % NB].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
rng = np.random.default_rng(0)
dims = [ex.pooled_dim, 128, 64, 3]
head = FusedMLPHead(np.zeros(dims[0]), np.ones(dims[0]), [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)], [np.zeros(dims[i + 1]) for i in range(3)])
def step(ev):
    if ev:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    res = ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
    lg = head(res["pooled"])
    if ev: e1.record()
    return lg, res
for _ in range(10): step(False)
torch.cuda.synchronize()
for n in (50, 200, 800):
    for ev in (False, True):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): lg, res = step(ev)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("steps=%4d events=%-5s enqueue %.1f us/step  total %.1f us/step" % (n, ev, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))

# sustained run in windows of 50 steps: does the per-step time drift?
torch.cuda.synchronize()
for w in range(8):
    t0 = time.perf_counter()
    for _ in range(50): lg, res = step(False)
    torch.cuda.synchronize()
    print("window %d: %.1f us/step" % (w, (time.perf_counter() - t0) / 50 * 1e6))
