"""GPU diagnostic: host-side enqueue cost vs GPU time of one bench step (feature launch + head)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.classifier import FusedMLPHead, MLPHead
from rsd_amd.synth import synth_bank
bank = synth_bank(32, workers=1)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
rng = np.random.default_rng(0)
dims = [ex.pooled_dim, 128, 64, 3]
w = [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)]
b = [np.zeros(dims[i + 1]) for i in range(3)]
heads = {"fused": FusedMLPHead(np.zeros(dims[0]), np.ones(dims[0]), w, b), "eager": MLPHead(np.zeros(dims[0]), np.ones(dims[0]), w, b).cuda(), "none": None}
for name, head in heads.items():
    for frames in (True, False):
        def step():
            res = ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=frames)
            return head(res["pooled"]) if head is not None else res["pooled"]
        with torch.no_grad():
            for _ in range(10): step()
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n): step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("head=%-5s frames=%-5s enqueue %.1f us/step, total %.1f us/step" % (name, frames, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
x = torch.randn(1024, dims[0], device="cuda")
for name in ("fused", "eager"):
    h = heads[name]
    with torch.no_grad():
        for _ in range(10): h(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): h(x)
        e1.record(); torch.cuda.synchronize()
    print("head=%s alone: %.1f us per call" % (name, e0.elapsed_time(e1) * 10))
