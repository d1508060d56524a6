"""GPU diagnostic: does running the MLP head on a side stream (overlapping the next batch's STFT kernel) pay?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.classifier import FusedMLPHead
from rsd_amd.synth import synth_bank
bank = synth_bank(64, workers=8)
pcm = torch.from_numpy(bank[np.arange(1024) % 64].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
rng = np.random.default_rng(0)
dims = [ex.pooled_dim, 128, 64, 3]
head = FusedMLPHead(np.zeros(dims[0]), np.ones(dims[0]), [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)], [np.zeros(dims[i + 1]) for i in range(3)])
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
def step_serial():
    res = ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
    return head(res["pooled"]), res
def step_overlap():
    res = ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        res["pooled"].record_stream(side)
        lg = head(res["pooled"])
    return lg, res
for name, fn in (("serial", step_serial), ("overlap", step_overlap), ("serial", step_serial), ("overlap", step_overlap)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): lg, res = fn()
    main.wait_stream(side); torch.cuda.synchronize()
    print("%-8s %.1f us/step" % (name, (time.perf_counter() - t0) / 400 * 1e6))
