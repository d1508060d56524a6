"""GPU diagnostic: a 16384-clip resident batch through one launch vs LaneExecutor.extract_large (chunks over lanes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rsd_amd
from rsd_amd.pipeline import LaneExecutor
from rsd_amd.synth import synth_bank
bank = synth_bank(64, workers=8)
n = 16384
pcm = torch.from_numpy(np.tile(bank.reshape(-1), n // 64)).cuda()
off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
lanes = LaneExecutor(rsd_amd.BENCH_512, 0, 3, first_extractor=ex)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
a = t(lambda: ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=False))
print("one launch           : %.2f ms  %.2f M clips/s" % (a * 1e3, n / a / 1e6))
for chunk in (1024, 2048, 4096):
    b = t(lambda: lanes.extract_large(pcm, off, 48000, chunk_clips=chunk, pooled=True, frames=False))
    print("extract_large %5d  : %.2f ms  %.2f M clips/s" % (chunk, b * 1e3, n / b / 1e6))
