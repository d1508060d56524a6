"""GPU diagnostic: do the feature kernels themselves get faster over the first few hundred launches of a process?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32, workers=1)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
ex.kernel_timing(True)
ts = []
for i in range(400):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
    ts.append(ex.kernel_timing(True))
for a in range(0, 400, 25):
    blk = ts[a:a + 25]
    print("launches %3d-%3d  stft %.1f us  mfcc %.1f us" % (a, a + 24, 1e3 * np.mean([t["stft_mel"] for t in blk]), 1e3 * np.mean([t["mfcc"] for t in blk])))
