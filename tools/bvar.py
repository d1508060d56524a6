import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32, workers=1)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
for fr, po in ((True, True), (False, True), (True, False)):
    for _ in range(10): ex.extract_batch_device(pcm, off, 48000, pooled=po, frames=fr)
    ex.kernel_timing(True); ts = []
    for _ in range(30):
        ex.extract_batch_device(pcm, off, 48000, pooled=po, frames=fr); ts.append(ex.kernel_timing(True))
    print("frames=%s pooled=%s  stft %.1f us  mfcc %.1f us" % (fr, po, 1e3 * np.median([t["stft_mel"] for t in ts]), 1e3 * np.median([t["mfcc"] for t in ts])))
