import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32)
ex = rsd_amd.FeatureExtractor(rsd_amd.REFERENCE_149, 0)
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
for name, pcm in (("int16", torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()),
                  ("float32", torch.from_numpy((bank[np.arange(1024) % 32].reshape(-1).astype(np.float32) / 32768.0)).cuda())):
    for _ in range(20): ex.extract_batch_device(pcm, off, 48000, pooled=True)
    ex.kernel_timing(True)
    ts = []
    for _ in range(30):
        ex.extract_batch_device(pcm, off, 48000, pooled=True); ts.append(ex.kernel_timing(True))
    print(name, "stft+mel median %.1f us" % (1e3 * np.median([t["stft_mel"] for t in ts])), "mfcc %.1f" % (1e3 * np.median([t["mfcc"] for t in ts])), "chroma %.1f" % (1e3 * np.median([t["chroma"] for t in ts])))
