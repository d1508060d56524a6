"""GPU diagnostic: per-kernel HIP-event times of the REFERENCE_149 and MAIN_286 presets on 1024 x 3 s clips."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(64, workers=8)
pcm = torch.from_numpy(bank[np.arange(1024) % 64].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
import dataclasses
for name, p in (("BENCH_512", rsd_amd.BENCH_512), ("REFERENCE_149 (parity)", rsd_amd.REFERENCE_149),
                ("149, float32 FFT", dataclasses.replace(rsd_amd.REFERENCE_149, parity=False)), ("MAIN_286", rsd_amd.MAIN_286),
                ("149 without chroma", rsd_amd.FeatureParams(chroma=False))):
    ex = rsd_amd.FeatureExtractor(p, 0)
    for _ in range(3): ex.extract_batch_device(pcm, off, 48000, pooled=True)
    ex.kernel_timing(True); ts = []
    for _ in range(10):
        ex.extract_batch_device(pcm, off, 48000, pooled=True); ts.append(ex.kernel_timing(True))
    med = lambda k: 1e3 * float(np.median([t[k] for t in ts]))
    print("%-24s stft+mel %.0f us  mfcc %.0f us  chroma(+rich) %.0f us" % (name, med("stft_mel"), med("mfcc"), med("chroma")))
