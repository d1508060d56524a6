"""GPU driver for profiling runs: N feature launches of the bench batch (1024 x 3 s clips), nothing else.
  python3 tools/kdrive.py [launches] [preset]      (DYSF_STFT=1 forces kernel A)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
preset = {"bench": rsd_amd.BENCH_512, "ref": rsd_amd.REFERENCE_149}[sys.argv[2] if len(sys.argv) > 2 else "bench"]
bank = synth_bank(32, workers=1)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(preset, 0)
fr = preset is rsd_amd.BENCH_512
fs = int(os.environ.get("KDRIVE_FRAME_STRIDE", "0")) or None
for _ in range(n): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=fr, frame_stride=fs)
torch.cuda.synchronize()
ex.kernel_timing(True); ts = []
for _ in range(10):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=fr, frame_stride=fs); ts.append(ex.kernel_timing(True))
print("stft %.1f us  mfcc %.1f us" % (1e3 * np.median([t["stft_mel"] for t in ts]), 1e3 * np.median([t["mfcc"] for t in ts])))
