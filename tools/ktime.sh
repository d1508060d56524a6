#!/bin/bash
# usage: tools/ktime.sh VAR v1 v2 ...  -- per-kernel HIP-event times of the bench batch for several values of one env knob
var=$1; shift
for v in "$@"; do env $var=$v python - "$var=$v" <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32, workers=1)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
for _ in range(5): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
ex.kernel_timing(True); ts = []
for _ in range(20):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True); ts.append(ex.kernel_timing(True))
print("%-22s stft %.1f us  mfcc %.1f us" % (sys.argv[1], 1e3 * np.median([t["stft_mel"] for t in ts]), 1e3 * np.median([t["mfcc"] for t in ts])))
PY
done
