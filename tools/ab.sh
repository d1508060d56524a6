#!/bin/bash
# GPU A/B of library builds in ONE process group on ONE box: tools/ab.sh libA.so libB.so ...  (paths relative to the repo root)
for rep in 1 2; do
for lib in "$@"; do
  DYSF_LIB=$PWD/$lib python - "$lib" <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32, workers=1)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
for _ in range(5): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print("%-40s median %.1f us  min %.1f us" % (sys.argv[1], sorted(ts)[2], min(ts)))
PY
done; done
