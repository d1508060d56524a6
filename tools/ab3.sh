#!/bin/bash
# same-box comparison of env-var variants of ONE library: tools/ab3.sh "VAR=val" "VAR2=val" ...   ("-" = no variable)
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e DYSF_VERBOSE=1 python - "$v" 2> gpurun_out/ab3.err <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32, workers=1)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
for _ in range(10): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
ex.kernel_timing(True); ts = []
for _ in range(30):
    ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True); ts.append(ex.kernel_timing(True))
print("%-28s stft %.1f us  mfcc %.1f us" % (sys.argv[1], 1e3 * np.median([t["stft_mel"] for t in ts]), 1e3 * np.median([t["mfcc"] for t in ts])))
PY
  grep -m1 "\[dysf\]" gpurun_out/ab3.err
done; done
