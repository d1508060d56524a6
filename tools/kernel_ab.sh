#!/bin/bash
# GPU diagnostic: same-box A/B of the feature kernels: the built library against a build with an extra -D<macro> (argument 1;
# the macro has to exist in the sources -- DYSF_NT_STORES of DESIGN.md section 9 was a temporary one and is not in the tree).
# Prints per-kernel HIP-event times of the default preset, three rounds each (FRAME_STRIDE=n: frame-row pitch of the output).
macro=${1:?usage: kernel_ab.sh MACRO}
cd recognizing-speech-dysfluencies-in-stuttering_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -D$macro -shared dysf_kernels.hip dysf_capi.hip dysf_forest.hip dysf_svc.hip dysf_mlp.hip dysf_resample.hip dysf_cnn.hip dysf_hfratio.hip dysf_clean.hip -o /tmp/libdysf_ab.so 2>/dev/null || exit 1
cd ../..
cat > /tmp/ab_time.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
fs = int(os.environ["FRAME_STRIDE"]) if os.environ.get("FRAME_STRIDE") else None
kw = dict(pooled=True, frames=True, frame_stride=fs)
for _ in range(100): ex.extract_batch_device(pcm, off, 48000, **kw)
ex.kernel_timing(True)
ts = []
for _ in range(40):
    for _ in range(3): ex.extract_batch_device(pcm, off, 48000, **kw)
    ts.append(ex.kernel_timing(True))
print("stft+mel %.1f us   mfcc %.1f us" % (1e3 * np.median([t["stft_mel"] for t in ts]), 1e3 * np.median([t["mfcc"] for t in ts])))
PY
for rep in 1 2 3 4 5; do
  echo -n "as built:   "; python /tmp/ab_time.py 2>/dev/null | tail -1
  echo -n "-D$macro: "; DYSF_LIB=/tmp/libdysf_ab.so python /tmp/ab_time.py 2>/dev/null | tail -1
done
