#!/bin/bash
# usage: tools/head_time.sh libA.so libB.so ...  -- GPU time of the fused MLP head (1024 x 240 -> 128 -> 64 -> 3) per library build
for lib in "$@"; do
DYSF_LIB=$PWD/$lib python - "$lib" <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.classifier import FusedMLPHead
rng = np.random.default_rng(0)
dims = [240, 128, 64, 3]
h = FusedMLPHead(np.zeros(240), np.ones(240), [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(3)], [np.zeros(dims[i + 1]) for i in range(3)])
x = torch.randn(1024, 240, device="cuda")
for _ in range(20): h(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500): h(x)
e1.record(); torch.cuda.synchronize()
print("%-60s %.1f us per call (pipelined)" % (sys.argv[1], e0.elapsed_time(e1) * 2))
PY
done
