#!/bin/bash
# Runs ON THE GPU BOX: shader clock (rocm-smi / sysfs) sampled while a workload loops for a few seconds.
probe() {
  for i in 1 2 3 4; do
    sleep 0.7
    (rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2 | tr '\n' ' '; echo) 
    (rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1)
  done
}
echo "== idle"; probe
echo "== CNN head loop"
python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.classifier import CNNHead, FusedCNNHead
fr = torch.randn(1024, 120, 304, device="cuda") * 20
fused = FusedCNNHead(CNNHead().cuda())
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(50): fused(fr, cols=301)
    torch.cuda.synchronize()
PY
sleep 2.5; probe; wait
echo "== feature kernels (BENCH_512) loop"
python - <<'PY' &
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(32)
pcm = torch.from_numpy(bank[np.arange(1024) % 32].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(200): ex.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
    torch.cuda.synchronize()
PY
sleep 2.5; probe; wait
echo "== packed-fma microbenchmark loop"
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate 2>/dev/null
(for i in 1 2 3 4 5 6; do /tmp/valu_rate > /dev/null 2>&1; done) &
sleep 1.0; probe; wait
