"""GPU diagnostic: per-kernel times of the HIP CNN head on the bench batch (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsd_amd
from rsd_amd.classifier import CNNHead, FusedCNNHead
fr = torch.randn(1024, 120, 304, device="cuda") * 20
fused = FusedCNNHead(CNNHead().cuda())
for _ in range(3): fused(fr, cols=301)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fused(fr, cols=301)
e1.record(); torch.cuda.synchronize()
print("cnn head: %.3f ms per 1024 clips" % (e0.elapsed_time(e1) / 20))
