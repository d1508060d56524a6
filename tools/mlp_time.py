"""GPU diagnostic: time of the fused MLP head (dysf_mlp_forward_dev) on 1024 rows; DYSF_LIB selects the build."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsd_amd
from rsd_amd.classifier import FusedMLPHead
head = FusedMLPHead.from_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "classifier_after.npz"))
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "classifier_after.npz"))
x = torch.from_numpy(np.tile(z["X"], (2, 1))[:1024].astype(np.float32)).cuda()
for _ in range(20): out = head(x)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(200): out = head(x)
ev[1].record(); torch.cuda.synchronize()
print("%-24s mlp head %.2f us per 1024 rows; checksum %.6f" % (os.path.basename(os.environ.get("DYSF_LIB", "libdysf_hip.so")), ev[0].elapsed_time(ev[1]) * 1e3 / 200, float(out.double().sum())))
