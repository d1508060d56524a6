"""GPU diagnostic: frame rows of the float16-split DCT (default) and of the float32-MFMA DCT (DYSF_DCT16=0) against the
oracle's frame rows recomputed in float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rsd_amd
from oracle import dysf_oracle as O
from rsd_amd.synth import synth_clip
clips = [synth_clip(i) for i in (0, 1, 2, 5, 63, 127, 191, 255)]
def mk(v):
    os.environ["DYSF_DCT16"] = v
    ex = rsd_amd.FeatureExtractor(rsd_amd.BENCH_512, 0)
    os.environ.pop("DYSF_DCT16")
    return ex
h16, f32 = mk("1"), mk("0")
a, b = h16.extract_batch(clips, frames=True), f32.extract_batch(clips, frames=True)
ref = [O.frame_level_stack(O.pcm16_to_float(c), O.BENCH_512) for c in clips]
for name, r in (("float16 split", a), ("float32 MFMA", b)):
    e_all, e0 = 0.0, 0.0
    for i, c in enumerate(clips):
        d = np.abs(r["frames"][i][:, :ref[i].shape[1]].astype(np.float64) - ref[i])
        e_all = max(e_all, float(d[1:40].max()), float(d[41:].max())); e0 = max(e0, float(d[0].max()))
    print(f"{name:14s} max |frame row - oracle|: MFCC row 0 {e0:.3e}, all other rows {e_all:.3e}")
print("float16 split vs float32 MFMA: max |d| frames %.3e, pooled %.3e" % (np.abs(a["frames"] - b["frames"]).max(), np.abs(a["pooled"] - b["pooled"]).max()))
