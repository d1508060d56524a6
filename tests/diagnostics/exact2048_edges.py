import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rsd_amd
from rsd_amd.synth import synth_clip
def make(v):
    os.environ["DYSF_EXACT"] = str(v); ex = rsd_amd.FeatureExtractor(rsd_amd.REFERENCE_149, 0); os.environ.pop("DYSF_EXACT"); return ex
a, x = make(1), make(0)
clips = [synth_clip(i)[: 5000 + 3001 * i] for i in range(12)] + [np.zeros(0, np.int16), synth_clip(3)[:100]]
pcm = torch.from_numpy(np.concatenate(clips)).cuda()
off = torch.from_numpy(np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)).cuda()
ms = max(len(c) for c in clips)
for kw in (dict(pooled=False, frames=False, tuning=True), dict(pooled=True, frames=False, tuning=False), dict(pooled=False, frames=True, tuning=True)):
    ra = a.extract_batch_device(pcm, off, ms, **kw); rx = x.extract_batch_device(pcm, off, ms, **kw)
    torch.cuda.synchronize()
    for k in ra:
        if ra[k] is not None:
            assert torch.equal(ra[k], rx[k]), (kw, k)
    print(kw, "equal", {k: tuple(v.shape) for k, v in rx.items() if v is not None})
# all clips rejected
clips2 = [np.zeros(0, np.int16), synth_clip(1)[:50], synth_clip(2)[:4000]]
r1 = a.extract_batch(clips2, frames=True, tuning=True); r2 = x.extract_batch(clips2, frames=True, tuning=True)
for k in r1: assert np.array_equal(r1[k], r2[k]), k
print("rejected batch equal, statuses", r2["status"])
# a second launch after a long-clip launch (workspace growth / reuse)
long_ = [np.concatenate([synth_clip(j) for j in range(20)])]
r1 = x.extract_batch(long_, tuning=True); r2 = x.extract_batch(clips[:6], tuning=True); r3 = a.extract_batch(clips[:6], tuning=True)
assert np.array_equal(r2["pooled"], r3["pooled"]) and np.array_equal(r2["tuning"], r3["tuning"])
print("after a long-clip launch: equal")
