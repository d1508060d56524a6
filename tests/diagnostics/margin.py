"""GPU diagnostic: error margins of the bench preset (BENCH_512) against the CPU oracle on synthetic clips, pooled and
frame level.  usage: python tests/diagnostics/margin.py [n_clips]   (DYSF_DCT64=1 selects the float64-MFMA DCT for comparison)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa: F401
import rsd_amd
from oracle import dysf_oracle as O
from rsd_amd.synth import synth_clip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ids = list(range(n)) + [64, 128, 192, 256]   # every 64th clip of the bank is an edge case
ex = rsd_amd.get_extractor(rsd_amd.BENCH_512, 0)
clips = [synth_clip(i) for i in ids]
res = ex.extract_batch(clips, frames=True)
worst_p, worst_f, allp = 0.0, 0.0, []
for k, (i, y) in enumerate(zip(ids, clips)):
    pooled_ref = O.extract_audio_features(O.pcm16_to_float(y), 16000, O.BENCH_512)
    rows_ref = O.frame_level_stack(O.pcm16_to_float(y), O.BENCH_512)
    T = rows_ref.shape[1]
    dp = np.abs(res["pooled"][k] - pooled_ref)
    df = np.abs(res["frames"][k][:, :T] - rows_ref)
    allp.append(dp)
    worst_p = max(worst_p, dp.max()); worst_f = max(worst_f, df.max())
    if dp.max() > 5e-5:
        print("clip %d pooled max %.2e at %d (value %.3f)" % (i, dp.max(), dp.argmax(), pooled_ref[dp.argmax()]))
allp = np.stack(allp)
print("DCT64=%s clips=%d pooled: max %.2e  p99 %.2e  mean %.2e | mfcc_mean_0 max %.2e | frames max %.2e" % (
    os.environ.get("DYSF_DCT64", "0"), len(ids), worst_p, np.quantile(allp, 0.99), allp.mean(), allp[:, 0].max(), worst_f))
