for rep in 1 2; do for t in 6 8 12 16 24; do DYSF_TILES=$t python - $t <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import rsd_amd
from rsd_amd.synth import synth_bank
bank = synth_bank(64, workers=8)
pcm = torch.from_numpy(bank[np.arange(1024) % 64].reshape(-1)).cuda()
off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
ex = rsd_amd.FeatureExtractor(rsd_amd.REFERENCE_149, 0)
for _ in range(3): ex.extract_batch_device(pcm, off, 48000, pooled=True)
ex.kernel_timing(True); ts = []
for _ in range(10):
    ex.extract_batch_device(pcm, off, 48000, pooled=True); ts.append(ex.kernel_timing(True))
print("tiles %s  stft+mel %.0f us" % (sys.argv[1], 1e3 * float(np.median([t["stft_mel"] for t in ts]))))
PY
done; done
