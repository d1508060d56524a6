"""GPU diagnostic: chunked cleaning vs the oracle around a chunk boundary."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import rsd_amd
from oracle import dysf_oracle as O, dysf_clean_oracle as CO
from rsd_amd.synth import synth_clip
rng = np.random.default_rng(161)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 601000
y = (O.pcm16_to_float(synth_clip(61, n=n)) * np.float32(0.5) + np.float32(0.01) * rng.standard_normal(n).astype(np.float32)).astype(np.float32)
cl = rsd_amd.AudioCleaner(sr=16000, device=0, prop_decrease=0.8)
got = cl.clean_batch([y], denoised=True, pcm16=False)["denoised"][0].astype(np.float64)
ref = CO.reduce_noise(y, 16000, 0.8).astype(np.float64)
d = np.abs(got - ref)
print("n", n, "max diff", d.max(), "at", int(d.argmax()), "peak", np.abs(ref).max())
for a, b in ((0, 600000), (600000, n)):
    print("span", a, b, "max diff", d[a:b].max(), "at", a + int(d[a:b].argmax()))
i0 = 600000
for i in list(range(i0 - 3, min(n, i0 + 8))) + list(range(n - 4, n)):
    print(i, got[i], ref[i], got[i] / ref[i] if ref[i] else 0)
if n > 600000 and n < 660000:
    m = n - 600000
    g = got[600000:]
    best = None
    for R in list(range(0, 4096, 256)) + [29000, 30000, 30256, 60000, 629000 + 1000 - m]:
        for left in (30000, 30000 - 256, 30000 + 256, 30000 - 512, 30000 + 512):
            chunk = np.zeros(left + m + R); chunk[:left + m] = y[600000 - left:n]
            o = CO.spectral_gate_nonstationary(chunk, 16000, 0.8)[left:left + m].astype(np.float32)
            e = float(np.abs(o - g).max())
            if best is None or e < best[0]: best = (e, R, left)
    print("closest variant (right zeros R, left context): max |gpu - variant| = %.3e at R=%d left=%d" % best)
