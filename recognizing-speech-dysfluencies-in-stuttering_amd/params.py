"""Feature parameters -- the module-level constants of the reference turned into one frozen record.

pipeline1.py:77-86 / main1.py:578-583 fix ``TARGET_SR = 16000``, ``MFCC_N = 20`` and otherwise take librosa's
defaults (n_fft 2048, hop 512, 128 mels, top_db 80, delta width 9); main.py:621-630 sets
``MFCC_N = 40, N_FFT = 512, HOP_LEN = 256``.  BASELINE.json configs[1] asks for 512 / 160 / 64 mel / 40 MFCC."""
import dataclasses
from typing import Optional

from . import _lib

TARGET_SR = 16000          # pipeline1.py:78
TEXT_FEATURE_LEN = 5       # pipeline1.py:85


@dataclasses.dataclass(frozen=True)
class FeatureParams:
    sr: int = TARGET_SR
    n_fft: int = 2048
    hop: int = 512
    n_mels: int = 128
    n_mfcc: int = 20
    fmin: float = 0.0
    fmax: Optional[float] = None
    top_db: float = 80.0
    amin: float = 1e-10
    delta_width: int = 9
    chroma: bool = True
    rich: bool = False         # main.py:738-754: + spectral-contrast mean/std (14), mean zcr, mean rms, mean centroid
    # Reproduce the reference's rounding steps where they decide results (DysfParams.parity): float64 FFT rounded to
    # complex64 + NumPy's complex abs (librosa.stft, p1:216/227), NumPy's pairwise float32 mean / std (p1:220-221).
    # Needed for the tuning histogram of chroma_stft to follow the reference on every clip; False = float32 FFT,
    # float64 statistics (the throughput path of the bench preset).
    parity: bool = True

    @property
    def n_bins(self) -> int:
        return self.n_fft // 2 + 1

    @property
    def audio_feature_len(self) -> int:       # AUDIO_FEATURE_LEN, pipeline1.py:83
        return 6 * self.n_mfcc + (24 if self.chroma else 0) + (17 if self.rich else 0)

    @property
    def total_feature_len(self) -> int:       # TOTAL_FEATURE_LEN, pipeline1.py:86
        return self.audio_feature_len + TEXT_FEATURE_LEN

    def n_frames(self, n_samples: int) -> int:
        return 1 + n_samples // self.hop

    def to_c(self) -> "_lib.DysfParams":
        c = _lib.DysfParams()
        c.sr, c.n_fft, c.hop, c.n_mels, c.n_mfcc = self.sr, self.n_fft, self.hop, self.n_mels, self.n_mfcc
        c.fmin = float(self.fmin)
        c.fmax = 0.0 if self.fmax is None else float(self.fmax)
        c.top_db = float(self.top_db)
        c.amin = float(self.amin)
        c.delta_width = self.delta_width
        c.chroma = 1 if self.chroma else 0
        c.rich = 1 if self.rich else 0
        c.parity = 1 if self.parity else 0
        return c


REFERENCE_149 = FeatureParams()
BENCH_512 = FeatureParams(n_fft=512, hop=160, n_mels=64, n_mfcc=40, chroma=False, parity=False)
# main.py:621-630, 718-763 (main.py:628 evaluates to 281 audio + 5 text values, not the 329 of its comment)
MAIN_286 = FeatureParams(n_fft=512, hop=256, n_mfcc=40, chroma=True, rich=True)
