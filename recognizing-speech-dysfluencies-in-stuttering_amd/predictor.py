"""The reference's serve path (main1.py:949-990, "Safe Real-time Prediction") as one object:

    cleaned = clean_audio_and_cache(tmp_path)             # m1:970, main1's variant m1:598-611: prop_decrease=0.8,
                                                          #         a failed cleaning returns the input path itself
    y, sr = load_audio(cleaned or tmp_path)               # m1:971
    feats = extract_features(y, sr, "")                   # m1:975
    if feats.shape[0] != scaler.n_features_in_: error     # m1:976-981
    feats_s = scaler.transform(feats.reshape(1, -1))      # m1:987   } inside the head: the heads take unscaled rows
    pred, prob = rf.predict(...), rf.predict_proba(...)   # m1:988-989
    pred_label = le.inverse_transform([pred])[0]          # m1:990

Every numeric step runs on the GPU behind the C ABI (cleaning: dysf_clean_batch_dev, features: dysf_extract_batch,
forest: dysf_forest_predict_dev).  Inputs must be decodable by ``load_audio`` (mono PCM_16 WAV)."""
import logging
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from .cleaning import CLEAR_DIR, get_cleaner, write_wav_pcm16
from .features import cached_path, extract_features, load_audio
from .params import TARGET_SR


def clean_audio_and_cache_main1(in_path: str, clear_dir: str = CLEAR_DIR, device: int = 0) -> Optional[str]:
    """main1.py:598-611 (main.py:649-663 is the same): ``prop_decrease=0.8``; returns the existing cleaned file, None when
    the input cannot be loaded, and the INPUT path when cleaning itself fails (pipeline1.py's version returns None there)."""
    out_path = cached_path(os.path.join(clear_dir, os.path.splitext(os.path.basename(in_path))[0] + ".wav"))
    if os.path.exists(out_path):
        return out_path
    y, sr = load_audio(in_path)
    if y is None:
        return None
    try:
        res = get_cleaner(sr, device, 0.8).clean_batch([y], pcm16=True)
        if int(res["status"][0]) != 0:
            raise ValueError("Input must be finite")       # librosa.util.normalize's ParameterError
        os.makedirs(clear_dir, exist_ok=True)
        write_wav_pcm16(out_path, res["pcm16"][0], sr)
        return out_path
    except Exception as e:
        logging.error(f"clean_audio {in_path}: {e}")
        return in_path


class DeployedPredictor:
    """``head``: a fitted-model head that takes UNscaled float32 rows (``RandomForestHead`` for model_rf.pkl + scaler_after.pkl,
    or ``FusedMLPHead`` / ``SVCHead`` / ``SoftVotingHead``); ``classes``: ``LabelEncoder.classes_``."""

    def __init__(self, head, classes: Sequence[str], clear_dir: str = CLEAR_DIR, device: int = 0):
        self.head, self.classes, self.clear_dir, self.device = head, [str(c) for c in classes], clear_dir, device

    def predict_file(self, path: str) -> Tuple[Optional[str], Optional[Dict[str, float]]]:
        """(label, {class: probability}) or (None, None) with a logged error, like the sidebar's error branches."""
        import torch
        try:
            cleaned = clean_audio_and_cache_main1(path, self.clear_dir, self.device)
            y, sr = load_audio(cleaned or path)
            if y is None:
                logging.error("Failed to load audio.")
                return None, None
            feats = extract_features(y, sr, "", device=self.device)
            expected = int(self.head.n_features_in_)
            if feats.shape[0] != expected:
                logging.error(f"Feature mismatch! Expected {expected}, got {feats.shape[0]}. Re-train the model with the current feature set.")
                return None, None
            x = torch.from_numpy(np.ascontiguousarray(feats.reshape(1, -1))).to(f"cuda:{self.device}")
            pred = int(self.head.predict(x)[0])
            prob = self.head.predict_proba(x)[0].detach().cpu().numpy()
            return self.classes[pred], {c: float(p) for c, p in zip(self.classes, prob)}
        except Exception as e:          # m1:997-998
            logging.error(f"Prediction error: {e}")
            return None, None
