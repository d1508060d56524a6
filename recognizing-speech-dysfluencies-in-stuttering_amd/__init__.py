"""rsd_amd -- MI355X-native drop-in for the feature-extraction + classify path of
kishormb/Recognizing-Speech-Dysfluencies-in-Stuttering (pipeline1.py:206-265, 429-440, 470-514).

Python host code over a thin C-ABI HIP library (``libdysf_hip.so``, ``include/dysf_hip.h``).  Import as
``import rsd_amd`` (shim at the repo root).  torch is imported lazily: only the classifier heads and the
device-resident entry points need it."""
from . import _lib
from ._lib import DysfError
from .params import BENCH_512, MAIN_286, REFERENCE_149, TARGET_SR, FeatureParams
from .features import (CACHE_DIR, FeatureExtractor, HostTables, cached_extract_features, cached_path,
                       extract_audio_features, extract_features, extract_text_features, get_extractor,
                       load_audio, load_cache, repetition_stats_from_text, save_cache)
from .metrics import audio_metrics_batch, high_freq_energy_ratio, snr_db, spectral_flatness_mean
from .cleaning import CLEAR_DIR, AudioCleaner, clean_audio_and_cache, get_cleaner, reduce_noise, write_wav_pcm16
from .preprocess import cached_extract_features_batch, list_audio_files, preprocess_files, write_per_file_analysis
from .predictor import DeployedPredictor, clean_audio_and_cache_main1

__all__ = [
    "DysfError", "FeatureParams", "REFERENCE_149", "BENCH_512", "MAIN_286", "TARGET_SR", "CACHE_DIR", "FeatureExtractor",
    "HostTables", "get_extractor", "load_audio", "save_cache", "load_cache", "cached_path",
    "extract_audio_features", "extract_text_features", "extract_features", "cached_extract_features",
    "repetition_stats_from_text", "snr_db", "spectral_flatness_mean", "high_freq_energy_ratio", "audio_metrics_batch",
    "CLEAR_DIR", "AudioCleaner", "get_cleaner", "reduce_noise", "clean_audio_and_cache", "write_wav_pcm16",
    "list_audio_files", "preprocess_files", "write_per_file_analysis", "cached_extract_features_batch",
    "DeployedPredictor", "clean_audio_and_cache_main1",
]
