"""CPU tier: the oracle against the reference's own known-answer data and against SciPy primitives."""
import json
import os

import numpy as np
import pytest
import scipy.fft
import scipy.signal

from oracle import dysf_oracle as O


def _pairs(golden_dir):
    d = os.path.join(golden_dir, "ref_pairs")
    for e in json.load(open(os.path.join(d, "index.json"))):
        pcm, sr = O.read_wav_pcm16(os.path.join(d, e["id"] + ".wav"))
        yield e, pcm, sr, np.load(os.path.join(d, e["id"] + ".npy"))


def test_validation_record_of_all_888_reference_pairs(golden_dir):
    rec = json.load(open(os.path.join(golden_dir, "validate_888.json")))
    assert rec["n_pairs"] == 888 and rec["n_over_1e-4"] == 0
    assert rec["max_abs_diff"] <= 6.2e-5   # one float32 ulp of a value in [512, 1024): mfcc_mean_0 of three clips
    assert rec["block_max_abs_diff"]["chroma_stats"] <= 1e-6 and rec["block_max_abs_diff"]["text"] == 0.0


def test_oracle_reproduces_committed_reference_vectors(golden_dir):
    n = 0
    for e, pcm, sr, gold in _pairs(golden_dir):
        assert sr == 16000 and len(pcm) == e["n_samples"]
        assert gold.dtype == np.float32 and gold.shape == (149,)
        feats = O.extract_features(O.pcm16_to_float(pcm), sr)
        assert feats.dtype == np.float32
        assert np.abs(feats - gold).max() <= 6.2e-5, e
        assert np.abs(feats[120:144] - gold[120:144]).max() <= 1e-6  # chroma incl. the tuning decision
        assert (feats[144:] == 0).all() and (gold[144:] == 0).all()
        n += 1
    assert n == 17   # 12 shortest + the config-1 clip + the four clips earlier GPU builds missed (make_golden.EDGE_CLIPS)


def test_cabs32_is_numpy_abs_on_an_fma_host():
    """cabs32 spells out NumPy's SIMD complex64 abs (FMA form); on a host with FMA3 / AVX512 the two agree bit for bit."""
    from numpy._core._multiarray_umath import __cpu_features__ as feats
    if not (feats.get("FMA3") or feats.get("AVX512F") or feats.get("ASIMD")):
        pytest.skip("host NumPy uses the unfused form")
    rng = np.random.default_rng(1)
    z = ((rng.standard_normal(200000) + 1j * rng.standard_normal(200000)) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.complex64)
    z[:4] = [0, 1e-30, 3e38 + 3e38j, 1 + 0j]
    assert (np.abs(z) == O.cabs32(z)).all()


def test_log10_f32_is_correctly_rounded():
    x = np.exp(np.random.default_rng(2).uniform(-23, 12, 100000)).astype(np.float32)
    got = O.log10_f32(x)
    exact = np.log10(x.astype(np.float64))
    assert got.dtype == np.float32 and (np.abs(got.astype(np.float64) - exact) <= 0.5000001 * np.spacing(np.abs(got))).all()


def test_oracle_metrics_reproduce_the_reference_csv(golden_dir):
    """snr_db / spectral_flatness_mean / high_freq_energy_ratio (pipeline1.py:151-186) against the reference's own
    per_file_analysis.csv values for the committed clips, and the 888-clip record."""
    rec = json.load(open(os.path.join(golden_dir, "validate_888.json")))["metrics"]
    assert rec["n_clips"] == 888 and rec["snr_db_max_abs_diff"] <= 5e-5 and rec["flatness_max_rel_diff"] <= 5e-4
    assert rec["hf_ratio_max_rel_diff"] <= 1e-6
    for e, pcm, sr, gold in _pairs(golden_dir):
        got = O.audio_metrics(O.pcm16_to_float(pcm), sr)
        want = np.array(e["metrics_after"])
        assert abs(got[0] - want[0]) <= 5e-5 and abs(got[1] - want[1]) <= 5e-4 * want[1] and abs(got[2] - want[2]) <= 1e-6 * want[2]


def test_config1_clip_is_among_the_fixtures(golden_dir):
    idx = json.load(open(os.path.join(golden_dir, "ref_pairs", "index.json")))
    assert any(e["reference_stem"] == "F_0050_10y9m_1 (mp3cut.net)10" and e["n_samples"] == 53951 for e in idx)


@pytest.mark.parametrize("order", [1, 2])
def test_delta_equals_scipy_savgol_interp(order):
    rng = np.random.default_rng(order)
    x = (rng.standard_normal((20, 57)) * 40 - 100).astype(np.float32)
    want = scipy.signal.savgol_filter(x, 9, deriv=order, polyorder=order, axis=-1, mode="interp")
    got = O.delta(x, order)
    assert got.dtype == np.float32 and np.abs(got - want).max() <= 1e-5
    taps = O.delta_taps(order)
    ref = np.arange(-4, 5) / 60.0 if order == 1 else np.array([28, 7, -8, -17, -20, -17, -8, 7, 28]) / 462.0
    assert np.abs(taps - ref).max() < 1e-15
    with pytest.raises(ValueError):
        O.delta(x[:, :8], order)  # fewer than 9 frames: the reference turns this into a zeros row


def test_dct_matrix_equals_scipy_ortho_dct():
    rng = np.random.default_rng(0)
    L = rng.standard_normal((128, 31))
    want = scipy.fft.dct(L, axis=0, type=2, norm="ortho")[:20]
    assert np.abs(O.dct_matrix(128, 20) @ L - want).max() < 1e-12


def test_mel_bank_sparsity_matches_survey():
    for p, nnz in ((O.BENCH_512, 498), (O.REFERENCE_149, 2020)):
        W = O.mel_filterbank(p)
        assert W.dtype == np.float32 and W.shape == (p.n_mels, p.n_bins)
        assert int((W != 0).sum()) == nnz and (W >= 0).all() and (W.sum(axis=1) > 0).all()


def test_tuning_grid_and_histogram_rule():
    g = O.tuning_grid()
    assert len(g) == 100 and g[0] == -0.5 and g[50] == 0.0
    r = O.pitch_residuals(np.array([440.0, 27.5 * 2 ** (0.4999 / 12), 27.5 * 2 ** (0.5 / 12)], np.float32))
    assert r.dtype == np.float32 and abs(r[0]) < 1e-6 and (r >= -0.5).all() and (r < 0.5).all()


def test_error_convention_zero_rows():
    ok = O.pcm16_to_float((np.sin(np.arange(9000) * 0.05) * 9000).astype(np.int16))
    assert np.abs(O.extract_audio_features(ok, 16000)).max() > 0
    assert (O.extract_audio_features(None, 16000) == 0).all()
    assert (O.extract_audio_features(ok[:4000], 16000) == 0).all()        # 8 frames < 9
    bad = ok.copy(); bad[7] = np.nan
    assert (O.extract_audio_features(bad, 16000) == 0).all()
    assert (O.extract_audio_features(np.zeros(0, np.float32), 16000) == 0).all()
    f = O.extract_features(ok, 16000)
    assert f.shape == (149,) and (f[144:] == 0).all()


def test_all_zero_clip_spec():
    f = O.extract_audio_features(np.zeros(20000, np.float32), 16000)
    assert abs(f[0] + 100 * np.sqrt(128.0)) <= 2e-4 and np.abs(f[1:]).max() <= 1e-6


def test_bench_preset_shapes():
    y = O.pcm16_to_float((np.random.default_rng(1).standard_normal(48000) * 3000).astype(np.int16))
    fr = O.frame_level_stack(y, O.BENCH_512)
    assert fr.shape == (120, 301) and fr.dtype == np.float32 and np.isfinite(fr).all()


def test_svc_restatement_reproduces_sklearn(golden_dir):
    """The libsvm restatement (decision values, one-vs-one votes, Platt + pairwise coupling) against scikit-learn's own
    outputs for the re-fitted SVC of pipeline1.py:498 on all 905 reference rows."""
    import json
    z = np.load(os.path.join(golden_dir, "svc_after.npz"), allow_pickle=False)
    ev = json.loads(str(z["evidence"]))
    assert ev["confusion_matches_reference"] and abs(ev["log_loss"] - ev["reference_metrics_summary_after_SVM"][1]) < 1e-9
    c = np.load(os.path.join(golden_dir, "classifier_after.npz"), allow_pickle=False)
    xs = O.scale_rows_like_sklearn(c["X"], z["mean"], z["scale"])  # the 905 cached 149-d rows, file-walk order
    dec = O.svc_decision_ovo(xs, z["support_vectors"], z["dual_coef"], z["intercept"], z["n_support"], float(z["gamma"]))
    assert np.abs(dec - z["expected_decision"]).max() < 1e-9
    assert (O.svc_predict(dec, 3) == z["expected_pred"]).all()
    assert np.abs(O.svc_predict_proba(dec, z["prob_a"], z["prob_b"], 3) - z["expected_proba"]).max() < 1e-9


# ---------------------------------------------------------------------------------------------
# Cleaning stage (pipeline1.py:126-146): oracle/dysf_clean_oracle.py
from oracle import dysf_clean_oracle as CO  # noqa: E402


def test_reference_clean_wavs_pin_normalise_and_pcm16(golden_dir):
    """librosa.util.normalize + sf.write (p1:140-141) leave every clear_audio/*.wav with a peak of exactly +32767 or
    -32768 (scale 32768, clipped): the committed record of all 892 files, the committed pairs, and the oracle's rule."""
    rec = json.load(open(os.path.join(golden_dir, "clean_pin.json")))
    assert rec["files"] == 892 and rec["other_peaks"] == [] and rec["peak_pos_32767"] + rec["peak_neg_32768"] == 892
    assert rec["peak_neg_32768"] > 0 and rec["peak_pos_32767"] > 0   # a symmetric 32767 scale could never give -32768
    assert rec["header_44_bytes_pcm16_mono_16k"] == 892
    for e, pcm, sr, _ in _pairs(golden_dir):
        assert (pcm.max() == 32767) != (pcm.min() == -32768), e
    rng = np.random.default_rng(3)
    for sign in (1.0, -1.0):
        y = rng.standard_normal(4000).astype(np.float32) * np.float32(0.01)
        y[1234] = np.float32(sign * 0.37)
        q = CO.float_to_pcm16(CO.normalize(y))
        assert (q.max(), q.min())[sign < 0] == (32767 if sign > 0 else -32768)
        assert np.abs(q.astype(np.int32)).max() == (32767 if sign > 0 else 32768)
    # round half to even, clipping on both sides
    got = CO.float_to_pcm16(np.array([0.5 / 32768, 1.5 / 32768, -0.5 / 32768, -2.5 / 32768, 0.99999, 1.0, -1.0, 1.5, -1.5], np.float32))
    assert got.tolist() == [0, 2, 0, -2, 32767, 32767, -32768, 32767, -32768]
    with pytest.raises(ValueError):
        CO.normalize(np.array([0.0, np.nan], np.float32))
    assert (CO.normalize(np.zeros(8, np.float32)) == 0).all()    # a peak below tiny divides by 1


def _explicit_gate(y, sr, prop=1.0):
    """The gate written out step by step (what the HIP kernels do), against the SciPy calls the oracle makes."""
    n, P, nfft, hop = len(y), CO.PADDING, CO.N_FFT, CO.N_FFT // 4
    chunk = np.zeros(n + 2 * P)
    chunk[P:P + n] = y
    ext = np.concatenate([np.zeros(nfft // 2), chunk, np.zeros(nfft // 2)])
    T = (len(ext) - nfft) // hop + 1
    win = scipy.signal.get_window("hann", nfft)
    X = np.fft.rfft(np.stack([ext[t * hop:t * hop + nfft] for t in range(T)]) * win, axis=1) / win.sum()
    A = np.abs(X)
    b = CO.time_smoothing_coefficient(sr, hop)
    yf = np.empty_like(A)
    yf[0] = A[0]
    for t in range(1, T):
        yf[t] = b * A[t] + (1 - b) * yf[t - 1]
    yb = np.empty_like(A)
    yb[T - 1] = yf[T - 1]
    for t in range(T - 2, -1, -1):
        yb[t] = b * yf[t] + (1 - b) * yb[t + 1]
    m = 1 / (1 + np.exp(-((A - yb) / yb - 2) * 10))
    ngf, ngt = CO.mask_smoothing_sizes(sr, nfft, hop)
    filt = CO.smoothing_filter(ngf, ngt)          # [freq taps, time taps]
    mp = np.pad(m, ((ngt, ngt), (ngf, ngf)))
    ms = np.zeros_like(m)
    for dt in range(filt.shape[1]):
        for df in range(filt.shape[0]):
            ms += filt[df, dt] * mp[dt:dt + T, df:df + m.shape[1]]
    Xd = X * (ms * prop + (1 - prop))
    xs = np.fft.irfft(Xd, n=nfft, axis=1) * win.sum()
    x = np.zeros(nfft + (T - 1) * hop)
    norm = np.zeros_like(x)
    for t in range(T):
        x[t * hop:t * hop + nfft] += xs[t] * win
        norm[t * hop:t * hop + nfft] += win ** 2
    x /= np.where(norm > 1e-10, norm, 1.0)
    x = x[nfft // 2:-(nfft // 2)]
    o = np.zeros(len(chunk))
    o[:len(x)] = x[:len(chunk)]
    return o[P:P + n].astype(y.dtype)


@pytest.mark.parametrize("prop", [1.0, 0.8])
def test_clean_oracle_equals_the_written_out_gate(prop):
    rng = np.random.default_rng(11)
    t = np.arange(9000) / 16000.0
    y = (0.3 * np.sin(2 * np.pi * 220 * t) * (np.sin(2 * np.pi * 3 * t) > 0) + 0.02 * rng.standard_normal(len(t))).astype(np.float32)
    ref = CO.reduce_noise(y, 16000, prop)
    exp = _explicit_gate(y, 16000, prop)
    assert ref.dtype == np.float32 and ref.shape == y.shape
    assert np.abs(ref.astype(np.float64) - exp).max() <= 2e-7 * np.abs(ref).max()
    assert (CO.float_to_pcm16(CO.normalize(ref)) != CO.float_to_pcm16(CO.normalize(exp))).sum() <= 2
    assert np.sum(ref.astype(np.float64) ** 2) < 0.9 * np.sum(y.astype(np.float64) ** 2)   # the gate removes energy
    sizes = CO.mask_smoothing_sizes(16000, 1024, 256)
    assert sizes == (16, 3) and CO.smoothing_filter(*sizes).shape == (33, 7)
    assert abs(CO.time_smoothing_coefficient(16000, 256) - 0.00796803) < 1e-7


def test_write_wav_pcm16_reproduces_the_reference_files_byte_for_byte(golden_dir, tmp_path):
    """sf.write's PCM_16 container (p1:141): our writer gives back the committed reference WAVs bit for bit."""
    from rsd_amd.cleaning import write_wav_pcm16
    n = 0
    for e, pcm, sr, _ in _pairs(golden_dir):
        out = tmp_path / "w.wav"
        write_wav_pcm16(str(out), pcm, sr)
        assert out.read_bytes() == open(os.path.join(golden_dir, "ref_pairs", e["id"] + ".wav"), "rb").read(), e
        n += 1
    assert n == 17


def test_deployed_models_fixture_reproduces_the_reference_cv_table(golden_dir):
    """tests/golden/deployed_models.npz (make_golden.py --deployed): the 5-fold CV of main1.py:874-893 re-run with the models
    of m1:859-868 reproduces output_results/FINAL_PERFORMANCE_TABLE.csv exactly for RandomForest and SVM (deterministic
    fits) and within one clip in 905 for the MLP and the ensemble that contains it (BLAS-order-sensitive training)."""
    z = np.load(os.path.join(golden_dir, "deployed_models.npz"))
    ev = json.loads(str(z["evidence"]))
    d = ev["max_abs_diff_vs_reference"]
    assert d["RandomForest"] < 1e-9 and d["SVM"] < 1e-9
    assert d["MLP"] < 0.6 and d["Ensemble"] < 1.0
    assert ev["reference_FINAL_PERFORMANCE_TABLE"]["RandomForest"][0] == pytest.approx(63.20441988950276)
    assert z["X"].shape == (905, 149) and len(z["rf_root"]) == 600
    assert [w.shape for w in (z["mlp_W1"], z["mlp_W2"], z["mlp_W3"], z["mlp_W4"])] == [(149, 256), (256, 128), (128, 64), (64, 3)]
    assert z["rf_proba"].shape == (905, 3) and np.array_equal(z["rf_proba"].argmax(1), z["rf_pred"])
    # the forest was fitted on all rows: it reproduces every training label except where one vector carries two labels
    # (17 stems sit in two class folders and share one cache entry: SURVEY 8a9)
    assert (z["rf_pred"] == z["y"]).sum() >= 905 - 17
    tun = json.load(open(os.path.join(golden_dir, "tuning_888.json")))["tuning_index"]
    assert len(tun) == 888 and all(0 <= v < 100 for v in tun.values())


def test_clean_oracle_chunk_loop_follows_noisereduce(monkeypatch):
    """oracle/dysf_clean_oracle.reduce_noise beyond chunk_size (SpectralGate.get_traces): every chunk is the full span
    [k c, (k + 1) c) gated inside `padding` real neighbouring samples -- so a chunk's output depends on nothing outside
    [k c - padding, (k + 1) c + padding) -- and clips up to chunk_size take the single-chunk path unchanged.  Run with a
    small chunk_size / padding so that it is fast."""
    monkeypatch.setattr(CO, "CHUNK_SIZE", 6000)
    monkeypatch.setattr(CO, "PADDING", 1500)
    rng = np.random.default_rng(3)
    y = (0.1 * rng.standard_normal(6000 * 2 + 2500)).astype(np.float32)
    out = CO.reduce_noise(y, 16000, 0.8)
    assert out.shape == y.shape and out.dtype == np.float32 and np.isfinite(out).all()
    # chunk 0 does not see samples at or beyond 6000 + 1500
    y2 = y.copy(); y2[7500:] = 0
    assert np.array_equal(CO.reduce_noise(y2, 16000, 0.8)[:6000], out[:6000])
    # chunk 2 (the last, partial one) does not see samples before 12000 - 1500
    y3 = y.copy(); y3[:10500] = 0
    assert np.array_equal(CO.reduce_noise(y3, 16000, 0.8)[12000:], out[12000:])
    # ... and is the crop of the full-span chunk read past the end as zeros
    chunk = np.zeros(6000 + 2 * 1500); chunk[:1500 + 2500] = y[10500:]
    want = CO.spectral_gate_nonstationary(chunk, 16000, 0.8)[1500:1500 + 2500].astype(np.float32)
    assert np.array_equal(out[12000:], want)
    # exactly chunk_size: one chunk between zeros
    y1 = y[:6000]
    c1 = np.zeros(6000 + 3000); c1[1500:7500] = y1
    assert np.array_equal(CO.reduce_noise(y1, 16000, 0.8), CO.spectral_gate_nonstationary(c1, 16000, 0.8)[1500:7500].astype(np.float32))
