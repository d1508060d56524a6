"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs, vs the committed
reference fixtures, and size-independent properties at the benchmark's full batch size.

Tolerances (stated here, used below):
  * pooled 149-d / 240-d vectors: 1e-4 absolute (BASELINE.json north_star), checked against the reference's own
    cached vectors and against the oracle;
  * power spectrum: bit-exact in the parity presets (float64 FFT rounded to complex64, NumPy's complex abs); 2e-6 of
    the frame's largest bin in the float32-FFT presets (the reference STFT is float64-accurate);
  * log-mel above the top_db floor and frame-level MFCC/delta rows: 1e-4 + 2e-6*|x| + a floor-noise term (see
    _frame_tol): MFCC row 0 is ~ -500, where one float32 ulp is already 6e-5 and both sides carry DCT rounding;
  * integer outputs (status, tuning index) exact.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import rsd_amd  # noqa: E402
from oracle import dysf_oracle as O  # noqa: E402
from rsd_amd.synth import synth_clip  # noqa: E402

P149, B512 = rsd_amd.REFERENCE_149, rsd_amd.BENCH_512
OP = {P149: O.REFERENCE_149, B512: O.BENCH_512}


@pytest.fixture(scope="module")
def ex149():
    return rsd_amd.get_extractor(P149, 0)


@pytest.fixture(scope="module")
def ex512():
    return rsd_amd.get_extractor(B512, 0)


def _pairs(golden_dir):
    d = os.path.join(golden_dir, "ref_pairs")
    for e in json.load(open(os.path.join(d, "index.json"))):
        pcm, sr = O.read_wav_pcm16(os.path.join(d, e["id"] + ".wav"))
        yield e["id"], pcm, sr, np.load(os.path.join(d, e["id"] + ".npy"))


def test_native_library_is_loaded():
    assert rsd_amd._lib.lib().dysf_abi_version() == rsd_amd._lib.ABI_VERSION
    assert torch.cuda.is_available()


@pytest.mark.parametrize("params", [B512, P149], ids=["fft512", "fft2048"])
@pytest.mark.parametrize("clip", [0, 5, 127])
def test_stft_power_vs_oracle(params, clip):
    ex = rsd_amd.get_extractor(params, 0)
    y = synth_clip(clip, n=24000)
    S = ex.stft_power(y)
    ref = O.stft_power(O.pcm16_to_float(y), OP[params]).T
    assert S.shape == ref.shape
    scale = ref.max(axis=1, keepdims=True) + 1e-30
    assert np.abs(S - ref).max() <= 2e-6 * scale.max()
    assert (np.abs(S - ref) / scale).max() <= 2e-6
    if params.parity:
        # two float64 FFTs of different factorisation agree to ~1e-16 of the frame's largest value, so the complex64
        # roundings differ only for weak bins: every bin within 60 dB of its frame's peak must match bit for bit
        strong = ref >= 1e-6 * scale
        assert np.array_equal(S[strong], ref[strong]), int((S[strong] != ref[strong]).sum())
        assert (np.abs(S - ref) <= 1e-9 * scale).all()   # weaker bins: float64 round-off of two different FFTs


@pytest.mark.parametrize("params", [B512, P149], ids=["fft512", "fft2048"])
def test_logmel_vs_oracle(params):
    ex = rsd_amd.get_extractor(params, 0)
    y = synth_clip(7, n=32000)
    L = ex.logmel(y)
    op = OP[params]
    S = O.stft_power(O.pcm16_to_float(y), op)
    ref = (np.float32(10.0) * O.log10_f32(np.maximum(np.float32(op.amin), O.mel_filterbank(op) @ S))).T
    assert L.shape == ref.shape
    live = ref > ref.max() - 80.0  # cells the top_db clamp keeps
    assert np.abs(L - ref)[live].max() <= 2e-3  # float32 FFT noise grows towards the -80 dB floor
    near = ref > ref.max() - 50.0
    assert np.abs(L - ref)[near].max() <= 1e-4


def _frame_tol(ref):
    return 2e-4 + 2e-6 * np.abs(ref)


@pytest.mark.parametrize("n", [48000, 16000, 1600, 51234])
def test_frames_vs_oracle_bench_preset(ex512, n):
    y = synth_clip(11, n=n)
    res = ex512.extract_batch([y], frames=True)
    ref = O.frame_level_stack(O.pcm16_to_float(y), O.BENCH_512)
    got = res["frames"][0]
    assert res["status"][0] == 0 and got.shape == ref.shape
    assert (np.abs(got - ref) <= _frame_tol(ref)).all(), float(np.abs(got - ref).max())
    pooled_ref = np.hstack([O.stat_pair(ref[:40]), O.stat_pair(ref[40:80]), O.stat_pair(ref[80:])])
    assert np.abs(res["pooled"][0] - pooled_ref).max() <= 1e-4


def test_reference_pairs_pooled_149(golden_dir):
    """The reference's own known-answer vectors: clear_audio/<stem>.wav -> cache_features/<stem>_clean_feats.npy."""
    worst = 0.0
    for cid, pcm, sr, gold in _pairs(golden_dir):
        y = O.pcm16_to_float(pcm)
        feats = rsd_amd.extract_features(y, sr)
        assert feats.dtype == np.float32 and feats.shape == (149,)
        d = np.abs(feats - gold)
        worst = max(worst, float(d.max()))
        assert d[:120].max() <= 1e-4, (cid, float(d[:120].max()), int(d.argmax()))
        assert d[120:144].max() <= 1e-4, (cid, "chroma", float(d[120:144].max()))
        assert (feats[144:] == 0).all()
    print("worst |gpu - reference| over committed pairs:", worst)


def test_tuning_index_matches_oracle(ex149, golden_dir):
    clips, want = [], []
    for cid, pcm, sr, gold in _pairs(golden_dir):
        clips.append(pcm)
        _, dbg = O.estimate_tuning(O.stft_power(O.pcm16_to_float(pcm), O.REFERENCE_149), O.REFERENCE_149)
        want.append(dbg["index"])
    for i in range(6):
        y = synth_clip(i, n=30000)
        clips.append(y)
        want.append(O.estimate_tuning(O.stft_power(O.pcm16_to_float(y), O.REFERENCE_149), O.REFERENCE_149)[1]["index"])
    res = ex149.extract_batch(clips, tuning=True)
    assert res["tuning"].tolist() == want


def test_parity_statistics_follow_numpy_summation_order(ex149):
    """Parity presets: the pooled mean / std are NumPy's float32 pairwise sums of the frame rows (pipeline1.py:220-221),
    bit for bit, at lengths on both sides of the 128-element block and of the 8-wide unroll."""
    for n in (4608, 7168, 48000, 65536 + 300, 66048, 131072 + 5 * 512, 160000):
        y = synth_clip(3, n=n)
        res = ex149.extract_batch([y], frames=True)
        fr = res["frames"][0]
        want = np.hstack([O.stat_pair(fr[:20]), O.stat_pair(fr[20:40]), O.stat_pair(fr[40:])])
        assert np.array_equal(res["pooled"][0][:120], want), (n, float(np.abs(res["pooled"][0][:120] - want).max()))
        only = ex149.extract_batch([y])  # without a frames output the rows go through the workspace stash
        assert np.array_equal(only["pooled"], res["pooled"])


def test_clip_longer_than_max_samples_is_rejected(ex512):
    """Device entry point with an under-reported max_samples: status TOO_LONG and a zero row, never a write past the
    per-clip workspace rows."""
    clips = [synth_clip(1, n=16000), synth_clip(2, n=48000), synth_clip(3, n=16000)]
    pcm = torch.from_numpy(np.concatenate(clips)).cuda()
    off = torch.tensor([0, 16000, 64000, 80000], dtype=torch.int64, device="cuda")
    res = ex512.extract_batch_device(pcm, off, 16000, pooled=True, frames=True)
    torch.cuda.synchronize()
    assert res["status"].tolist() == [0, 4, 0]
    assert float(res["pooled"][1].abs().max()) == 0.0 and float(res["frames"][1].abs().max()) == 0.0
    ok = ex512.extract_batch([clips[0], clips[2]], frames=True)
    assert np.array_equal(res["pooled"][[0, 2]].cpu().numpy(), ok["pooled"])


def test_more_clips_than_one_grid_dimension(ex512):
    """One call with more than 65535 clips (gridDim.y of the STFT kernel): split into sub-batches internally."""
    base = [synth_clip(i, n=1600) for i in range(5)]
    n = 66000
    pcm = torch.from_numpy(np.concatenate(base)).cuda().repeat(n // 5)
    off = torch.arange(0, n + 1, dtype=torch.int64, device="cuda") * 1600
    res = ex512.extract_batch_device(pcm, off, 1600, pooled=True)
    torch.cuda.synchronize()
    assert int(res["status"].abs().sum()) == 0
    po = res["pooled"].view(n // 5, 5, -1)
    assert bool((po == po[:1]).all())
    want = ex512.extract_batch(base)["pooled"]
    assert np.array_equal(po[-1].cpu().numpy(), want)


def test_deployed_predictor_wav_to_label_string(golden_dir):
    """The reference's serve path, main1.py:965-990, as one call chain: WAV file -> load_audio -> extract_features ->
    guard on n_features_in_ -> scaler.transform -> rf.predict / predict_proba -> LabelEncoder.inverse_transform.
    Expected label = scikit-learn's own prediction (the fitted forest of pipeline1.py:496, tests/golden/forest_after.npz)
    on the reference's CACHED vector of the same clip."""
    from rsd_amd.classifier import RandomForestHead
    c = np.load(os.path.join(golden_dir, "classifier_after.npz"), allow_pickle=False)
    f = np.load(os.path.join(golden_dir, "forest_after.npz"), allow_pickle=False)
    classes = [str(x) for x in c["classes"]]                      # LabelEncoder.classes_ (sorted label strings)
    rf = RandomForestHead.from_npz(os.path.join(golden_dir, "forest_after.npz"), classes=classes)
    d = os.path.join(golden_dir, "ref_pairs")
    n = 0
    for e in json.load(open(os.path.join(d, "index.json"))):
        gold = np.load(os.path.join(d, e["id"] + ".npy"))
        rows = np.nonzero((c["X"] == gold[None, :]).all(axis=1))[0]   # this clip's row(s) in the 905 x 149 training matrix
        assert len(rows) >= 1, e
        want_label = classes[int(f["expected_pred"][rows[0]])]
        y, sr = rsd_amd.load_audio(os.path.join(d, e["id"] + ".wav"))          # main1.py:971
        feats = rsd_amd.extract_features(y, sr, "")                             # main1.py:975
        assert feats.shape[0] == rf.n_features_in_                              # main1.py:976-981
        x = torch.from_numpy(feats.reshape(1, -1)).cuda()
        pred = int(rf.predict(x)[0])                                            # main1.py:987-988 (scaler inside the head)
        proba = rf.predict_proba(x)[0].cpu().numpy()                            # main1.py:989
        assert classes[pred] == want_label, (e["id"], classes[pred], want_label)  # main1.py:990
        assert np.abs(proba - f["expected_proba"][rows[0]]).max() <= 0.0051      # at most one of 200 trees may differ
        n += 1
    assert n == 17
    rf.close()


def test_audio_metrics_match_reference_csv_and_oracle(golden_dir):
    """pipeline1.py:151-186 on the GPU against (i) the reference's own per_file_analysis.csv values for the committed clips
    (snr_after, flat_after, hf_after) and (ii) the oracle on synthetic clips of awkward lengths (prime, shorter than one
    25 ms frame, silent).  Tolerances: snr 2e-4 dB, flatness 5e-4 relative (float32 log / exp of two libraries),
    high-frequency ratio 2e-5 relative (float32 FFT of a different factorisation)."""
    idx = json.load(open(os.path.join(golden_dir, "ref_pairs", "index.json")))
    clips, want = [], []
    for (cid, pcm, sr, gold), e in zip(_pairs(golden_dir), idx):
        assert e["metrics_after"] is not None
        clips.append(pcm); want.append(e["metrics_after"])
    got = rsd_amd.audio_metrics_batch(clips, 16000)
    want = np.array(want)
    assert np.abs(got[:, 0] - want[:, 0]).max() <= 2e-4, np.abs(got[:, 0] - want[:, 0]).max()
    assert (np.abs(got[:, 1] - want[:, 1]) / want[:, 1]).max() <= 5e-4
    assert (np.abs(got[:, 2] - want[:, 2]) / want[:, 2]).max() <= 2e-5
    # the reference-shaped single-clip functions
    y = O.pcm16_to_float(clips[3])
    assert abs(rsd_amd.snr_db(y) - want[3, 0]) <= 2e-4 and abs(rsd_amd.high_freq_energy_ratio(y, 16000) - want[3, 2]) <= 2e-5 * want[3, 2]
    assert rsd_amd.snr_db(None) == 0.0
    odd = [synth_clip(5, n=n) for n in (7919, 48000, 30011, 300, 4099, 160001)] + [np.zeros(9000, np.int16)]   # 160001: a 2^19 chirp-z transform
    got = rsd_amd.audio_metrics_batch(odd, 16000)
    for i, c in enumerate(odd):
        w = O.audio_metrics(O.pcm16_to_float(c), 16000)
        assert abs(got[i, 0] - w[0]) <= 2e-4, (i, got[i], w)
        assert abs(got[i, 1] - w[1]) <= 5e-4 * max(w[1], 1e-6), (i, got[i], w)
        assert abs(got[i, 2] - w[2]) <= 2e-5 * max(w[2], 1e-6), (i, got[i], w)


def test_int16_and_float32_inputs_agree(ex149, ex512):
    y = synth_clip(2, n=20000)
    for ex in (ex149, ex512):
        a = ex.extract_batch([y], frames=True)
        b = ex.extract_batch([O.pcm16_to_float(y)], frames=True)
        assert np.array_equal(a["pooled"], b["pooled"]) and np.array_equal(a["frames"], b["frames"])


def test_error_convention_zero_rows(ex149):
    ok = synth_clip(1, n=9000)
    short = synth_clip(1, n=4000)            # T = 8 < 9 -> delta raises in the reference -> zeros
    nan = O.pcm16_to_float(ok).copy(); nan[1234] = np.nan
    inf = O.pcm16_to_float(ok).copy(); inf[-1] = np.inf
    res = ex149.extract_batch([O.pcm16_to_float(ok), O.pcm16_to_float(short), nan, inf, np.zeros(0, np.float32)], frames=True)
    assert res["status"].tolist() == [0, 1, 2, 2, 3]
    assert np.abs(res["pooled"][0]).max() > 0
    for i in (1, 2, 3, 4):
        assert (res["pooled"][i] == 0).all() and (res["frames"][i] == 0).all()
    assert (rsd_amd.extract_features(None, None) == 0).all()
    assert (rsd_amd.extract_features(nan, 16000) == 0).all()
    with pytest.raises(ValueError):
        rsd_amd.extract_audio_features(nan, 16000, strict=True)


def test_all_zero_clip_matches_spec(ex149):
    """L == -100 everywhere, mfcc0 = -100*sqrt(128), every std 0, tuning 0.0, chroma 0."""
    res = ex149.extract_batch([np.zeros(20000, np.int16)], tuning=True)
    f = res["pooled"][0]
    assert res["status"][0] == 0 and res["tuning"][0] == 50
    assert f[0] == np.float32(-100.0 * np.sqrt(128.0))  # one float32 ulp is 1.2e-4 here: demand the exact rounding
    assert np.abs(f[1:]).max() <= 1e-6
    want = O.extract_audio_features(np.zeros(20000, np.float32), 16000)
    # the oracle's (scipy's) float32 DCT lands one ulp (1.2e-4) off the exact -100*sqrt(128); everything else is 0
    assert abs(f[0] - want[0]) <= 1.3e-4 and np.abs(f[1:] - want[1:]).max() <= 1e-6


def test_ragged_batch_equals_single_clip_calls(ex149, ex512):
    lens = [4096, 5000, 48000, 12345, 30000, 9999]
    clips = [synth_clip(20 + i, n=n) for i, n in enumerate(lens)]
    for ex in (ex149, ex512):
        batch = ex.extract_batch(clips, frames=True, tuning=True)
        for i, c in enumerate(clips):
            one = ex.extract_batch([c], frames=True, tuning=True)
            T = ex.params.n_frames(len(c))
            assert np.array_equal(batch["pooled"][i], one["pooled"][0])
            assert np.array_equal(batch["frames"][i][:, :T], one["frames"][0])
            assert (batch["frames"][i][:, T:] == 0).all()
            assert batch["tuning"][i] == one["tuning"][0]


def test_long_clip_uses_global_workspace(ex149, ex512):
    """Clips whose log-mel does not fit in LDS take the global-workspace variant of the kernel; same numbers."""
    y = np.concatenate([synth_clip(40 + i, n=48000) for i in range(4)])  # 12 s
    f = O.pcm16_to_float(y)
    got = ex149.extract_batch([y])["pooled"][0]
    assert np.abs(got - O.extract_audio_features(f, 16000)).max() <= 1e-4
    res = ex512.extract_batch([y], frames=True)
    ref = O.frame_level_stack(f, O.BENCH_512)
    assert (np.abs(res["frames"][0] - ref) <= _frame_tol(ref)).all()


def test_full_batch_properties_bench_config(ex512):
    """BASELINE configs[1] size: 1024 clips x 48000 samples resident in HBM.  Size-independent checks:
    rows of identical clips are bit-identical wherever they sit in the batch, a sample of rows matches the
    oracle, and the pooled row equals the statistics of the frame-level rows."""
    bank = np.stack([synth_clip(i) for i in range(32)])
    idx = np.arange(1024) % 32
    pcm = torch.from_numpy(bank[idx].reshape(-1)).cuda()
    off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
    res = ex512.extract_batch_device(pcm, off, 48000, pooled=True, frames=True)
    torch.cuda.synchronize()
    fr, po = res["frames"], res["pooled"]
    assert fr.shape == (1024, 120, 301) and int(res["status"].abs().sum()) == 0
    base = fr[:32]
    for r in range(1, 32):
        assert torch.equal(fr[r * 32:(r + 1) * 32], base)
        assert torch.equal(po[r * 32:(r + 1) * 32], po[:32])
    for j in (0, 13, 31):
        ref = O.frame_level_stack(O.pcm16_to_float(bank[j]), O.BENCH_512)
        got = fr[992 + j].cpu().numpy()
        assert (np.abs(got - ref) <= _frame_tol(ref)).all()
    f64 = fr.double()
    mean, std = f64.mean(dim=2), f64.var(dim=2, unbiased=False).sqrt()
    want = torch.cat([mean[:, :40], std[:, :40], mean[:, 40:80], std[:, 40:80], mean[:, 80:], std[:, 80:]], dim=1)
    assert float((po.double() - want).abs().max()) <= 1e-4


def test_full_batch_properties_reference_preset(ex149):
    """The reference's own parameterisation at the bench batch size (1024 clips x 3 s resident in HBM, parity mode):
    identical clips give bit-identical rows and tuning indices wherever they sit in the batch, a sample matches the oracle
    (pooled 1e-4, tuning exact), statuses are clean."""
    bank = np.stack([synth_clip(i) for i in range(16)])
    idx = np.arange(1024) % 16
    pcm = torch.from_numpy(bank[idx].reshape(-1)).cuda()
    off = torch.arange(0, 1025, dtype=torch.int64, device="cuda") * 48000
    res = ex149.extract_batch_device(pcm, off, 48000, pooled=True, tuning=True)
    torch.cuda.synchronize()
    po, tu = res["pooled"], res["tuning"]
    assert po.shape == (1024, 144) and int(res["status"].abs().sum()) == 0
    for r in range(1, 64):
        assert torch.equal(po[r * 16:(r + 1) * 16], po[:16]) and torch.equal(tu[r * 16:(r + 1) * 16], tu[:16])
    for j in (0, 7, 15):
        y = O.pcm16_to_float(bank[j])
        want = O.extract_audio_features(y, 16000)
        assert np.abs(po[1008 + j].cpu().numpy() - want).max() <= 1e-4
        assert int(tu[1008 + j]) == O.estimate_tuning(O.stft_power(y, O.REFERENCE_149), O.REFERENCE_149)[1]["index"]


def test_mlp_head_argmax_on_gpu(golden_dir):
    from rsd_amd.classifier import MLPHead
    z = np.load(os.path.join(golden_dir, "classifier_after.npz"))
    head = MLPHead.from_npz(os.path.join(golden_dir, "classifier_after.npz")).cuda()
    x = torch.from_numpy(z["X"]).cuda()
    pred = head.predict(x).cpu().numpy()
    proba = head.predict_proba(x).cpu().numpy()
    assert np.array_equal(pred, z["expected_pred"])
    assert np.abs(proba - z["expected_proba"]).max() <= 1e-4


def test_fused_pipeline_features_to_logits(golden_dir):
    """wav -> HIP features -> scaler + MLP on the same GPU: label equals the oracle-features label."""
    from rsd_amd.classifier import MLPHead
    from rsd_amd.pipeline import DysfluencyPipeline
    head = MLPHead.from_npz(os.path.join(golden_dir, "classifier_after.npz")).cuda()
    pipe = DysfluencyPipeline(rsd_amd.get_extractor(P149, 0), head, "pooled")
    clips, gold = [], []
    for cid, pcm, sr, g in _pairs(golden_dir):
        clips.append(pcm); gold.append(g)
    pcm = torch.from_numpy(np.concatenate(clips)).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum([len(c) for c in clips])]), dtype=torch.int64, device="cuda")
    res = pipe.run_resident(pcm, off, max(len(c) for c in clips))
    torch.cuda.synchronize()
    want = head.predict(torch.from_numpy(np.stack(gold)).cuda())
    assert torch.equal(res["logits"].argmax(dim=1), want)


def test_cnn_head_runs_on_frame_features(ex512):
    from rsd_amd.classifier import CNNHead
    from rsd_amd.pipeline import DysfluencyPipeline
    cnn = CNNHead().cuda()
    pipe = DysfluencyPipeline(ex512, cnn, "frames")
    clips = np.stack([synth_clip(i) for i in range(8)])
    pcm = torch.from_numpy(clips.reshape(-1)).cuda()
    off = torch.arange(0, 9, dtype=torch.int64, device="cuda") * 48000
    res = pipe.run_resident(pcm, off, 48000)
    torch.cuda.synchronize()
    ref_in = torch.from_numpy(np.stack([O.frame_level_stack(O.pcm16_to_float(c), O.BENCH_512) for c in clips]))
    ref = CNNHead()(ref_in)
    assert torch.equal(res["logits"].argmax(1).cpu(), ref.argmax(1))
    assert float((res["logits"].cpu() - ref.detach()).abs().max()) <= 1e-3


def test_config5_variable_length_22k(golden_dir):
    """BASELINE configs[4] shape: clips of 1-10 s at 22.05 kHz (the rate is just a parameter: no resampling),
    n_fft 2048 / hop 512 / 128 mels / 20 MFCC + delta + delta2, ragged batch, one launch."""
    p = rsd_amd.FeatureParams(sr=22050, chroma=True)
    op = O.FeatureParams(sr=22050)
    ex = rsd_amd.get_extractor(p, 0)
    rng = np.random.default_rng(4321)
    lens = (rng.uniform(1.0, 10.0, size=6) * 22050).astype(int)
    clips = [synth_clip(100 + i, n=int(n), sr=22050) for i, n in enumerate(lens)]
    res = ex.extract_batch(clips, frames=True, tuning=True)
    assert res["status"].tolist() == [0] * 6
    for i, c in enumerate(clips):
        y = O.pcm16_to_float(c)
        want = O.extract_audio_features(y, 22050, op)
        assert np.abs(res["pooled"][i][:120] - want[:120]).max() <= 1e-4, i
        S = O.stft_power(y, op)
        tun, dbg = O.estimate_tuning(S, op)
        assert res["tuning"][i] == dbg["index"], (i, res["tuning"][i], dbg["index"])  # parity preset: no flips
        assert np.abs(res["pooled"][i][120:] - want[120:]).max() <= 1e-4, i
        T = op.n_frames(len(c))
        ref = O.frame_level_stack(y, op)
        assert (np.abs(res["frames"][i][:, :T] - ref) <= _frame_tol(ref)).all()
        assert (res["frames"][i][:, T:] == 0).all()


def test_sub_batched_launch_equals_single_launch(monkeypatch):
    """Huge batches (BASELINE configs[3]: 100k clips) run as several sub-batches that reuse one workspace."""
    clips = [synth_clip(i % 7, n=16000 + 800 * (i % 5)) for i in range(40)]
    ref = rsd_amd.FeatureExtractor(B512, 0).extract_batch(clips, frames=True)
    monkeypatch.setenv("DYSF_WS_MB", "1")  # 1 MiB of per-clip workspace -> a handful of clips per sub-batch
    small = rsd_amd.FeatureExtractor(B512, 0)
    got = small.extract_batch(clips, frames=True)
    small.close()
    for k in ("pooled", "frames", "status"):
        assert np.array_equal(ref[k], got[k]), k


def test_sub_batched_launch_parity_preset_and_metrics(monkeypatch):
    """The same for the chroma / parity preset (spectrogram, row maxima, peak lists, frame-row stash per sub-batch) and for
    the metrics launch (flatness rows, frame energies, chirp-z buffers)."""
    clips = [synth_clip(i % 5, n=9000 + 1700 * (i % 4)) for i in range(14)]
    ref_ex = rsd_amd.FeatureExtractor(P149, 0)
    ref = ref_ex.extract_batch(clips, tuning=True)
    ref_m = rsd_amd.audio_metrics_batch(clips, 16000)
    monkeypatch.setenv("DYSF_WS_MB", "2")   # a few clips per sub-batch
    small = rsd_amd.FeatureExtractor(P149, 0)
    got = small.extract_batch(clips, tuning=True)
    for k in ("pooled", "status", "tuning"):
        assert np.array_equal(ref[k], got[k]), k
    rsd_amd.features._extractors.clear()    # audio_metrics_batch caches its extractor: force one that sees DYSF_WS_MB
    got_m = rsd_amd.audio_metrics_batch(clips, 16000)
    rsd_amd.features._extractors.clear()
    assert np.array_equal(ref_m, got_m)
    small.close(); ref_ex.close()


def test_tuning_only_call_equals_full_call(ex149):
    """dysf_extract_batch_dev with only the tuning output requested (no pooled rows): the estimate is still computed."""
    clips = [synth_clip(30 + i, n=20000 + 3000 * i) for i in range(4)]
    pcm = torch.from_numpy(np.concatenate(clips)).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum([len(c) for c in clips])]), dtype=torch.int64, device="cuda")
    full = ex149.extract_batch_device(pcm, off, max(len(c) for c in clips), pooled=True, tuning=True)
    only = ex149.extract_batch_device(pcm, off, max(len(c) for c in clips), pooled=False, tuning=True)
    torch.cuda.synchronize()
    assert torch.equal(full["tuning"], only["tuning"]) and len(set(full["tuning"].tolist())) > 1


def test_cnn_head_fp16_agreement(ex512):
    """configs[4] asks for the CNN in fp16: report agreement with the fp32 head (random weights, so no oracle)."""
    from rsd_amd.classifier import CNNHead
    clips = np.stack([synth_clip(i) for i in range(16)])
    pcm = torch.from_numpy(clips.reshape(-1)).cuda()
    off = torch.arange(0, 17, dtype=torch.int64, device="cuda") * 48000
    fr = ex512.extract_batch_device(pcm, off, 48000, pooled=False, frames=True)["frames"]
    cnn = CNNHead().cuda()
    with torch.no_grad():
        l32 = cnn(fr)
        l16 = cnn.half()(fr.half()).float()
    agree = float((l32.argmax(1) == l16.argmax(1)).float().mean())
    assert torch.isfinite(l16).all() and agree >= 0.8, agree


def test_streaming_ingest_matches_resident_path(ex149, golden_dir, tmp_path):
    """WAV files -> pinned double-buffered H2D -> kernels -> pooled rows; same numbers as the
    resident path, unreadable files become zero rows."""
    from rsd_amd.ingest import StreamingExtractor, pack_wavs
    d = os.path.join(golden_dir, "ref_pairs")
    paths = [os.path.join(d, f"{i:02d}.wav") for i in range(13)] + [str(tmp_path / "missing.wav")]
    flat, offsets, ok = pack_wavs(paths, 16000)
    assert ok == [True] * 13 + [False]
    got = StreamingExtractor(ex149, chunk_clips=4).run(flat, offsets)
    clips = [flat[offsets[i]:offsets[i + 1]] for i in range(14)]
    want = ex149.extract_batch(clips)
    assert np.array_equal(got["pooled"], want["pooled"]) and np.array_equal(got["status"], want["status"])
    assert got["status"][-1] == 3 and (got["pooled"][-1] == 0).all()
    for i in range(13):
        gold = np.load(os.path.join(d, f"{i:02d}.npy"))
        assert np.abs(got["pooled"][i] - gold[:144]).max() <= 1e-4


def test_random_forest_head_matches_sklearn_fixture(golden_dir):
    """the reference's deployed predictor (RandomForest(200) on scaled features, main1.py:987-989)
    as a HIP tree-walk kernel.  Fixture = the forest re-fit exactly as pipeline1.py:496 does (its accuracy, log-loss
    and confusion matrix equal output_results/metrics_summary.csv:5 / confusion_after_RandomForest.csv)."""
    from rsd_amd.classifier import RandomForestHead
    z = np.load(os.path.join(golden_dir, "forest_after.npz"))
    X = np.load(os.path.join(golden_dir, "classifier_after.npz"))["X"]
    head = RandomForestHead.from_npz(os.path.join(golden_dir, "forest_after.npz"))
    x = torch.from_numpy(X).cuda()
    pred = head.predict(x).cpu().numpy()
    proba = head.predict_proba(x).cpu().numpy()
    assert np.array_equal(pred, z["expected_pred"])            # arg-max exact on all 905 rows
    assert np.abs(proba - z["expected_proba"]).max() <= 1e-12   # float64 average of the same leaf fractions
    with pytest.raises(ValueError):
        head.predict(torch.zeros(1, 148).cuda())
    head.close()


def test_svc_head_matches_sklearn_fixture(golden_dir):
    """SVC(probability=True, kernel='rbf', C=1.0) of pipeline1.py:498 as a HIP kernel that follows
    libsvm's prediction algorithm.  Fixture = the SVC re-fit exactly as the reference does (accuracy, log-loss and
    confusion matrix equal metrics_summary.csv / confusion_after_SVM.csv) with scikit-learn's own outputs.
    Tolerances: float64 decision values 1e-9, one-vs-one votes exact, coupled Platt probabilities 1e-9."""
    from rsd_amd.classifier import SVCHead
    z = np.load(os.path.join(golden_dir, "svc_after.npz"))
    X = np.load(os.path.join(golden_dir, "classifier_after.npz"))["X"]
    head = SVCHead.from_npz(os.path.join(golden_dir, "svc_after.npz"))
    x = torch.from_numpy(X).cuda()
    dec = head.decision_function(x).cpu().numpy()
    assert np.abs(dec - z["expected_decision"]).max() <= 1e-9
    assert np.array_equal(head.predict(x).cpu().numpy(), z["expected_pred"])
    assert np.abs(head.predict_proba(x).cpu().numpy() - z["expected_proba"]).max() <= 1e-9
    with pytest.raises(ValueError):
        head.predict(torch.zeros(1, 148).cuda())
    head.close()


def test_soft_voting_of_the_three_heads(golden_dir):
    """VotingClassifier(voting='soft') of main1.py:867 over forest + MLP + SVC: mean of the members' probabilities.
    Expected = the mean of scikit-learn's own per-model probabilities from the three fixtures (MLP within 1e-4)."""
    from rsd_amd.classifier import MLPHead, RandomForestHead, SVCHead, SoftVotingHead
    c = np.load(os.path.join(golden_dir, "classifier_after.npz"))
    f = np.load(os.path.join(golden_dir, "forest_after.npz"))
    s = np.load(os.path.join(golden_dir, "svc_after.npz"))
    want = (f["expected_proba"] + c["expected_proba"] + s["expected_proba"]) / 3.0
    heads = [RandomForestHead.from_npz(os.path.join(golden_dir, "forest_after.npz")),
             MLPHead.from_npz(os.path.join(golden_dir, "classifier_after.npz")).cuda(),
             SVCHead.from_npz(os.path.join(golden_dir, "svc_after.npz"))]
    vote = SoftVotingHead(heads)
    x = torch.from_numpy(c["X"]).cuda()
    got = vote.predict_proba(x).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4
    margin = np.sort(want, axis=1)
    clear = (margin[:, -1] - margin[:, -2]) > 1e-3          # rows whose winner is not within the MLP's float32 noise
    assert np.array_equal(vote.predict(x).cpu().numpy()[clear], want.argmax(1)[clear])
    assert clear.mean() > 0.95


def test_fused_mlp_head_matches_fixture_and_eager_head(golden_dir):
    """The one-kernel MLP head (dysf_mlp_*) against scikit-learn's outputs for the re-fitted MLP of pipeline1.py:497
    (arg-max exact, probabilities 1e-4) and against the eager torch head on random rows (logits 1e-4 relative)."""
    from rsd_amd.classifier import FusedMLPHead, MLPHead
    path = os.path.join(golden_dir, "classifier_after.npz")
    z = np.load(path)
    fused, eager = FusedMLPHead.from_npz(path), MLPHead.from_npz(path).cuda()
    x = torch.from_numpy(z["X"]).cuda()
    assert np.array_equal(fused.predict(x).cpu().numpy(), z["expected_pred"])
    assert np.abs(fused.predict_proba(x).cpu().numpy() - z["expected_proba"]).max() <= 1e-4
    lg, le = fused(x).cpu().numpy(), eager(x).cpu().numpy()
    assert np.abs(lg - le).max() <= 1e-4 * max(1.0, np.abs(le).max())
    for n in (1, 7, 8, 9):  # ragged tails of the 8-row workgroups
        assert np.array_equal(fused.predict(x[:n]).cpu().numpy(), z["expected_pred"][:n])
    with pytest.raises(ValueError):
        fused.predict(torch.zeros(1, 148).cuda())
    fused.close()


def test_workspace_is_clean_after_rejected_clips(ex512):
    """Presets without chroma skip the per-launch workspace memset: the MFCC kernel zeroes the clip maximum and the
    non-finite flag after reading them.  A launch with rejected clips must leave nothing behind for the next one."""
    good = [O.pcm16_to_float(synth_clip(i, n=16000)) for i in range(4)]
    want = ex512.extract_batch(good)
    assert (want["status"] == 0).all()
    bad = [g.copy() for g in good]
    bad[1][100] = np.nan
    bad[2] = bad[2][:300]            # too short
    bad[3] = np.zeros(0, np.float32)  # empty
    res = ex512.extract_batch(bad)
    assert list(res["status"]) == [rsd_amd._lib.DYSF_OK, rsd_amd._lib.DYSF_NONFINITE, rsd_amd._lib.DYSF_TOO_SHORT, rsd_amd._lib.DYSF_EMPTY]
    assert not res["pooled"][1:].any()
    again = ex512.extract_batch(good)
    assert (again["status"] == 0).all()
    assert np.array_equal(again["pooled"], want["pooled"])


def test_resampler_matches_scipy_resample_poly(tmp_path):
    """22.05 kHz -> 16 kHz on the GPU against scipy.signal.resample_poly (float64) on the same samples:
    float32 rounding only (2e-7 of full scale), exact output lengths, ragged batch, int16 and float32 input, and
    load_audio() of a 22.05 kHz WAV."""
    from rsd_amd.resample import Resampler
    rs = Resampler(22050, 16000, 0)
    rng = np.random.default_rng(5)
    lens = [66150, 22050, 1, 0, 4410, 100001]
    clips = [np.clip(rng.standard_normal(n) * 6000, -32768, 32767).astype(np.int16) for n in lens]
    pcm = torch.from_numpy(np.concatenate(clips)).cuda()
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
    out, out_off = rs.resample_batch_device(pcm, off, max_samples=max(lens))
    out, out_off = out.cpu().numpy(), out_off.cpu().numpy()
    for i, c in enumerate(clips):
        want = O.resample_poly_ref(c, 22050, 16000) if len(c) else np.zeros(0, np.float32)
        got = out[out_off[i]:out_off[i + 1]]
        assert len(got) == len(want) == rs.out_len(len(c))
        if len(c):
            assert np.abs(got - want).max() <= 2e-7, (i, float(np.abs(got - want).max()))
    f32 = (clips[0].astype(np.float32) / 32768.0)
    assert np.array_equal(rs.resample(f32), rs.resample(clips[0]))
    # other ratios take the general kernel (taps from L2): many taps per phase (44.1 -> 16 kHz), tiny up (8 -> 16 kHz)
    for sr_in, sr_out in ((44100, 16000), (8000, 16000), (16000, 22050)):
        r2 = Resampler(sr_in, sr_out, 0)
        got = r2.resample(clips[4])
        want = O.resample_poly_ref(clips[4], sr_in, sr_out)
        assert len(got) == len(want) and np.abs(got - want).max() <= 2e-7, (sr_in, sr_out)
        r2.close()
    # a 22.05 kHz PCM_16 WAV through load_audio(sr=16000): same samples, then the feature path accepts them
    import struct
    path = str(tmp_path / "tone22k.wav")
    t = np.arange(22050 * 2) / 22050.0
    y = (0.4 * np.sin(2 * np.pi * 440.0 * t) * 32767).astype(np.int16)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + 2 * len(y)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 22050, 44100, 2, 16))
        f.write(b"data" + struct.pack("<I", 2 * len(y)) + y.tobytes())
    got, sr = rsd_amd.load_audio(path, sr=16000)
    assert sr == 16000 and np.abs(got - O.resample_poly_ref(y, 22050, 16000)).max() <= 2e-7
    feats = rsd_amd.extract_audio_features(got, sr)
    assert np.abs(feats - O.extract_audio_features(got, 16000)).max() <= 1e-4
    rs.close()


@pytest.mark.parametrize("cfg", [
    dict(n_fft=512, hop=256, n_mels=128, n_mfcc=40, chroma=True),    # main.py:621-630, 724-736 (MFCC_N 40, N_FFT 512, HOP 256)
    dict(n_fft=1024, hop=256, n_mels=128, n_mfcc=20, chroma=True),
    dict(n_fft=1024, hop=512, n_mels=64, n_mfcc=13, chroma=False),
    dict(n_fft=256, hop=128, n_mels=40, n_mfcc=13, chroma=False),
    dict(n_fft=256, hop=64, n_mels=64, n_mfcc=20, chroma=True),
    dict(n_fft=2048, hop=512, n_mels=64, n_mfcc=40, chroma=False),
], ids=lambda c: "fft%d_hop%d_mel%d_mfcc%d_%s" % (c["n_fft"], c["hop"], c["n_mels"], c["n_mfcc"], "chroma" if c["chroma"] else "nochroma"))
def test_other_geometries_vs_oracle(cfg):
    """Every FFT size the kernels are instantiated for (256 / 512 / 1024 / 2048 points), with and without the chroma
    pass, including the parameterisation of main.py:718-736 -- pooled vectors 1e-4 against the oracle, tuning exact."""
    gp, op = rsd_amd.FeatureParams(**cfg), O.FeatureParams(**cfg)
    ex = rsd_amd.FeatureExtractor(gp, 0)
    clips = [synth_clip(i, n=n) for i, n in ((1, 16000), (2, 9000), (3, 23456), (64, 12000), (7, 40000))]
    res = ex.extract_batch(clips, tuning=cfg["chroma"])
    assert (res["status"] == 0).all()
    for k, y in enumerate(clips):
        yf = O.pcm16_to_float(y)
        want = O.extract_audio_features(yf, 16000, op)
        d = np.abs(res["pooled"][k] - want)
        assert d[: 6 * cfg["n_mfcc"]].max() <= 1e-4, (k, float(d.max()), int(d.argmax()))
        if cfg["chroma"]:
            idx = O.estimate_tuning(O.stft_power(yf, op), op)[1]["index"]
            assert res["tuning"][k] == idx, (k, int(res["tuning"][k]), idx)
            assert d[6 * cfg["n_mfcc"]:].max() <= 1e-4, (k, "chroma", float(d[6 * cfg["n_mfcc"]:].max()))
    ex.close()


@pytest.mark.parametrize("cfg", [
    dict(n_fft=512, hop=256, n_mels=128, n_mfcc=40, chroma=True, rich=True),     # MAIN_286 = main.py:621-630, 718-763
    dict(n_fft=2048, hop=512, n_mels=128, n_mfcc=20, chroma=False, rich=True),
    dict(n_fft=1024, hop=256, n_mels=64, n_mfcc=13, chroma=True, rich=True),
], ids=["main286", "fft2048_rich", "fft1024_rich"])
def test_rich_descriptors_vs_oracle(cfg):
    """spectral-contrast mean/std, mean zero-crossing rate, mean RMS and mean spectral centroid of
    main.py:738-754 (kernel D) against the oracle's restatement of the librosa primitives (parity unpinned: the
    reference's own output for this set is all zeros, m:753).  Tolerances: MFCC/chroma blocks 1e-4 as everywhere;
    contrast (dB) 2e-4; zcr 1e-7; rms 1e-6; centroid (Hz, values ~1e3) 5e-3."""
    gp, op = rsd_amd.FeatureParams(**cfg), O.FeatureParams(**cfg)
    ex = rsd_amd.FeatureExtractor(gp, 0)
    assert ex.pooled_dim == op.audio_feature_len
    clips = [synth_clip(i, n=n) for i, n in ((1, 16000), (2, 9000), (3, 23456), (64, 12000), (7, 40000), (128, 30000))]
    res = ex.extract_batch(clips)
    assert (res["status"] == 0).all()
    base = 6 * cfg["n_mfcc"] + (24 if cfg["chroma"] else 0)
    for k, y in enumerate(clips):
        yf = O.pcm16_to_float(y)
        want = O.extract_audio_features(yf, 16000, op, strict=True)
        got = res["pooled"][k]
        assert np.abs(got[: 6 * cfg["n_mfcc"]] - want[: 6 * cfg["n_mfcc"]]).max() <= 1e-4
        r, w = got[base:], want[base:]
        assert np.abs(r[:14] - w[:14]).max() <= 2e-4, (k, "contrast", r[:14], w[:14])
        assert abs(r[14] - w[14]) <= 1e-7, (k, "zcr", float(r[14]), float(w[14]))
        assert abs(r[15] - w[15]) <= 1e-6, (k, "rms", float(r[15]), float(w[15]))
        assert abs(r[16] - w[16]) <= 5e-3, (k, "centroid", float(r[16]), float(w[16]))
    if cfg["n_fft"] == 512:
        assert ex.pooled_dim == 281 and rsd_amd.MAIN_286 == gp
    ex.close()


def test_lane_executor_equals_serial_path(ex512):
    """Batches alternating over three HIP streams / extractor handles (rsd_amd.pipeline.LaneExecutor, what bench.py runs)
    give bit-identical features to the single-stream path, in submission order, also when batches differ."""
    from rsd_amd.classifier import FusedMLPHead
    from rsd_amd.pipeline import LaneExecutor
    rng = np.random.default_rng(1)
    dims = [ex512.pooled_dim, 16, 3]
    head = FusedMLPHead(np.zeros(dims[0]), np.ones(dims[0]), [rng.standard_normal((dims[i], dims[i + 1])) * 0.05 for i in range(2)],
                        [np.zeros(dims[i + 1]) for i in range(2)])
    lanes = LaneExecutor(B512, 0, 3, head=head)
    batches = []
    for b in range(7):
        clips = [synth_clip(10 * b + i, n=8000 + 997 * ((b + i) % 5)) for i in range(6)]
        lens = [len(c) for c in clips]
        pcm = torch.from_numpy(np.concatenate(clips)).cuda()
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
        batches.append((pcm, off, max(lens)))
    torch.cuda.synchronize()
    lanes.fork()
    got = [lanes.submit(p, o, m, pooled=True, frames=True) for p, o, m in batches]
    lanes.join()
    torch.cuda.synchronize()
    for (p, o, m), g in zip(batches, got):
        want = ex512.extract_batch_device(p, o, m, pooled=True, frames=True)
        torch.cuda.synchronize()
        assert torch.equal(g["pooled"], want["pooled"]) and torch.equal(g["frames"], want["frames"])
        assert torch.equal(g["logits"], head(want["pooled"]))


def test_fused_cnn_head_vs_torch_module(ex512):
    """BASELINE configs[2]: the CNN head as hand-written HIP kernels (float32 conv1, float16-MFMA conv2 with float32
    accumulation, dysf_cnn_*) against the same torch module in float32 on the same frame-level features.  There is no
    reference CNN; the bar is the one configs[4] sets for fp16: finite, arg-max agreement (here 100 % on clips whose
    float32 margin exceeds the float16 noise) and logits within 2e-3 of the float32 module."""
    from rsd_amd.classifier import CNNHead, FusedCNNHead
    clips = np.stack([synth_clip(i) for i in range(24)])
    pcm = torch.from_numpy(clips.reshape(-1)).cuda()
    off = torch.arange(0, 25, dtype=torch.int64, device="cuda") * 48000
    fr = ex512.extract_batch_device(pcm, off, 48000, pooled=False, frames=True)["frames"]
    cnn = CNNHead().cuda()
    fused = FusedCNNHead(cnn)
    with torch.no_grad():
        want = cnn(fr)
    got = fused(fr)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max())
    assert err <= 2e-3 * max(1.0, float(want.abs().max())), err
    margin = want.sort(dim=1).values
    clear = (margin[:, -1] - margin[:, -2]) > 4 * err
    assert torch.equal(got.argmax(1)[clear], want.argmax(1)[clear]) and float(clear.float().mean()) >= 0.8
    # a padded row pitch and a ragged width (odd number of frames, pooled width floor(T/2)) take the same path
    fr2 = ex512.extract_batch_device(pcm, off, 48000, pooled=False, frames=True, frame_stride=320)["frames"]
    got2 = fused(fr2, cols=301)
    assert float((got2 - got).abs().max()) == 0.0
    # small and odd shapes (a single pixel tile, odd rows / columns dropped by the pooling floor)
    for shape in ((3, 10, 13), (2, 7, 40), (5, 32, 33)):
        x = torch.randn(*shape, device="cuda")
        with torch.no_grad():
            w = cnn(x)
        g = fused(x)
        assert float((g - w).abs().max()) <= 2e-3 * max(1.0, float(w.abs().max())), shape
    fused.close()


def test_fused_cnn_head_float32_matches_cpu_module_on_every_clip(ex512):
    """BASELINE configs[2] / SURVEY 8d Config 3: "CNN oracle = the same module on CPU fp32 (arg-max must match for fp32)".
    conv2 on v_mfma_f32_16x16x4_f32 (exact fmaf chains), float32 activations: arg-max equal on EVERY clip, logits within
    2e-5 of the CPU float32 module (summation order is the only difference)."""
    from rsd_amd.classifier import CNNHead, FusedCNNHead
    clips = np.stack([synth_clip(i) for i in range(64)])
    pcm = torch.from_numpy(clips.reshape(-1)).cuda()
    off = torch.arange(0, 65, dtype=torch.int64, device="cuda") * 48000
    fr = ex512.extract_batch_device(pcm, off, 48000, pooled=False, frames=True)["frames"]
    cnn = CNNHead()                      # float32 module on the CPU: the oracle
    fused = FusedCNNHead(cnn, precision="f32")
    with torch.no_grad():
        want = cnn(fr.cpu())
    got = fused(fr).cpu()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert torch.equal(got.argmax(1), want.argmax(1))
    fused.close()


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_fused_cnn_head_ragged_batch_equals_clip_by_clip_module(precision):
    """BASELINE configs[4]: clips of different lengths in one launch.  Convolution padding, the pooling floor and the
    global average follow every clip's OWN width: equal to the torch module run on each clip alone."""
    from rsd_amd.classifier import CNNHead, FusedCNNHead
    p = rsd_amd.FeatureParams(sr=22050, chroma=False, parity=False)
    ex = rsd_amd.get_extractor(p, 0)
    rng = np.random.default_rng(4321)
    lens = (rng.uniform(1.0, 10.0, size=7) * 22050).astype(int).tolist() + [5000, 512 * 9]
    clips = [synth_clip(200 + i, n=int(n), sr=22050) for i, n in enumerate(lens)]
    pcm = torch.from_numpy(np.concatenate(clips)).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device="cuda")
    fr = ex.extract_batch_device(pcm, off, max(lens), pooled=False, frames=True)["frames"]   # [9, 60, Tmax], zero past T_i
    T = torch.tensor([p.n_frames(n) for n in lens], dtype=torch.int32, device="cuda")
    cnn = CNNHead()
    fused = FusedCNNHead(cnn, precision=precision)
    got = fused(fr, cols=T).cpu()
    with torch.no_grad():
        want = torch.cat([cnn(fr[i:i + 1, :, :int(T[i])].cpu()) for i in range(len(lens))])
    tol = (2e-5 if precision == "f32" else 2e-3) * max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= tol, float((got - want).abs().max())
    if precision == "f32":
        assert torch.equal(got.argmax(1), want.argmax(1))
    # the padded batch without per-clip widths is a different (wrong for ragged input) computation: the widths matter
    assert float((fused(fr).cpu() - want).abs().max()) > tol
    fused.close()


def _run_bench(*flags, timeout=400):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], cwd=root, capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_two_ranks_rehearsal():
    """`python bench.py --gpus 2` exactly as a shell would invoke it: the script starts its own two rank processes (lanes,
    the logits all-gather on each lane's stream, max-over-ranks timing).  Rehearsed with both ranks on this box's GPU and a
    gloo gather (--share-gpu; never a measurement)."""
    d = _run_bench("--gpus", "2", "--steps", "6", "--warmup", "2", "--share-gpu", "--no-cpu-baseline", "--bank-workers", "2")
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"
    v = d["roofline"]["valu"]   # the second roof: useful float32 arithmetic against the vector peak
    assert v["peak_tflops"] == 157.3 and abs(v["frac"] - v["useful_flops_per_launch"] / (d["roofline"]["kernel_ms"] * 1e-3) / 1e12 / 157.3) < 2e-4
    r = d["roofline"]   # frac follows from the SURVEY 8d bytes and the measured kernel time alone
    assert r["bytes_per_clip"] == 2 * 48000 + 4 * (120 * 301 + 240)
    assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / r["peak"]) < 2e-4


def test_bench_under_torch_distributed_run():
    """The driver's form: `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` (ranks from the
    environment; the script must not start ranks of its own then)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--share-gpu", "--no-cpu-baseline", "--bank-workers", "2"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0


def test_bench_cnn_head_full_batch():
    """BASELINE configs[2] at its full size through bench.py: 1024 clips, frame rows -> HIP CNN head, MFMA roofline entry."""
    d = _run_bench("--head", "cnn", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--bank-workers", "2")
    assert d["config"]["head"] == "cnn" and d["value"] > 0 and d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1


def test_bench_config5_ragged_buckets_cnn():
    """BASELINE configs[4] through bench.py: 1-10 s clips at 22.05 kHz in ten length buckets, ragged launches, float16-MFMA CNN
    with per-clip widths; the chosen numbers are echoed into the JSON."""
    d = _run_bench("--config", "5", "--steps", "3", "--warmup", "1", "--bank-workers", "2")
    c = d["config"]
    assert c["sr"] == 22050 and c["seed"] == 4321 and sum(c["clips_per_bucket"].values()) == 1024 and len(c["clips_per_bucket"]) >= 9
    assert d["dtype"] == "f16" and d["roofline"]["bound"] == "mfma" and d["value"] > 0 and c["hours_of_audio_per_s"] > 0


def test_bench_config5_two_ranks_rehearsal():
    """BASELINE configs[4] at N = 2 as SURVEY 8e specifies it: ONE list of 2048 clips (seed 4321), contiguous shards balanced by
    cumulative sample count, buckets formed per rank, one all-gather of the unequal shards; the gathered logits are checked
    inside bench.py against a single-GPU pass over the whole list.  Both ranks on this box's GPU, gloo gather."""
    d = _run_bench("--gpus", "2", "--config", "5", "--steps", "2", "--warmup", "1", "--share-gpu", "--bank-workers", "2")
    c = d["config"]
    assert d["n_gpus"] == 2 and c["total_clips"] == 2048 and c["gathered_argmax_equals_single_gpu"] is True
    assert c["gathered_logits_max_abs_diff_vs_single_gpu_pass"] <= 1e-3 and "shard_by_samples" in c["shards"]
    assert c["clips_per_gpu"] != 1024   # balanced by samples, not by clip count


def test_chroma_preset_at_8_khz_wide_pitch_band():
    """sr 8000 at n_fft 2048: piptrack's 150-4000 Hz band spans ~985 bins, the chroma kernel's LDS candidate list is 126 KB
    (more than a kernel gets without the dynamic-LDS attribute).  The launch must work and agree with the oracle."""
    import dataclasses
    p = dataclasses.replace(rsd_amd.REFERENCE_149, sr=8000)
    op = dataclasses.replace(O.REFERENCE_149, sr=8000)
    ex = rsd_amd.FeatureExtractor(p, 0)
    clips = [synth_clip(i, n=20000, sr=8000) for i in range(3)]
    res = ex.extract_batch(clips, tuning=True)
    assert (res["status"] == 0).all()
    for i, c in enumerate(clips):
        want = O.extract_audio_features(O.pcm16_to_float(c), 8000, op)
        _, dbg = O.estimate_tuning(O.stft_power(O.pcm16_to_float(c), op), op)
        assert int(res["tuning"][i]) == dbg["index"]
        assert np.abs(res["pooled"][i] - want).max() <= 1e-4
    got = rsd_amd.audio_metrics_batch(clips, 8000)
    for i, c in enumerate(clips):
        w = O.audio_metrics(O.pcm16_to_float(c), 8000)
        assert abs(got[i, 0] - w[0]) <= 2e-4 and abs(got[i, 2] - w[2]) <= 2e-5 * max(w[2], 1e-6)
    ex.close()


def test_exact_2048_kernel_equals_kernel_a_bit_for_bit():
    """Kernel A-exact (parity presets at n_fft 2048: two workgroups per CU, half-size float64 exchange buffer, PCM straight
    to registers, packed |X|^2, piptrack's local maxima picked from the rows in LDS) runs the arithmetic of kernel A's float64
    path: pooled vectors, frame rows, tuning indices, statuses and the per-file metrics are bit-equal (DYSF_EXACT=1 keeps
    kernel A) -- ragged clips at odd offsets, clips shorter than a frame, silence (the general |X|^2 sequence), float32 PCM
    with non-finite samples, and the 8 kHz preset whose pitch band spans the whole row."""
    import dataclasses

    def pair(p):
        os.environ["DYSF_EXACT"] = "1"
        try:
            a = rsd_amd.FeatureExtractor(p, 0)
        finally:
            os.environ.pop("DYSF_EXACT", None)
        return a, rsd_amd.FeatureExtractor(p, 0)

    def same(r1, r2, what):
        for k in ("pooled", "frames", "status", "tuning"):
            assert np.array_equal(np.asarray(r1[k]), np.asarray(r2[k]), equal_nan=True), (what, k)

    rng = np.random.default_rng(11)
    base = [synth_clip(i) for i in range(12)]
    rag = [base[i % 12][: int(rng.integers(4097, 48001))].copy() for i in range(40)]
    rag += [base[0][:4095], base[1][:4096], base[2][:4097], base[3][:0], base[4][:1], base[5][:47999], base[6][:512 * 9 + 3],
            np.zeros(20000, np.int16), np.concatenate([base[7][:9000], np.zeros(9000, np.int16), base[7][:9000]])]
    ex_a, ex_x = pair(rsd_amd.REFERENCE_149)
    same(ex_a.extract_batch(rag, frames=True, tuning=True), ex_x.extract_batch(rag, frames=True, tuning=True), "int16")
    ragf = [c.astype(np.float32) / 32768.0 for c in rag[:24]]
    ragf[3] = ragf[3].copy(); ragf[3][1000] = np.nan
    ragf[5] = ragf[5].copy(); ragf[5][5] = np.inf
    r1, r2 = ex_a.extract_batch(ragf, frames=True, tuning=True), ex_x.extract_batch(ragf, frames=True, tuning=True)
    same(r1, r2, "float32")
    assert r2["status"][3] == 2 and r2["status"][5] == 2   # DYSF_NONFINITE

    def metrics_of(ex, clips):   # rsd_amd.audio_metrics_batch on a given handle
        import ctypes
        from rsd_amd import _lib
        pcm, offsets, kind = ex._pack(clips)
        out = torch.zeros((len(clips), 3), dtype=torch.float32, device="cuda:0")
        dpcm, doff = torch.from_numpy(pcm).cuda(), torch.from_numpy(offsets).cuda()
        _lib.check(_lib.lib().dysf_metrics_batch_dev(
            ex._h, ctypes.c_void_p(dpcm.data_ptr()), kind, ctypes.c_void_p(doff.data_ptr()), len(clips), int(np.max(np.diff(offsets))),
            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out.cpu().numpy()
    assert np.array_equal(metrics_of(ex_a, rag[:16]), metrics_of(ex_x, rag[:16]), equal_nan=True)
    # long clips: thousands of frames (prefix sums over many 64-frame steps, far more maxima than the selection keeps in
    # registers, chroma product without the K split), next to short ones in the same launch
    rng2 = np.random.default_rng(5)
    long_clips = [np.concatenate([base[(i + j) % 12] for j in range(k)])[: int(48000 * k - rng2.integers(0, 4000))]
                  for i, k in ((0, 42), (3, 7), (5, 1), (7, 13))]
    ra, rx = ex_a.extract_batch(long_clips, frames=False, tuning=True), ex_x.extract_batch(long_clips, frames=False, tuning=True)
    for k in ("pooled", "status", "tuning"):
        assert np.array_equal(np.asarray(ra[k]), np.asarray(rx[k]), equal_nan=True), ("long", k)
    want = O.extract_audio_features(O.pcm16_to_float(long_clips[1]), 16000, O.REFERENCE_149)
    assert np.abs(rx["pooled"][1] - want).max() <= 1e-4
    ex_a.close(); ex_x.close()
    p8 = dataclasses.replace(rsd_amd.REFERENCE_149, sr=8000)
    ex_a, ex_x = pair(p8)
    clips8 = [synth_clip(i, n=20000, sr=8000) for i in range(4)]
    same(ex_a.extract_batch(clips8, frames=True, tuning=True), ex_x.extract_batch(clips8, frames=True, tuning=True), "8 kHz")
    ex_a.close(); ex_x.close()


def test_bench_config4_two_ranks_rehearsal():
    """BASELINE configs[3] through the same self-launch path: 100 000 clips in two 50 000-clip shards (extract_large over the
    lanes), one all-gather of [50 000, 3] logits, gathered arg-max equal to the single-GPU result."""
    d = _run_bench("--gpus", "2", "--config", "4", "--steps", "2", "--warmup", "1", "--share-gpu", "--no-cpu-baseline",
                   "--bank-workers", "2")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_clips"] == 100000
    assert d["config"]["clips_per_gpu"] == 50000 and d["config"]["gathered_argmax_equals_single_gpu"] is True


def test_lane_executor_large_batch_equals_single_call(ex512):
    """LaneExecutor.extract_large: one resident batch of 5000 ragged clips cut into 1024-clip chunks over three lanes,
    written into slices of one output set -- bit-identical to the single-launch result."""
    from rsd_amd.pipeline import LaneExecutor
    rng = np.random.default_rng(3)
    base = [synth_clip(i, n=4000 + 531 * i) for i in range(16)]
    idx = rng.integers(0, 16, size=5000)
    lens = np.array([len(base[i]) for i in idx])
    pcm = torch.from_numpy(np.concatenate([base[i] for i in idx])).cuda()
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
    ms = int(lens.max())
    want = ex512.extract_batch_device(pcm, off, ms, pooled=True, frames=True)
    lanes = LaneExecutor(B512, 0, 3)
    got = lanes.extract_large(pcm, off, ms, chunk_clips=1024, pooled=True, frames=True)
    torch.cuda.synchronize()
    for k in ("status", "pooled", "frames"):
        assert torch.equal(got[k], want[k]), k


# ---------------------------------------------------------------------------------------------
# Cleaning stage (pipeline1.py:126-146): dysf_clean_batch_dev against oracle/dysf_clean_oracle.py.
# The oracle's gate is parity-unpinned (noisereduce absent, see its header); normalise + PCM_16 are pinned on reference data.
from oracle import dysf_clean_oracle as CO  # noqa: E402


def _noisy(i, n):
    rng = np.random.default_rng(100 + i)
    y = O.pcm16_to_float(synth_clip(i, n=n))
    return (y * np.float32(0.5) + np.float32(0.01) * rng.standard_normal(n).astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("prop", [1.0, 0.8], ids=["p1_default", "main_prop0.8"])
def test_cleaning_stage_matches_oracle(prop, golden_dir):
    """nr.reduce_noise -> librosa.util.normalize -> PCM_16 (p1:139-141; m:655 with prop_decrease 0.8): the int16 samples
    the reference would write, bit for bit, on a ragged batch of float32 and int16-derived clips."""
    clips = [_noisy(0, 48000), _noisy(1, 16001), _noisy(2, 1000), _noisy(5, 255), _noisy(7, 70000)]
    clips.append(O.pcm16_to_float(next(iter(_pairs(golden_dir)))[1]))       # a reference clear_audio clip, cleaned again
    cl = rsd_amd.AudioCleaner(sr=16000, device=0, prop_decrease=prop)
    res = cl.clean_batch(clips, denoised=True, normalized=True, pcm16=True)
    assert (res["status"] == 0).all()
    mism = 0
    for y, den, nrm, q in zip(clips, res["denoised"], res["normalized"], res["pcm16"]):
        ref = CO.reduce_noise(y, 16000, prop)
        assert den.dtype == np.float32 and den.shape == ref.shape
        assert np.abs(den.astype(np.float64) - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-30)
        ref_n = CO.normalize(ref)
        assert np.abs(nrm - ref_n).max() <= 2e-6
        assert np.abs(nrm).max() == 1.0
        ref_q = CO.float_to_pcm16(ref_n)
        assert q.dtype == np.int16
        d = np.abs(q.astype(np.int32) - ref_q.astype(np.int32))
        assert d.max() <= 1                      # float64 round-off may move a sample across a rounding boundary
        mism += int((d != 0).sum())
        # the integer step itself is exact: the oracle's normalise + PCM_16 applied to OUR float32 signal
        assert (q == CO.float_to_pcm16(CO.normalize(den))).all()
        assert (nrm == CO.normalize(den)).all()
    assert mism <= 3, mism                       # of ~190 000 samples
    cl.close()


def test_cleaning_int16_input_equals_float32_input():
    y16 = synth_clip(9, n=30000)
    a = rsd_amd.get_cleaner(16000, 0).clean_batch([y16], pcm16=True, denoised=True)
    b = rsd_amd.get_cleaner(16000, 0).clean_batch([O.pcm16_to_float(y16)], pcm16=True, denoised=True)
    assert (a["pcm16"][0] == b["pcm16"][0]).all() and (a["denoised"][0] == b["denoised"][0]).all()
    one = rsd_amd.reduce_noise(O.pcm16_to_float(y16), 16000)
    assert one.dtype == np.float32 and (one == a["denoised"][0]).all()
    with pytest.raises(NotImplementedError):
        rsd_amd.reduce_noise(O.pcm16_to_float(y16), 16000, stationary=True)


def test_cleaning_error_convention_and_batch_independence():
    """An all-zero clip makes the gate divide 0 by 0: librosa.util.normalize raises on the NaNs and the reference writes
    no file (p1:143-145).  Here: status NONFINITE, zero outputs, and the neighbours in the batch are untouched."""
    good = _noisy(3, 20000)
    res = rsd_amd.get_cleaner(16000, 0).clean_batch([good, np.zeros(5000, np.float32), np.zeros(0, np.float32), good],
                                                    pcm16=True, normalized=True)
    assert res["status"].tolist() == [0, 2, 3, 0]          # DYSF_OK, DYSF_NONFINITE, DYSF_EMPTY, DYSF_OK
    assert (res["pcm16"][1] == 0).all() and len(res["pcm16"][2]) == 0
    alone = rsd_amd.get_cleaner(16000, 0).clean_batch([good], pcm16=True)
    assert (res["pcm16"][0] == alone["pcm16"][0]).all() and (res["pcm16"][3] == alone["pcm16"][0]).all()
    with pytest.raises(ValueError):
        CO.normalize(CO.reduce_noise(np.zeros(5000, np.float32), 16000))    # the reference's failure, same clip
    # device entry point with a max_samples below a clip's length: that clip is rejected (status 4), the others are cleaned
    import torch
    short = _noisy(6, 6000)
    pcm = torch.from_numpy(np.concatenate([short, good])).cuda()
    off = torch.tensor([0, len(short), len(short) + len(good)], dtype=torch.int64, device="cuda")
    r2 = rsd_amd.get_cleaner(16000, 0).clean_batch_device(pcm, off, 8000, pcm16=True)
    assert r2["status"].cpu().tolist() == [0, 4]
    q = r2["pcm16"].cpu().numpy()
    assert (q[:len(short)] == rsd_amd.get_cleaner(16000, 0).clean_batch([short], pcm16=True)["pcm16"][0]).all() and (q[len(short):] == 0).all()


def test_clean_audio_and_cache_end_to_end(tmp_path):
    """pipeline1.py:126-146 as one call: WAV in -> clear_dir/<stem>.wav (memoised) -> the feature path reads it back."""
    y = _noisy(4, 40000)
    src = tmp_path / "raw clip (1).wav"
    rsd_amd.write_wav_pcm16(str(src), CO.float_to_pcm16(y), 16000)
    clear = tmp_path / "clear_audio"
    out = rsd_amd.clean_audio_and_cache(str(src), clear_dir=str(clear))
    assert out == os.path.normpath(str(clear / "raw clip (1).wav")) and os.path.exists(out)
    pcm, sr = O.read_wav_pcm16(out)
    y_in, _ = rsd_amd.load_audio(str(src))
    assert sr == 16000 and (pcm == CO.clean_audio(y_in, 16000)).mean() > 0.9999
    assert max(pcm.max(), -int(pcm.min())) in (32767, 32768)
    stamp = os.path.getmtime(out)
    assert rsd_amd.clean_audio_and_cache(str(src), clear_dir=str(clear)) == out and os.path.getmtime(out) == stamp
    y2, sr2 = rsd_amd.load_audio(out)
    feats = rsd_amd.extract_features(y2, sr2)
    assert feats.shape == (149,) and np.abs(feats - O.extract_features(O.pcm16_to_float(pcm), 16000)).max() <= 1e-4
    silent = tmp_path / "silent.wav"
    rsd_amd.write_wav_pcm16(str(silent), np.zeros(3000, np.int16), 16000)
    assert rsd_amd.clean_audio_and_cache(str(silent), clear_dir=str(clear)) is None
    assert rsd_amd.clean_audio_and_cache(str(tmp_path / "missing.wav"), clear_dir=str(clear)) is None


def test_preprocess_loop_and_feature_caching_as_batched_passes(golden_dir, tmp_path):
    """pipeline1.py:371-417 + 447-456 through rsd_amd.preprocess_files / cached_extract_features_batch: labels from the folder,
    metrics before / after, clear_dir memoisation (a reference clear_audio file already present is reused, so its `after`
    columns must equal the reference's per_file_analysis.csv), fall-back to the original when cleaning fails, and the cache
    files of the feature loop equal to the reference's committed vectors."""
    idx = json.load(open(os.path.join(golden_dir, "ref_pairs", "index.json")))
    root, clear, cache = tmp_path / "segrigated_samples", tmp_path / "clear_audio", tmp_path / "cache_features"
    (root / "word repetition").mkdir(parents=True); (root / "Prolongatio sample").mkdir(); clear.mkdir()
    # (a) two noisy synthetic "raw" files; (b) three reference stems whose cleaned WAV already sits in clear_dir; (c) silence
    rsd_amd.write_wav_pcm16(str(root / "word repetition" / "w1.wav"), CO.float_to_pcm16(_noisy(11, 30000)), 16000)
    rsd_amd.write_wav_pcm16(str(root / "word repetition" / "w2 (1).wav"), CO.float_to_pcm16(_noisy(12, 21000)), 16000)
    rsd_amd.write_wav_pcm16(str(root / "word repetition" / "silent.wav"), np.zeros(4000, np.int16), 16000)
    (root / "word repetition" / "notes.txt").write_text("not audio")
    (root / "word repetition" / "broken.wav").write_bytes(b"RIFFxxxx")
    ref = {}
    for (cid, pcm, sr, gold), e in list(zip(_pairs(golden_dir), idx))[:3]:
        stem = e["reference_stem"]
        rsd_amd.write_wav_pcm16(str(root / "Prolongatio sample" / f"{stem}.wav"), CO.float_to_pcm16(_noisy(20 + len(ref), len(pcm))), 16000)
        rsd_amd.write_wav_pcm16(str(clear / f"{stem}.wav"), pcm, 16000)      # the reference's own cleaned file
        ref[f"{stem}.wav"] = (e["metrics_after"], gold, pcm)
    files = rsd_amd.list_audio_files(str(root))
    assert len(files) == 7 and all(f.endswith(".wav") for f in files) and files == sorted(files)
    res = rsd_amd.preprocess_files(files, clear_dir=str(clear), batch=4)
    rows = {r["file"]: r for r in res["per_file"]}
    assert res["skipped"] == 1 and len(rows) == 6 and "broken.wav" not in rows
    assert rows["w1.wav"]["label"] == "word repetition" and abs(rows["w1.wav"]["duration_sec"] - 30000 / 16000) < 1e-12
    for name, (want, gold, pcm) in ref.items():          # memoised reference files: the CSV's `after` columns
        r = rows[name]
        assert r["label"] == "Prolongatio sample"
        assert abs(r["snr_after_db"] - want[0]) <= 2e-4 and abs(r["spectral_flatness_after"] - want[1]) <= 5e-4 * want[1]
        assert abs(r["hf_energy_ratio_after"] - want[2]) <= 2e-5 * want[2]
        assert (O.read_wav_pcm16(str(clear / name))[0] == pcm).all()          # not overwritten
    # freshly cleaned files: written, equal to the oracle's samples, and the `after` metrics are those of the written file
    for name in ("w1.wav", "w2 (1).wav"):
        y_in, _ = rsd_amd.load_audio(str(root / "word repetition" / name))
        q, _ = O.read_wav_pcm16(str(clear / name))
        assert (q != CO.clean_audio(y_in, 16000)).sum() <= 1
        m = rsd_amd.audio_metrics_batch([q], 16000)[0]
        assert abs(rows[name]["snr_after_db"] - m[0]) <= 1e-4 and rows[name]["snr_after_db"] > rows[name]["snr_before_db"]
        b = rsd_amd.audio_metrics_batch([O.read_wav_pcm16(str(root / "word repetition" / name))[0]], 16000)[0]
        assert abs(rows[name]["snr_before_db"] - b[0]) <= 1e-4 and abs(rows[name]["hf_energy_ratio_before"] - b[2]) <= 1e-6
    # cleaning failed for the silent file (non-finite gate output): the original is used, as p1:386-388 does
    assert not (clear / "silent.wav").exists()
    assert res["clean_paths"][res["raw_paths"].index(str(root / "word repetition" / "silent.wav"))].endswith(os.path.join("word repetition", "silent.wav"))
    rsd_amd.write_per_file_analysis(res["per_file"], str(tmp_path / "per_file_analysis.csv"))
    import csv as _csv
    back = list(_csv.DictReader(open(tmp_path / "per_file_analysis.csv")))
    assert len(back) == 6 and float(back[0]["snr_after_db"]) == res["per_file"][0]["snr_after_db"] and back[0]["transcript"] == ""
    # feature loop p1:447-456
    Xa = rsd_amd.cached_extract_features_batch(res["clean_paths"], "clean", cache_dir=str(cache), batch=4)
    Xb = rsd_amd.cached_extract_features_batch(res["raw_paths"], "raw", cache_dir=str(cache), batch=4)
    assert Xa.shape == Xb.shape == (6, 149) and Xa.dtype == np.float32
    for i, p in enumerate(res["clean_paths"]):
        name = os.path.basename(p)
        blob = open(cache / (name.rsplit(".", 1)[0] + "_clean_feats.npy"), "rb").read()
        assert len(blob) == 724 and (np.load(cache / (name.rsplit(".", 1)[0] + "_clean_feats.npy")) == Xa[i]).all()
        if name in ref:
            assert np.abs(Xa[i] - ref[name][1]).max() <= 1e-4          # the reference's cache_features vector
        assert (Xa[i] == rsd_amd.cached_extract_features(p, "", "clean", cache_dir=str(cache))).all()
    assert (Xa[res["raw_paths"].index(str(root / "word repetition" / "silent.wav"))][:144] == O.extract_features(np.zeros(4000, np.float32), 16000)[:144]).all()
    assert (rsd_amd.cached_extract_features_batch(res["clean_paths"], "clean", cache_dir=str(cache)) == Xa).all()   # second run: cache hits


def test_deployed_predictor_object_with_cleaning(golden_dir, tmp_path):
    """main1.py:949-990 through rsd_amd.DeployedPredictor: (i) an upload whose cleaned WAV already sits in clear_audio/ (the
    reference's own file) must give scikit-learn's label and probabilities for the reference's cached vector; (ii) a fresh
    noisy upload is cleaned with main1's prop_decrease=0.8, memoised, and classified like the oracle's chain says;
    (iii) silence: cleaning fails, the input itself is classified (m1:609 returns in_path); (iv) unreadable input."""
    from rsd_amd.classifier import RandomForestHead
    c = np.load(os.path.join(golden_dir, "classifier_after.npz"), allow_pickle=False)
    f = np.load(os.path.join(golden_dir, "forest_after.npz"), allow_pickle=False)
    classes = [str(x) for x in c["classes"]]
    rf = RandomForestHead.from_npz(os.path.join(golden_dir, "forest_after.npz"), classes=classes)
    clear = tmp_path / "clear_audio"
    clear.mkdir()
    dp = rsd_amd.DeployedPredictor(rf, classes, clear_dir=str(clear))
    idx = json.load(open(os.path.join(golden_dir, "ref_pairs", "index.json")))
    for (cid, pcm, sr, gold), e in list(zip(_pairs(golden_dir), idx))[:4]:
        up = tmp_path / f"{e['reference_stem']}.wav"                       # the "upload": any audio with this stem ...
        rsd_amd.write_wav_pcm16(str(up), CO.float_to_pcm16(_noisy(30, len(pcm))), 16000)
        rsd_amd.write_wav_pcm16(str(clear / up.name), pcm, 16000)           # ... whose cleaned file the reference already has
        rows = np.nonzero((c["X"] == gold[None, :]).all(axis=1))[0]
        label, prob = dp.predict_file(str(up))
        assert label == classes[int(f["expected_pred"][rows[0]])], (cid, label)
        assert max(abs(prob[k] - f["expected_proba"][rows[0]][i]) for i, k in enumerate(classes)) <= 0.0051
        assert abs(sum(prob.values()) - 1.0) < 1e-6
    # (ii) fresh upload
    y = _noisy(31, 26000)
    up = tmp_path / "fresh upload.wav"
    rsd_amd.write_wav_pcm16(str(up), CO.float_to_pcm16(y), 16000)
    label, prob = dp.predict_file(str(up))
    y_in, _ = rsd_amd.load_audio(str(up))
    want_pcm = CO.clean_audio(y_in, 16000, 0.8)
    got_pcm, _ = O.read_wav_pcm16(str(clear / "fresh upload.wav"))
    assert (got_pcm != want_pcm).sum() <= 1
    feats = O.extract_features(O.pcm16_to_float(got_pcm), 16000)
    x = torch.from_numpy(feats.reshape(1, -1)).cuda()
    assert label == classes[int(rf.predict(x)[0])] and label in classes
    assert max(abs(prob[k] - float(rf.predict_proba(x)[0][i])) for i, k in enumerate(classes)) <= 0.0051
    assert dp.predict_file(str(up))[0] == label                              # memoised cleaned file
    # (iii) silence: the gate is non-finite, main1 falls back to the input path; features of silence still classify
    sil = tmp_path / "silence.wav"
    rsd_amd.write_wav_pcm16(str(sil), np.zeros(8000, np.int16), 16000)
    label_s, prob_s = dp.predict_file(str(sil))
    assert label_s in classes and not (clear / "silence.wav").exists()
    # (iv) unreadable
    assert dp.predict_file(str(tmp_path / "nope.wav")) == (None, None)
    rf.close()


@pytest.mark.parametrize("dims", [(37, 5), (149, 200, 17, 3), (512, 256, 256, 130, 7), (20, 16, 16), (300, 100, 2)],
                         ids=["1layer", "wide_hidden_nb4", "max_sizes", "exact_16s", "two_layer"])
def test_fused_mlp_head_generic_shapes(dims):
    """dysf_mlp_kernel for layer shapes other than the reference's 149-128-64-3: inputs and widths that are not multiples
    of 16, up to the 512-input / 256-unit limits, one to four layers (all three blocks-per-wave builds), ragged batch
    tails -- against the eager float64 module (the MFMA accumulates float32: 2e-5 of the logit scale)."""
    from rsd_amd.classifier import FusedMLPHead, MLPHead
    rng = np.random.default_rng(len(dims) * 1000 + dims[0])
    ws = [rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]) * 0.1 for i in range(len(dims) - 1)]
    mean, scale = rng.standard_normal(dims[0]), rng.uniform(0.5, 2.0, dims[0])
    fused = FusedMLPHead(mean, scale, ws, bs)
    eager = MLPHead(mean, scale, ws, bs, dtype=torch.float64).cuda()
    for n in (1, 15, 16, 17, 1000):
        x = torch.from_numpy(rng.standard_normal((n, dims[0])).astype(np.float32)).cuda()
        got, want = fused(x).cpu().numpy(), eager(x).cpu().numpy()
        assert got.shape == want.shape == (n, dims[-1])
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (n, np.abs(got - want).max())
        assert torch.equal(fused(x), fused(x))                              # fixed summation order
        p = fused.predict_proba(x).cpu().numpy()
        assert np.abs(p.sum(axis=1) - 1).max() <= 1e-5
        clear = np.sort(want, axis=1)[:, -1] - np.sort(want, axis=1)[:, -2] > 1e-3 if dims[-1] > 1 else np.ones(n, bool)
        assert (fused.predict(x).cpu().numpy()[clear] == want.argmax(axis=1)[clear]).all()
    fused.close()
    with pytest.raises(rsd_amd.DysfError):
        FusedMLPHead(np.zeros(8), np.ones(8), [rng.standard_normal((8, 300))], [np.zeros(300)])   # wider than 256 units


@pytest.mark.parametrize("sr,n", [(22050, 50000), (8000, 12000), (16000, 160001), (44100, 30000)], ids=["22k", "8k", "10s", "44k"])
def test_cleaning_stage_other_rates_and_lengths(sr, n):
    """The gate's smoothing widths follow the sample rate (SpectralGate._generate_mask_smoothing_filter: 16 x 3 bins x frames
    at 16 kHz, 11 x 4 at 22.05 kHz, 32 x 1 at 8 kHz, 5 x 8 at 44.1 kHz) and the one-pole coefficient follows sr / hop: the same
    bit-level agreement with the oracle at other rates, and on a 10 s clip (more frames than any 3 s test)."""
    y = _noisy(40 + sr // 1000, n)
    cl = rsd_amd.AudioCleaner(sr=sr, device=0)
    res = cl.clean_batch([y], denoised=True, pcm16=True)
    assert int(res["status"][0]) == 0
    ref = CO.reduce_noise(y, sr)
    assert np.abs(res["denoised"][0].astype(np.float64) - ref).max() <= 1e-6 * np.abs(ref).max()
    want = CO.float_to_pcm16(CO.normalize(ref))
    d = np.abs(res["pcm16"][0].astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and int((d != 0).sum()) <= 2, (int(d.max()), int((d != 0).sum()))
    cl.close()


# ------------------------------------------------------------------------------------------------
# All 888 reference pairs and the models main1.py deploys (round 3)

def test_reference_888_pairs_all_features_tuning_and_metrics(golden_dir):
    """EVERY known-answer pair the reference holds: clear_audio/<stem>.wav -> cache_features/<stem>_clean_feats.npy (888
    pairs; the audio travels to the GPU box in the git-ignored tests/golden/_all888/, so the test skips where it is
    absent).  One batched launch.  Asserted: all 149 values of every clip within the north-star 1e-4; the tuning index of
    chroma_stft equal to the oracle's (tests/golden/tuning_888.json) on every clip; the three per-file metrics of
    pipeline1.py:151-186 against the reference's own per_file_analysis.csv values (tolerances of
    test_audio_metrics_match_reference_csv_and_oracle)."""
    d = os.path.join(golden_dir, "_all888")
    if not os.path.exists(os.path.join(d, "index.json")):
        pytest.skip("tests/golden/_all888 (the reference's 888 WAV / vector pairs) is not on this machine")
    idx = json.load(open(os.path.join(d, "index.json")))
    assert len(idx) == 888
    want_tuning = json.load(open(os.path.join(golden_dir, "tuning_888.json")))["tuning_index"]
    clips, golds = [], []
    for e in idx:
        pcm, sr = O.read_wav_pcm16(os.path.join(d, e["id"] + ".wav"))
        assert sr == 16000
        clips.append(pcm); golds.append(np.load(os.path.join(d, e["id"] + ".npy")))
    ex = rsd_amd.get_extractor(P149, 0)
    res = ex.extract_batch(clips, tuning=True)
    assert (res["status"] == 0).all()
    G = np.stack(golds).astype(np.float64)
    F = np.hstack([res["pooled"], np.zeros((len(clips), 5), np.float32)]).astype(np.float64)
    D = np.abs(F - G)
    worst = int(D.max(axis=1).argmax())
    print("888 pairs: max |gpu - reference| %.3e (clip %s, feature %d), p99 of per-clip maxima %.3e, chroma block max %.3e"
          % (D.max(), idx[worst]["stem"], int(D[worst].argmax()), np.percentile(D.max(axis=1), 99), D[:, 120:144].max()))
    assert D.max() <= 1e-4, (idx[worst]["stem"], float(D.max()), int(D[worst].argmax()))
    assert int((D.max(axis=1) <= 1e-4).sum()) == 888
    mism = [(e["stem"], int(t), want_tuning[e["stem"]]) for e, t in zip(idx, res["tuning"]) if int(t) != want_tuning[e["stem"]]]
    assert not mism, mism[:5]
    have = [i for i, e in enumerate(idx) if e.get("metrics_after")]
    assert len(have) == 888
    gm = rsd_amd.audio_metrics_batch([clips[i] for i in have], 16000)
    wm = np.array([idx[i]["metrics_after"] for i in have])
    assert np.abs(gm[:, 0] - wm[:, 0]).max() <= 2e-4
    assert (np.abs(gm[:, 1] - wm[:, 1]) / wm[:, 1]).max() <= 5e-4
    assert (np.abs(gm[:, 2] - wm[:, 2]) / wm[:, 2]).max() <= 2e-5


def _deployed(golden_dir):
    return os.path.join(golden_dir, "deployed_models.npz")


def test_deployed_forest_600_trees_all_rows(golden_dir):
    """The forest main1.py serves (m1:860-862 fitted on all 905 scaled rows, m1:896-897): RandomForest(600) as flat node
    arrays (313 k nodes), walked by dysf_forest_predict_dev.  scikit-learn's predict exact, predict_proba to 1e-12."""
    from rsd_amd.classifier import RandomForestHead
    z = np.load(_deployed(golden_dir))
    assert len(z["rf_root"]) == 600
    head = RandomForestHead.from_npz(_deployed(golden_dir), prefix="rf_")
    x = torch.from_numpy(z["X"]).cuda()
    assert np.array_equal(head.predict(x).cpu().numpy(), z["rf_pred"])
    assert np.abs(head.predict_proba(x).cpu().numpy() - z["rf_proba"]).max() <= 1e-12
    head.close()


def test_deployed_mlp_three_hidden_layers(golden_dir):
    """MLPClassifier(256, 128, 64) of m1:863-864 through the one-kernel MFMA head and the eager head: scikit-learn's arg-max
    on every row whose top-two probability margin exceeds float32 noise (1e-3), probabilities within 1e-4."""
    from rsd_amd.classifier import FusedMLPHead, MLPHead
    z = np.load(_deployed(golden_dir))
    fused = FusedMLPHead.from_npz(_deployed(golden_dir), prefix="mlp_")
    eager = MLPHead.from_npz(_deployed(golden_dir), prefix="mlp_").cuda()
    assert len(eager.weights) == 4
    x = torch.from_numpy(z["X"]).cuda()
    srt = np.sort(z["mlp_proba"], axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    for head in (fused, eager):
        assert np.abs(head.predict_proba(x).cpu().numpy() - z["mlp_proba"]).max() <= 1e-4
        assert np.array_equal(head.predict(x).cpu().numpy()[clear], z["mlp_pred"][clear])
    assert clear.mean() > 0.99
    fused.close()


def test_deployed_svc_c10(golden_dir):
    """SVC(C=10, gamma='scale', probability=True) of m1:865 fitted on all rows: decision values and coupled Platt
    probabilities to 1e-9, one-vs-one votes exact."""
    from rsd_amd.classifier import SVCHead
    z = np.load(_deployed(golden_dir))
    head = SVCHead.from_npz(_deployed(golden_dir), prefix="svc_")
    x = torch.from_numpy(z["X"]).cuda()
    assert np.abs(head.decision_function(x).cpu().numpy() - z["svc_decision"]).max() <= 1e-9
    assert np.array_equal(head.predict(x).cpu().numpy(), z["svc_pred"])
    assert np.abs(head.predict_proba(x).cpu().numpy() - z["svc_proba"]).max() <= 1e-9
    head.close()


def test_deployed_soft_voting_ensemble(golden_dir):
    """VotingClassifier(voting='soft') of m1:867-868 over the three deployed models against scikit-learn's own ensemble
    outputs (its members are clones re-fitted on the same rows: the fixture records that their mean equals it)."""
    from rsd_amd.classifier import FusedMLPHead, RandomForestHead, SVCHead, SoftVotingHead
    z = np.load(_deployed(golden_dir))
    assert json.loads(str(z["evidence"]))["ensemble_is_mean_of_the_three"] < 1e-12
    heads = [RandomForestHead.from_npz(_deployed(golden_dir), prefix="rf_"),
             FusedMLPHead.from_npz(_deployed(golden_dir), prefix="mlp_"),
             SVCHead.from_npz(_deployed(golden_dir), prefix="svc_")]
    vote = SoftVotingHead(heads)
    x = torch.from_numpy(z["X"]).cuda()
    got = vote.predict_proba(x).cpu().numpy()
    assert np.abs(got - z["ens_proba"]).max() <= 1e-4
    srt = np.sort(z["ens_proba"], axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert np.array_equal(vote.predict(x).cpu().numpy()[clear], z["ens_pred"][clear]) and clear.mean() > 0.97


def test_deployed_predictor_with_the_served_forest(golden_dir, tmp_path):
    """main1.py:970-990 end to end with the model it names: reference WAV (already cleaned: memoised in clear_audio/) ->
    features -> guard -> scaler + RandomForest(600, all rows) -> label string, against scikit-learn's label and
    probabilities for the reference's cached vector of the same clip."""
    from rsd_amd.classifier import RandomForestHead
    z = np.load(_deployed(golden_dir))
    classes = [str(c) for c in z["classes"]]
    rf = RandomForestHead.from_npz(_deployed(golden_dir), classes=classes, prefix="rf_")
    clear = tmp_path / "clear_audio"
    clear.mkdir()
    dp = rsd_amd.DeployedPredictor(rf, classes, clear_dir=str(clear))
    idx = json.load(open(os.path.join(golden_dir, "ref_pairs", "index.json")))
    n = 0
    for (cid, pcm, sr, gold), e in zip(_pairs(golden_dir), idx):
        rows = np.nonzero((z["X"] == gold[None, :]).all(axis=1))[0]
        if len(rows) == 0:
            continue
        up = tmp_path / f"{e['reference_stem']}.wav"
        rsd_amd.write_wav_pcm16(str(up), pcm, 16000)
        rsd_amd.write_wav_pcm16(str(clear / up.name), pcm, 16000)
        label, prob = dp.predict_file(str(up))
        # the GPU features differ from the cached vector by <= 1e-4: a split threshold can fall in between on a few trees
        assert max(abs(prob[k] - z["rf_proba"][rows[0]][i]) for i, k in enumerate(classes)) <= 0.0051, cid
        srt = np.sort(z["rf_proba"][rows[0]])
        if srt[-1] - srt[-2] > 0.011:
            assert label == classes[int(z["rf_pred"][rows[0]])], (cid, label)
        n += 1
    assert n >= 12
    rf.close()


def test_cleaning_long_clips_are_gated_in_noisereduce_chunks():
    """Clips beyond noisereduce's chunk_size (600 000 samples = 37.5 s at 16 kHz) -- the reference's 126.5 s sidebar uploads
    went through main1.py:598-611 like this -- are gated in full 600 000-sample spans, each inside its 30 000 real
    neighbouring samples, and the spans concatenated (SpectralGate.get_traces).  A mixed batch (short, exactly one chunk,
    one sample over, two and a bit chunks) against the oracle's chunk loop: float32 signal to 1e-6 of its peak, int16
    samples equal up to two one-LSB differences per clip."""
    lens = [26000, 600000, 600001, 1250000]
    clips = [_noisy(60 + i, n) for i, n in enumerate(lens)]
    cl = rsd_amd.AudioCleaner(sr=16000, device=0, prop_decrease=0.8)
    res = cl.clean_batch(clips, denoised=True, pcm16=True)
    assert res["status"].tolist() == [0, 0, 0, 0]
    for i, y in enumerate(clips):
        ref = CO.reduce_noise(y, 16000, 0.8)
        assert ref.shape == y.shape
        assert np.abs(res["denoised"][i].astype(np.float64) - ref).max() <= 1e-6 * np.abs(ref).max(), lens[i]
        want = CO.float_to_pcm16(CO.normalize(ref))
        d = np.abs(res["pcm16"][i].astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and int((d != 0).sum()) <= 2, (lens[i], int(d.max()), int((d != 0).sum()))
    cl.close()


def test_feature_kernels_are_unchanged_beside_the_float16_cnn_on_another_stream():
    """Concurrency with float16 MFMA work: on gfx950 packed float32 adds / multiplies with mixed source halves go wrong while
    another wave of the CU runs a float16 MFMA (dysf_fft.hpp add_mi; tools/ubench/coexec.hip) -- round 2's float32 FFT kernels
    returned wrong frames whenever the float16 CNN head of another lane shared a CU with them.  Nine feature launches on three
    lanes, alone and with the float16 CNN running on a fourth stream over fixed frames: bit-identical outputs, for the
    default preset (kernel A2), kernel A, and the 22.05 kHz / 2048-point preset of configs[4]."""
    from rsd_amd.classifier import CNNHead, FusedCNNHead
    from rsd_amd.pipeline import LaneExecutor
    dev = torch.device("cuda", 0)
    bank = np.stack([synth_clip(i) for i in range(16)])
    pcm = torch.from_numpy(bank[np.arange(256) % 16].reshape(-1)).to(dev)
    off = (torch.arange(0, 257, dtype=torch.int64) * 48000).to(dev)
    torch.manual_seed(0)
    cnn = FusedCNNHead(CNNHead().to(dev))
    fixed = torch.randn(128, 60, 301, device=dev) * 20 - 30
    side = torch.cuda.Stream(dev)
    for name, p, env in (("A2", B512, None), ("A", B512, "1"), ("22k", rsd_amd.FeatureParams(sr=22050, chroma=False, parity=False), None)):
        if env:
            os.environ["DYSF_STFT"] = env
        try:
            lanes = LaneExecutor(p, 0, 3)
        finally:
            os.environ.pop("DYSF_STFT", None)

        def run(with_cnn):
            outs = []
            lanes.fork()
            for i in range(9):
                k = i % 3
                with torch.cuda.stream(lanes.streams[k]):
                    outs.append(lanes.extractors[k].extract_batch_device(pcm, off, 48000, pooled=True, frames=True))
                if with_cnn:
                    with torch.cuda.stream(side):
                        cnn(fixed)
            lanes.join()
            torch.cuda.synchronize()
            return outs
        ref = run(False)
        for rep in range(2):
            got = run(True)
            for i in range(9):
                assert torch.equal(ref[0]["frames"], got[i]["frames"]) and torch.equal(ref[0]["pooled"], got[i]["pooled"]), (name, rep, i)
