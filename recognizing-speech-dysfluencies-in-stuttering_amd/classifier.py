"""Classifier heads (PyTorch-ROCm; the only part of the path that is not hand-written HIP).

MLPHead reproduces the inference half of the reference's classify step -- ``StandardScaler.transform`` followed
by ``MLPClassifier(hidden_layer_sizes=(128, 64)).predict / predict_proba`` (pipeline1.py:470-473, 497, 512-514;
single-clip form main1.py:987-990) -- from exported arrays.  Training stays scikit-learn on CPU (out of scope).
CNNHead is the benchmark-only 2-layer CNN of BASELINE.json configs[2]; the reference has no CNN.
"""
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _npz_layers(z, prefix: str = ""):
    """(weights, biases) of a dense stack stored as <prefix>W1, <prefix>b1, <prefix>W2, ... in an .npz fixture."""
    Ws, bs = [], []
    while f"{prefix}W{len(Ws) + 1}" in z.files:
        i = len(Ws) + 1
        Ws.append(z[f"{prefix}W{i}"]); bs.append(z[f"{prefix}b{i}"])
    if not Ws:
        raise KeyError(f"no {prefix}W1 in the fixture")
    return Ws, bs


class MLPHead(nn.Module):
    """(x - mean_) / scale_  ->  relu(x W1 + b1) -> relu(. W2 + b2) -> . W3 + b3  (logits); softmax for proba."""

    def __init__(self, mean, scale, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], classes=None,
                 dtype=torch.float32):
        super().__init__()
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).to(dtype)  # noqa: E731
        self.register_buffer("mean", t(mean))
        self.register_buffer("inv_scale", t(1.0 / np.asarray(scale, dtype=np.float64)))
        self.weights = nn.ParameterList([nn.Parameter(t(w), requires_grad=False) for w in weights])
        self.biases = nn.ParameterList([nn.Parameter(t(b), requires_grad=False) for b in biases])
        self.classes = None if classes is None else [str(c) for c in classes]
        self.n_features_in_ = int(np.asarray(mean).shape[0])

    @classmethod
    def from_npz(cls, path: str, dtype=torch.float32, prefix: str = "") -> "MLPHead":
        """``prefix`` selects one model of a multi-model fixture (tests/golden/deployed_models.npz: "mlp_"); any number
        of layers W1, b1, W2, b2, ... is taken (pipeline1.py:497 has two hidden layers, main1.py:863 three)."""
        z = np.load(path, allow_pickle=False)
        Ws, bs = _npz_layers(z, prefix)
        return cls(z["mean"], z["scale"], Ws, bs, classes=z["classes"], dtype=dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-1] != self.n_features_in_:  # the guard of main1.py:976-981
            raise ValueError(f"Feature mismatch! Expected {self.n_features_in_}, got {x.shape[-1]}.")
        h = (x.to(self.mean.dtype) - self.mean) * self.inv_scale
        n = len(self.weights)
        for i in range(n):
            h = torch.addmm(self.biases[i], h, self.weights[i])
            if i + 1 < n:
                h = F.relu(h)
        return h

    @torch.no_grad()
    def predict_proba(self, x: torch.Tensor) -> torch.Tensor:
        return torch.softmax(self.forward(x), dim=-1)

    @torch.no_grad()
    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.forward(x), dim=-1)

    def inverse_transform(self, idx) -> list:
        """LabelEncoder.inverse_transform (pipeline1.py:462-465; classes are the sorted folder names)."""
        return [self.classes[int(i)] for i in idx]


class FusedMLPHead:
    """The same scaler + MLP as ``MLPHead`` evaluated by ONE HIP kernel behind the C ABI (``dysf_mlp_*``): rows in,
    logits / probabilities / arg-max out.  For batches that live on the GPU the eager module's seven small launches
    cost more than its arithmetic; this is the head bench.py and the fused pipeline use."""

    def __init__(self, mean, scale, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], classes=None,
                 device: int = 0):
        import ctypes
        from . import _lib
        self._lib = _lib
        self.device = device
        ws = [np.ascontiguousarray(w, np.float64) for w in weights]
        bs = [np.ascontiguousarray(b, np.float64) for b in biases]
        dims = np.array([ws[0].shape[0]] + [w.shape[1] for w in ws], np.int32)
        for i, w in enumerate(ws):
            if w.shape != (dims[i], dims[i + 1]) or bs[i].shape != (dims[i + 1],):
                raise ValueError("layer shapes do not chain")
        self.n_features_in_, self.n_outputs = int(dims[0]), int(dims[-1])
        self.classes = None if classes is None else [str(c) for c in classes]
        mean = np.ascontiguousarray(mean, np.float64)
        scale = np.ascontiguousarray(scale, np.float64)
        wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
        bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        rc = L.dysf_mlp_create(device, len(ws), dims.ctypes.data_as(ctypes.c_void_p), wp, bp,
                               mean.ctypes.data_as(ctypes.c_void_p), scale.ctypes.data_as(ctypes.c_void_p),
                               ctypes.byref(self._h))
        if rc != 0:
            raise _lib.DysfError(f"dysf_mlp_create failed ({rc}): {L.dysf_mlp_last_error().decode()}")

    @classmethod
    def from_npz(cls, path: str, device: int = 0, prefix: str = "") -> "FusedMLPHead":
        z = np.load(path, allow_pickle=False)
        Ws, bs = _npz_layers(z, prefix)
        return cls(z["mean"], z["scale"], Ws, bs, classes=z["classes"], device=device)

    def _run(self, x: torch.Tensor, logits=True, proba=False, pred=False):
        import ctypes
        if x.shape[-1] != self.n_features_in_:  # the guard of main1.py:976-981
            raise ValueError(f"Feature mismatch! Expected {self.n_features_in_}, got {x.shape[-1]}.")
        x = x.to(torch.float32).contiguous()
        assert x.is_cuda
        n = x.shape[0]
        mk = lambda on, dt, *s: torch.empty(s, dtype=dt, device=x.device) if on else None  # noqa: E731
        lo, pr, pd = mk(logits, torch.float32, n, self.n_outputs), mk(proba, torch.float32, n, self.n_outputs), mk(pred, torch.int32, n)
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
        L = self._lib.lib()
        rc = L.dysf_mlp_forward_dev(self._h, ptr(x), n, ptr(lo), ptr(pr), ptr(pd),
                                    ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise self._lib.DysfError(f"dysf_mlp_forward_dev failed ({rc}): {L.dysf_mlp_last_error().decode()}")
        return lo, pr, pd

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x)[0]

    def predict_proba(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, logits=False, proba=True)[1]

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, logits=False, pred=True)[2].to(torch.int64)

    def inverse_transform(self, idx) -> list:
        return [self.classes[int(i)] for i in idx]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lib().dysf_mlp_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RandomForestHead:
    """scaler.transform + RandomForestClassifier.predict / predict_proba (main1.py:987-989, model of pipeline1.py:496)
    on the GPU: a HIP tree-walk kernel behind the C ABI (``dysf_forest_*``).  Built from flat node arrays exported
    from a fitted scikit-learn forest (see tests/golden/make_golden.py: forest_fixture)."""

    def __init__(self, feature, threshold, left, right, value, root, mean, scale, classes=None, device: int = 0):
        import ctypes
        from . import _lib
        self._lib = _lib
        self.device = device
        self.n_classes = int(np.asarray(value).shape[1])
        self.n_features_in_ = int(np.asarray(mean).shape[0])
        self.classes = None if classes is None else [str(c) for c in classes]
        arrs = [np.ascontiguousarray(feature, np.int32), np.ascontiguousarray(threshold, np.float64),
                np.ascontiguousarray(left, np.int32), np.ascontiguousarray(right, np.int32),
                np.ascontiguousarray(value, np.float64), np.ascontiguousarray(root, np.int32),
                np.ascontiguousarray(mean, np.float64), np.ascontiguousarray(scale, np.float64)]
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        rc = L.dysf_forest_create(device, self.n_features_in_, self.n_classes, len(arrs[5]), len(arrs[0]),
                                  *[vp(a) for a in arrs], ctypes.byref(self._h))
        if rc != 0:
            raise _lib.DysfError(f"dysf_forest_create failed ({rc}): {L.dysf_forest_last_error().decode()}")

    @classmethod
    def from_npz(cls, path: str, classes=None, device: int = 0, prefix: str = "") -> "RandomForestHead":
        z = np.load(path, allow_pickle=False)
        g = lambda k: z[prefix + k]  # noqa: E731
        return cls(g("feature"), g("threshold"), g("left"), g("right"), g("value"), g("root"), z["mean"], z["scale"],
                   classes=classes, device=device)

    def _run(self, x: torch.Tensor):
        import ctypes
        if x.shape[-1] != self.n_features_in_:
            raise ValueError(f"Feature mismatch! Expected {self.n_features_in_}, got {x.shape[-1]}.")
        x = x.to(torch.float32).contiguous()
        assert x.is_cuda
        n = x.shape[0]
        proba = torch.empty((n, self.n_classes), dtype=torch.float64, device=x.device)
        pred = torch.empty(n, dtype=torch.int32, device=x.device)
        L = self._lib.lib()
        rc = L.dysf_forest_predict_dev(self._h, ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(proba.data_ptr()),
                                       ctypes.c_void_p(pred.data_ptr()),
                                       ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise self._lib.DysfError(f"dysf_forest_predict_dev failed ({rc}): {L.dysf_forest_last_error().decode()}")
        return proba, pred

    def predict_proba(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x)[0]

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x)[1].to(torch.int64)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lib().dysf_forest_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SVCHead:
    """scaler.transform + SVC(kernel='rbf', probability=True).predict / predict_proba / decision_function(ovo)
    (pipeline1.py:498, 512-514; main1.py:865) on the GPU: a HIP kernel behind the C ABI (``dysf_svc_*``) that follows
    libsvm's prediction algorithm.  Built from the arrays scikit-learn exposes on a fitted SVC (see
    tests/golden/make_golden.py: svc_fixture)."""

    def __init__(self, support_vectors, dual_coef, intercept, n_support, gamma, mean, scale, prob_a=None, prob_b=None,
                 classes=None, device: int = 0):
        import ctypes
        from . import _lib
        self._lib = _lib
        self.device = device
        sv = np.ascontiguousarray(support_vectors, np.float64)
        self.n_classes = int(np.asarray(n_support).shape[0])
        self.n_pairs = self.n_classes * (self.n_classes - 1) // 2
        self.n_features_in_ = int(sv.shape[1])
        self.has_proba = prob_a is not None and len(np.atleast_1d(prob_a)) > 0
        self.classes = None if classes is None else [str(c) for c in classes]
        vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        arrs = dict(sv=sv, dc=np.ascontiguousarray(dual_coef, np.float64), ic=np.ascontiguousarray(intercept, np.float64),
                    ns=np.ascontiguousarray(n_support, np.int32),
                    pa=np.ascontiguousarray(prob_a, np.float64) if self.has_proba else None,
                    pb=np.ascontiguousarray(prob_b, np.float64) if self.has_proba else None,
                    mean=np.ascontiguousarray(mean, np.float64), scale=np.ascontiguousarray(scale, np.float64))
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        rc = L.dysf_svc_create(device, self.n_features_in_, self.n_classes, sv.shape[0], vp(arrs["sv"]), vp(arrs["dc"]),
                               vp(arrs["ic"]), vp(arrs["ns"]), vp(arrs["pa"]), vp(arrs["pb"]), float(gamma),
                               vp(arrs["mean"]), vp(arrs["scale"]), ctypes.byref(self._h))
        if rc != 0:
            raise _lib.DysfError(f"dysf_svc_create failed ({rc}): {L.dysf_svc_last_error().decode()}")

    @classmethod
    def from_npz(cls, path: str, classes=None, device: int = 0, prefix: str = "") -> "SVCHead":
        z = np.load(path, allow_pickle=False)
        g = lambda k: z[prefix + k]  # noqa: E731
        return cls(g("support_vectors"), g("dual_coef"), g("intercept"), g("n_support"), float(g("gamma")), z["mean"],
                   z["scale"], g("prob_a"), g("prob_b"), classes=classes, device=device)

    def _run(self, x: torch.Tensor, want_proba: bool = True):
        import ctypes
        if x.shape[-1] != self.n_features_in_:
            raise ValueError(f"Feature mismatch! Expected {self.n_features_in_}, got {x.shape[-1]}.")
        x = x.to(torch.float32).contiguous()
        assert x.is_cuda
        n = x.shape[0]
        dec = torch.empty((n, self.n_pairs), dtype=torch.float64, device=x.device)
        proba = torch.empty((n, self.n_classes), dtype=torch.float64, device=x.device) if want_proba and self.has_proba else None
        pred = torch.empty(n, dtype=torch.int32, device=x.device)
        L = self._lib.lib()
        rc = L.dysf_svc_predict_dev(self._h, ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(dec.data_ptr()),
                                    ctypes.c_void_p(proba.data_ptr()) if proba is not None else None,
                                    ctypes.c_void_p(pred.data_ptr()),
                                    ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise self._lib.DysfError(f"dysf_svc_predict_dev failed ({rc}): {L.dysf_svc_last_error().decode()}")
        return dec, proba, pred

    def decision_function(self, x: torch.Tensor) -> torch.Tensor:
        """Pair decision values in one-vs-one order (decision_function_shape='ovo')."""
        return self._run(x, want_proba=False)[0]

    def predict_proba(self, x: torch.Tensor) -> torch.Tensor:
        if not self.has_proba:
            raise AttributeError("predict_proba is not available when probability=False")  # sklearn's own message
        return self._run(x)[1]

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, want_proba=False)[2].to(torch.int64)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lib().dysf_svc_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SoftVotingHead:
    """VotingClassifier(voting='soft') of main1.py:867: the arithmetic mean of the members' predict_proba, arg-max
    (first maximum).  Members are heads of this module; they all take the same UNscaled float32 rows."""

    def __init__(self, members, classes=None):
        self.members = list(members)
        self.classes = None if classes is None else [str(c) for c in classes]

    def predict_proba(self, x: torch.Tensor) -> torch.Tensor:
        ps = [m.predict_proba(x).to(torch.float64) for m in self.members]
        return torch.stack(ps).mean(0)

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return self.predict_proba(x).argmax(-1)


class CNNHead(nn.Module):
    """Conv2d(1->32,3x3,p1) ReLU MaxPool2 Conv2d(32->64,3x3,p1) ReLU global-avg-pool Linear(64->3) on
    [B, 1, 3*n_mfcc, T] frame-level features; random weights from ``torch.manual_seed(seed)``."""

    def __init__(self, n_classes: int = 3, seed: int = 0):
        super().__init__()
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.conv1 = nn.Conv2d(1, 32, 3, padding=1)
        self.conv2 = nn.Conv2d(32, 64, 3, padding=1)
        self.fc = nn.Linear(64, n_classes)
        torch.random.set_rng_state(g)

    def forward(self, frames: torch.Tensor) -> torch.Tensor:
        x = frames.unsqueeze(1) if frames.dim() == 3 else frames
        x = F.max_pool2d(F.relu(self.conv1(x)), 2)
        x = F.relu(self.conv2(x))
        return self.fc(x.mean(dim=(2, 3)))


class FusedCNNHead:
    """``CNNHead`` evaluated by hand-written HIP kernels behind the C ABI (``dysf_cnn_*``): conv1 + ReLU + max-pool in
    float32, conv2 as an implicit GEMM on the MFMA, pooled sums, Linear.  ``precision="f16"`` (default; BASELINE
    configs[4]) runs conv2 on the float16 MFMA with float32 accumulation; ``precision="f32"`` runs it on the float32 MFMA
    (configs[2]: arg-max equal to the float32 CPU module).  Takes the frame-level tensor [n, rows, T] exactly as
    ``extract_batch_device(frames=True)`` returns it (no copy, any row pitch); ``cols`` may be a per-clip int32 tensor of
    valid widths (ragged batches).  One C handle (= one partial-sum workspace) per HIP stream the head is used on."""

    def __init__(self, module: "CNNHead", device: int = 0, precision: str = "f16"):
        from . import _lib
        self._lib = _lib
        self.device = device
        if precision not in ("f16", "f32"):
            raise ValueError("precision must be 'f16' or 'f32'")
        self.precision = precision
        f = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), np.float32)  # noqa: E731
        self._arrs = [f(module.conv1.weight), f(module.conv1.bias), f(module.conv2.weight), f(module.conv2.bias),
                      f(module.fc.weight), f(module.fc.bias)]
        a = self._arrs
        if a[0].shape != (32, 1, 3, 3) or a[2].shape != (64, 32, 3, 3) or a[4].shape[1] != 64:
            raise ValueError("FusedCNNHead implements Conv(1->32,3x3) / Conv(32->64,3x3) / Linear(64->k) only")
        self.n_classes = int(a[4].shape[0])
        self._handles = {}
        self._handle(0)

    def _handle(self, stream_id: int):
        import ctypes
        h = self._handles.get(stream_id)
        if h is None:
            h = ctypes.c_void_p()
            L = self._lib.lib()
            rc = L.dysf_cnn_create(self.device, self.n_classes, self._lib.CNN_F32 if self.precision == "f32" else self._lib.CNN_F16,
                                   *[a.ctypes.data_as(ctypes.c_void_p) for a in self._arrs], ctypes.byref(h))
            if rc != 0:
                raise self._lib.DysfError(f"dysf_cnn_create failed ({rc}): {L.dysf_cnn_last_error().decode()}")
            self._handles[stream_id] = h
        return h

    def __call__(self, frames: torch.Tensor, cols=None) -> torch.Tensor:
        """frames: [n, rows, pitch] float32 CUDA tensor (last dimension contiguous); ``cols``: valid columns -- an int
        for the whole batch (default: all), or an int32 CUDA tensor [n] of per-clip widths."""
        import ctypes
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.dim() == 3 and frames.stride(2) == 1
        assert frames.stride(0) == frames.shape[1] * frames.stride(1)
        n, rows, pitch = frames.shape[0], frames.shape[1], frames.stride(1)
        d_cols = None
        if torch.is_tensor(cols):
            assert cols.is_cuda and cols.dtype == torch.int32 and cols.is_contiguous() and cols.numel() == n
            d_cols, cols = ctypes.c_void_p(cols.data_ptr()), int(frames.shape[2])
        cols = int(frames.shape[2] if cols is None else cols)
        out = torch.empty((n, self.n_classes), dtype=torch.float32, device=frames.device)
        L = self._lib.lib()
        stream = torch.cuda.current_stream(frames.device)
        rc = L.dysf_cnn_forward_dev(self._handle(stream.cuda_stream), ctypes.c_void_p(frames.data_ptr()), n, rows, cols, d_cols,
                                    pitch, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise self._lib.DysfError(f"dysf_cnn_forward_dev failed ({rc}): {L.dysf_cnn_last_error().decode()}")
        return out

    def close(self):
        for h in getattr(self, "_handles", {}).values():
            if h.value:
                self._lib.lib().dysf_cnn_destroy(h)
        self._handles = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
