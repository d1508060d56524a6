"""Host-side ingest for the feature path: PCM_16 WAV files / host int16 buffers ->
pinned staging -> double-buffered H2D copies overlapped with the feature kernels -> pooled rows back on the host.

This is the step just before the hot path (``load_audio``, pipeline1.py:100-106, then the per-file loop p1:447-453).
Decoding other containers and resampling stay out of scope (no decoder here, soxr parity unpinnable).  torch is
plumbing only: pinned memory, streams, events."""
import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np


def read_wav_pcm16(path: str) -> Tuple[np.ndarray, int]:
    """Mono PCM_16 RIFF/WAVE -> (int16 samples, sample rate); raises ValueError on anything else."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        cid, size = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", blob, pos + 8)
        elif cid == b"data":
            data = blob[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or fmt[0] != 1 or fmt[1] != 1 or fmt[5] != 16:
        raise ValueError("only mono PCM_16 is supported on this path")
    return np.frombuffer(data[: len(data) // 2 * 2], dtype="<i2"), int(fmt[2])


def pack_wavs(paths: Sequence[str], sr: int) -> Tuple[np.ndarray, np.ndarray, List[bool]]:
    """Files -> (flat int16, offsets[n+1], ok flags).  Unreadable / wrong-rate files become empty clips, which the
    kernel reports as DYSF_EMPTY and turns into zero rows (the reference's load failure -> zeros convention)."""
    clips, ok = [], []
    for p in paths:
        try:
            pcm, fsr = read_wav_pcm16(p)
            if fsr != sr:
                raise ValueError("sample rate mismatch")
            clips.append(pcm); ok.append(True)
        except (OSError, ValueError):
            clips.append(np.zeros(0, np.int16)); ok.append(False)
    offsets = np.zeros(len(clips) + 1, np.int64)
    offsets[1:] = np.cumsum([len(c) for c in clips])
    flat = np.concatenate(clips) if clips else np.zeros(0, np.int16)
    return flat, offsets, ok


def pinned_empty(n: int, dtype=np.int16) -> np.ndarray:
    """A page-locked host array (torch's pinned allocator = hipHostMalloc): what a reader should decode PCM into, so that
    the H2D copies of ``StreamingExtractor`` start straight from it."""
    import torch
    t = torch.empty(int(n), dtype={np.dtype(np.int16): torch.int16, np.dtype(np.float32): torch.float32}[np.dtype(dtype)]).pin_memory()
    return t.numpy()   # the ndarray (and every view of it) holds a reference to the tensor: the page-locked memory is
                       # released when the last of them goes away


class StreamingExtractor:
    """Chunks of clips flow host -> device -> host: the H2D copy of chunk i+1 runs on a copy stream while the kernels of
    chunk i run on the compute stream, into two device buffers.  The copies start from the CALLER's buffer: it is used in
    place if it is page-locked already (``pinned_empty``) or is page-locked for the duration of the call
    (hipHostRegister; ``register=False`` or a failed registration falls back to two pinned staging buffers filled by
    ``copy_threads`` host threads, which caps the rate at the host's memcpy speed)."""

    def __init__(self, extractor, chunk_clips: int = 1024, copy_threads: int = 4, register: bool = True):
        self.ex = extractor
        self.chunk = int(chunk_clips)
        self.copy_threads = max(1, int(copy_threads))   # host threads that fill the pinned staging buffer
        self.register = bool(register)
        self.path = ""
        self.h2d_gbs = None

    def describe(self) -> str:
        return self.path

    def _stage_copy(self, pool, dst: np.ndarray, src: np.ndarray):
        """dst[:] = src with the work cut into slices for the thread pool (NumPy's copy releases the GIL); one thread
        moves ~10 GB/s, which would otherwise cap the whole ingest far below the PCIe rate."""
        n = len(src)
        k = self.copy_threads if n >= (1 << 20) else 1
        if k == 1 or pool is None:
            dst[:n] = src
            return
        step = (n + k - 1) // k
        futs = [pool.submit(np.copyto, dst[i:min(n, i + step)], src[i:min(n, i + step)]) for i in range(0, n, step)]
        for f in futs:
            f.result()

    @staticmethod
    def _page_locked_view(pcm: np.ndarray, allow_register: bool):
        """(torch view of pcm that H2D copies may start from, registered?) or (None, False)."""
        import torch
        if not pcm.flags["C_CONTIGUOUS"] or not pcm.flags["WRITEABLE"]:
            return None, False
        t = torch.from_numpy(pcm)
        if t.is_pinned():
            return t, False
        if allow_register and pcm.nbytes > 0:
            try:
                rc = torch.cuda.cudart().cudaHostRegister(pcm.ctypes.data, pcm.nbytes, 0)
            except Exception:      # a runtime without the call: stage through pinned buffers instead
                return None, False
            if int(rc) == 0:
                return t, True
        return None, False

    def run(self, pcm: np.ndarray, offsets: np.ndarray) -> Dict[str, np.ndarray]:
        import time
        import torch
        from concurrent.futures import ThreadPoolExecutor
        assert pcm.dtype == np.int16 and offsets.dtype == np.int64
        dev = torch.device("cuda", self.ex.device)
        n = len(offsets) - 1
        bounds = [(i, min(n, i + self.chunk)) for i in range(0, n, self.chunk)]
        pooled = torch.empty((n, self.ex.pooled_dim), dtype=torch.float32).pin_memory()
        status = torch.empty(n, dtype=torch.int32).pin_memory()
        if n == 0:
            return {"pooled": pooled.numpy(), "status": status.numpy()}
        torch.cuda.set_device(dev)
        direct, registered = self._page_locked_view(pcm, self.register)
        self.path = ("H2D straight from the caller's buffer (" + ("page-locked by hipHostRegister for the call" if registered
                     else "already page-locked") + "), double-buffered on the device") if direct is not None else \
            f"pinned staging buffers filled by {self.copy_threads} host threads, double-buffered H2D"
        pool = ThreadPoolExecutor(self.copy_threads) if (direct is None and self.copy_threads > 1) else None
        max_chunk = max(int(offsets[b] - offsets[a]) for a, b in bounds)
        max_chunk = (max_chunk + 7) // 8 * 8 + 8
        host = [torch.empty(max_chunk, dtype=torch.int16).pin_memory() for _ in range(2)] if direct is None else None
        dpcm = [torch.empty(max_chunk, dtype=torch.int16, device=dev) for _ in range(2)]
        doff = [torch.empty(self.chunk + 1, dtype=torch.int64, device=dev) for _ in range(2)]
        rel_all = torch.empty(n + len(bounds), dtype=torch.int64).pin_memory()   # every chunk's relative offsets, pinned
        copy_s, comp_s = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
        ready = [torch.cuda.Event() for _ in range(2)]   # H2D of slot done
        free = [torch.cuda.Event() for _ in range(2)]    # kernels reading slot done
        host_free = [None, None]
        h2d_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        h2d_bytes = 0

        def stage(ci):
            nonlocal h2d_bytes
            a, b = bounds[ci]
            s = ci & 1
            lo, hi = int(offsets[a]), int(offsets[b])
            if direct is None:
                if host_free[s] is not None:
                    host_free[s].synchronize()            # the previous H2D out of this pinned buffer has finished
                self._stage_copy(pool, host[s][: hi - lo].numpy(), pcm[lo:hi])
                src = host[s][: hi - lo]
            else:
                src = direct[lo:hi]
            rel = rel_all[a + ci: b + ci + 1]
            rel.copy_(torch.from_numpy(offsets[a:b + 1] - lo))
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(free[s])            # the kernels of the chunk that used this slot are done
                if ci == 0:
                    h2d_ev[0].record(copy_s)
                dpcm[s][: hi - lo].copy_(src, non_blocking=True)
                doff[s][: b - a + 1].copy_(rel, non_blocking=True)
                ready[s].record(copy_s)
                if direct is None:
                    host_free[s] = torch.cuda.Event(); host_free[s].record(copy_s)
                if ci == len(bounds) - 1:
                    h2d_ev[1].record(copy_s)
            h2d_bytes += (hi - lo) * 2

        try:
            for s in range(2):
                free[s].record(comp_s)
            stage(0)
            for ci, (a, b) in enumerate(bounds):
                s = ci & 1
                if ci + 1 < len(bounds):
                    stage(ci + 1)
                comp_s.wait_event(ready[s])
                ms = int(np.max(np.diff(offsets[a:b + 1]))) if b > a else 0
                res = self.ex.extract_batch_device(dpcm[s], doff[s][: b - a + 1], ms, pooled=True, frames=False)
                pooled[a:b].copy_(res["pooled"], non_blocking=True)
                status[a:b].copy_(res["status"], non_blocking=True)
                free[s].record(comp_s)
            torch.cuda.synchronize(dev)
            self.h2d_gbs = h2d_bytes / (h2d_ev[0].elapsed_time(h2d_ev[1]) * 1e-3) / 1e9 if len(bounds) > 1 else None
        finally:
            if registered:
                torch.cuda.synchronize(dev)
                try:
                    torch.cuda.cudart().cudaHostUnregister(pcm.ctypes.data)
                except Exception:
                    pass
            if pool is not None:
                pool.shutdown()
        return {"pooled": pooled.numpy(), "status": status.numpy()}
