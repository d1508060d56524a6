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


class StreamingExtractor:
    """Chunks of clips flow host -> device -> host through two pinned staging buffers and two device buffers:
    the H2D copy of chunk i+1 runs on a copy stream while the kernels of chunk i run on the compute stream."""

    def __init__(self, extractor, chunk_clips: int = 1024, copy_threads: int = 4):
        self.ex = extractor
        self.chunk = int(chunk_clips)
        self.copy_threads = max(1, int(copy_threads))   # host threads that fill the pinned staging buffer

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

    def run(self, pcm: np.ndarray, offsets: np.ndarray) -> Dict[str, np.ndarray]:
        import torch
        from concurrent.futures import ThreadPoolExecutor
        assert pcm.dtype == np.int16 and offsets.dtype == np.int64
        pool = ThreadPoolExecutor(self.copy_threads) if self.copy_threads > 1 else None
        dev = torch.device("cuda", self.ex.device)
        n = len(offsets) - 1
        bounds = [(i, min(n, i + self.chunk)) for i in range(0, n, self.chunk)]
        pooled = torch.empty((n, self.ex.pooled_dim), dtype=torch.float32).pin_memory()
        status = torch.empty(n, dtype=torch.int32).pin_memory()
        if n == 0:
            return {"pooled": pooled.numpy(), "status": status.numpy()}
        max_chunk = max(int(offsets[b] - offsets[a]) for a, b in bounds)
        max_chunk = (max_chunk + 7) // 8 * 8 + 8
        host = [torch.empty(max_chunk, dtype=torch.int16).pin_memory() for _ in range(2)]
        dpcm = [torch.empty(max_chunk, dtype=torch.int16, device=dev) for _ in range(2)]
        doff = [torch.empty(self.chunk + 1, dtype=torch.int64, device=dev) for _ in range(2)]
        copy_s, comp_s = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
        ready = [torch.cuda.Event() for _ in range(2)]   # H2D of slot done
        free = [torch.cuda.Event() for _ in range(2)]    # kernels reading slot done
        host_free = [None, None]

        def stage(ci):
            a, b = bounds[ci]
            s = ci & 1
            lo, hi = int(offsets[a]), int(offsets[b])
            if host_free[s] is not None:
                host_free[s].synchronize()            # the previous H2D out of this pinned buffer has finished
            self._stage_copy(pool, host[s][: hi - lo].numpy(), pcm[lo:hi])
            rel = torch.from_numpy((offsets[a:b + 1] - lo).astype(np.int64))
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(free[s])            # the kernels of the chunk that used this slot are done
                dpcm[s][: hi - lo].copy_(host[s][: hi - lo], non_blocking=True)
                doff[s][: b - a + 1].copy_(rel, non_blocking=True)
                ready[s].record(copy_s)
                host_free[s] = torch.cuda.Event(); host_free[s].record(copy_s)

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
        if pool is not None:
            pool.shutdown()
        return {"pooled": pooled.numpy(), "status": status.numpy()}
