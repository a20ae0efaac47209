"""Thin host layer over the C ABI: holds tensors in PyTorch-ROCm, passes raw
device pointers + the current HIP stream to libdlc_hip.so.  PyTorch is used for
device memory, streams and torch.distributed only; every computation below
runs in the hand-written HIP kernels.
"""
import collections
import contextlib
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib as L

_TORCH_TO_DLC = {torch.bfloat16: L.DLC_BF16, torch.float16: L.DLC_F16, torch.float32: L.DLC_F32,
                 torch.float64: L.DLC_F64, torch.int8: L.DLC_I8}
_NAME_TO_TORCH = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16,
                  "float16": torch.float16, "f32": torch.float32, "float32": torch.float32, "f64": torch.float64,
                  "float64": torch.float64}

SCRATCH_BYTES = 256 << 20   # default split-K scratch per engine (Engine.set_scratch)
K_STEP = 64   # the score GEMM's K step: stored descriptor rows are padded to a multiple of it


TopK = collections.namedtuple("TopK", "scores idx scores_f64 status")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def torch_dtype(name):
    if isinstance(name, torch.dtype):
        return name
    try:
        return _NAME_TO_TORCH[str(name).lower()]
    except KeyError:
        raise ValueError("unknown dtype %r" % (name,))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    """One engine per device / rank (wraps a dlc_ctx)."""

    def __init__(self, device=None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("deeploopcloser_amd needs a visible MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self._dev_index = self.device.index
        ctx = C.c_void_p()
        rc = self.lib.dlc_create(self.device.index, C.byref(ctx))
        if rc != L.DLC_OK:
            raise L.DlcError("dlc_create(%d) failed: %s" % (self.device.index, self.lib.dlc_status_string(rc).decode()))
        self.ctx = ctx
        self._ws = {}
        self._scratch = None
        # The copy streams of run_chunked / for_each_chunk, created NOW and touched once: the HIP runtime deals its few
        # hardware queues (4 by default) to streams in the order they come to life, and two streams on one queue run one
        # after the other.  Created lazily -- behind whatever streams the application had made by then (a MatchPipeline per
        # database, ...) -- an upload stream landed on the compute stream's queue and upload, kernels and download of
        # SDAV.transform(ndarray) stopped overlapping: 47 ms instead of 36 (docs/LAB.md 11.6; GPU_MAX_HW_QUEUES=8 hid it).
        self._s_in, self._s_out = torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)
        # ONE second stream per engine, shared by every MatchPipeline of the engine and by SDAV.train_steps' mask draws: the
        # default four hardware queues are taken (caller's stream, upload, download, this one) and a fifth stream would
        # share a queue with one of them.  Users of it are ordered among themselves (two pipelines' select / exchange /
        # merge stages run one behind the other; results are the same, ordering is by events); an application that needs
        # two pipelines overlapping independently gives each its own Engine (dlc_create is cheap) -- one per database.
        self.side_stream = torch.cuda.Stream(device=self.device)
        for st in (self._s_in, self._s_out, self.side_stream):
            torch.cuda.Event().record(st)

    def set_scratch(self, nbytes=SCRATCH_BYTES):
        """Latency mode: lend the context `nbytes` of HBM for the split-K form of the dense GEMMs
        (a single frame: SDAV layers with 30 rows, conv3-5 with 130 output pixels -- 5-8x lower
        encode latency); 0 turns it off.  OFF by default: one-pass GEMMs make an encode bit-identical
        whatever the batch it is part of, split-K changes the summation order (by ~1e-16 relative)."""
        torch.cuda.synchronize(self.device)         # nothing in flight may still use the old buffer
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device) if nbytes else None
        self._check(self.lib.dlc_set_scratch(self.ctx, _ptr(t) if t is not None else None, int(nbytes)))
        self._scratch = t              # keeps the previous buffer alive until the context has the new one

    @contextlib.contextmanager
    def latency_mode(self, nbytes=SCRATCH_BYTES):
        """`with engine.latency_mode():` -- split-K scratch on inside the block (no-op if already on)."""
        if self._scratch is not None:
            yield
            return
        self.set_scratch(nbytes)
        try:
            yield
        finally:
            self.set_scratch(0)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.dlc_destroy(self.ctx)
            self.ctx = None
            self._scratch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -------------------------------------------------------------
    def _check(self, rc):
        L.raise_for_status(self.lib, self.ctx, rc)

    def _raw_stream(self):
        """The current stream's handle as an int (torch._C._cuda_getCurrentRawStream: a tenth of the cost of building a
        torch.cuda.Stream object, which three calls per batch of a streaming detector made a quarter of its host time)."""
        if _RAW_STREAM is not None:
            return _RAW_STREAM(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream      # (a torch build without the private fast path)

    def _stream(self):
        return C.c_void_p(self._raw_stream())

    WORKSPACE_STREAMS = 4      # per name: the workspaces of at most this many streams are kept (least recently used out)

    def workspace(self, key, nbytes):
        """Named scratch tensor of at least nbytes, one per (name, current stream): calls on different streams must not
        share a workspace (a match keeps its group maxima there between its kernels).  A caller that rotates streams
        does not grow memory without bound: per name the WORKSPACE_STREAMS most recently used streams keep theirs, an
        evicted one is released once its stream has finished with it (the caching allocator's record_stream)."""
        nbytes = max(int(nbytes), 256)
        k = (key, self._raw_stream())
        t = self._ws.pop(k, None)
        if t is None or t.numel() < nbytes:
            t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._ws[k] = t                                   # (re-)inserted last: dicts keep insertion order
        if len(self._ws) <= self.WORKSPACE_STREAMS:       # (cannot hold more than that many of one name)
            return t
        same = [kk for kk in self._ws if kk[0] == key]
        for kk in same[:max(0, len(same) - self.WORKSPACE_STREAMS)]:
            old = self._ws.pop(kk)
            old.record_stream(torch.cuda.ExternalStream(kk[1], device=self.device) if kk[1] else torch.cuda.default_stream(self.device))
        return t

    def _check_out(self, name, t, shape, dtype):
        """Caller-supplied buffers reach the kernels as raw pointers: refuse anything that is not exactly
        the contiguous tensor the kernel will write."""
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dtype or \
                not t.is_contiguous() or t.device != self.device:
            raise ValueError("%s must be a contiguous %s tensor of shape %s on %s" % (name, dtype, tuple(shape), self.device))
        return t

    @staticmethod
    def _wrote(t):
        """The library has written (or is about to write) t through its raw pointer: torch must see that as the in-place
        write it is -- the version counter t shares with its base and its views moves, and whatever was derived from the
        old contents and keyed on the version (the unit_rows() mark) is outdated, as after any torch-side write."""
        torch.autograd.graph.increment_version(t)

    def to_device(self, x, dtype=None):
        if isinstance(x, torch.Tensor):
            t = x.to(self.device)
            if dtype is not None and t.dtype != dtype:
                t = t.to(dtype)
            return t.contiguous()
        a = np.ascontiguousarray(np.asarray(x))
        t = self.upload(a) if a.nbytes >= self.STAGED_MIN_BYTES else None
        if t is None:
            if not a.flags.writeable:
                a = a.copy()
            t = torch.from_numpy(a).to(self.device)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t

    # ---- pageable host arrays <-> HBM through the context's pinned staging ring (dlc_host_to_device / dlc_device_to_host)
    STAGED_MIN_BYTES = 4 << 20

    @staticmethod
    def _torch_dtype_of(a):
        try:
            return torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype
        except TypeError:
            return None

    def upload(self, a, out=None, stream=None):
        """C-contiguous numpy array -> device tensor of the same shape / dtype (None if torch has no such dtype).
        Returns when `a` has been consumed; the tensor is ready in `stream` order (default: the current stream)."""
        dt = self._torch_dtype_of(a)
        if dt is None or not a.flags.c_contiguous:
            return None
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        if out is None:
            with torch.cuda.stream(st):
                out = torch.empty(a.shape, dtype=dt, device=self.device)
        else:
            self._wrote(out)
        if a.nbytes:
            self._check(self.lib.dlc_host_to_device(self.ctx, C.c_void_p(out.data_ptr()), C.c_void_p(a.ctypes.data), a.nbytes,
                                                     C.c_void_p(st.cuda_stream)))
        return out

    # Result arrays handed to the caller.  A fresh pageable array costs the kernel a page fault and a zeroed page per
    # 4 KiB (638 MB of SDAV descriptors: 24 ms on top of a 13 ms copy), so large results live in page-locked blocks of
    # torch's caching host allocator instead: an ordinary ndarray for the caller (its base keeps the block), filled by
    # the DMA engine directly -- no staging copy -- and recycled once the caller drops it.  At most PINNED_RESULT_CAP
    # bytes of such results are alive at a time; past that (a caller that keeps everything) results are pageable again.
    PINNED_RESULT_CAP = 4 << 30
    _pinned_live = 0

    def _release_pinned(self, nbytes):
        self._pinned_live -= nbytes

    def result_array(self, shape, torch_dt):
        """(ndarray, pinned uint8 tensor behind it or None) for a result of `shape` / torch dtype."""
        np_dt = torch.empty(0, dtype=torch_dt).numpy().dtype
        nbytes = int(np.prod(shape, dtype=np.int64)) * np_dt.itemsize
        if nbytes >= self.STAGED_MIN_BYTES and self._pinned_live + nbytes <= self.PINNED_RESULT_CAP:
            try:
                t = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
            except RuntimeError:
                t = None
            if t is not None:
                self._pinned_live += nbytes
                weakref.finalize(t, self._release_pinned, nbytes)
                return t.numpy().view(np_dt).reshape(shape), t
        return np.empty(shape, dtype=np_dt), None

    def download(self, t, out=None, stream=None):
        """Device tensor -> numpy array (a fresh one, or `out`: C-contiguous, same bytes), read in `stream` order; blocks
        until the array is complete."""
        t = t.contiguous()
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        if out is None:
            out, pin = self.result_array(tuple(t.shape), t.dtype)
            if pin is not None:
                with torch.cuda.stream(st):
                    pin.copy_(t.reshape(-1).view(torch.uint8), non_blocking=True)
                st.synchronize()
                return out
        if out.nbytes != t.numel() * t.element_size() or not out.flags.c_contiguous:
            raise ValueError("download: out must be a C-contiguous array of %d bytes" % (t.numel() * t.element_size()))
        if out.nbytes:
            self._check(self.lib.dlc_device_to_host(self.ctx, C.c_void_p(out.ctypes.data), C.c_void_p(t.data_ptr()), out.nbytes,
                                                     C.c_void_p(st.cuda_stream)))
        return out

    def run_chunked(self, x, chunk, compute):
        """The reference's NumPy-in / NumPy-out contract without serialising on the link: x [N, ...] (numpy) is cut
        into chunks of `chunk` items; the upload of chunk c+1 (copy stream 1), compute(device chunk) -> device tensor
        of chunk c (the current stream) and the download of chunk c-1's result (copy stream 2, into the returned
        array) overlap.  compute must be batch-invariant (every kernel of this library is); results are concatenated
        along dimension 0."""
        x = np.ascontiguousarray(x)
        n = x.shape[0]
        dev = self.device
        dt = self._torch_dtype_of(x)
        if dt is None:
            raise ValueError("run_chunked: unsupported array dtype %s" % x.dtype)
        cur = torch.cuda.current_stream(dev)
        chunk = max(1, min(int(chunk), n))
        bufs = [torch.empty((chunk,) + tuple(x.shape[1:]), dtype=dt, device=dev) for _ in range(2 if n > chunk else 1)]
        self._s_in.wait_stream(cur)                 # the buffers' memory may still be in use by earlier work of this stream
        free_ev = [None, None]
        state = {"out": None, "pin": None, "row": 0}
        pend = None

        def flush(y, done, items):
            if state["out"] is None:
                per = y.shape[0] // items
                state["out"], state["pin"] = self.result_array((n * per,) + tuple(y.shape[1:]), y.dtype)
            self._s_out.wait_event(done)
            r0 = state["row"]
            if state["pin"] is not None:              # page-locked result: the DMA engine writes it, nobody waits
                row_bytes = y.numel() // max(1, y.shape[0]) * y.element_size()
                with torch.cuda.stream(self._s_out):
                    state["pin"][r0 * row_bytes:(r0 + y.shape[0]) * row_bytes].copy_(y.reshape(-1).view(torch.uint8), non_blocking=True)
                y.record_stream(self._s_out)
            else:
                self.download(y, out=state["out"][r0:r0 + y.shape[0]], stream=self._s_out)
            state["row"] = r0 + y.shape[0]

        for c, lo in enumerate(range(0, n, chunk)):
            hi = min(lo + chunk, n)
            b = c % len(bufs)
            if free_ev[b] is not None:
                self._s_in.wait_event(free_ev[b])     # chunk c-2's kernels have read this buffer
            self.upload(x[lo:hi], out=bufs[b][:hi - lo], stream=self._s_in)
            up = torch.cuda.Event()
            up.record(self._s_in)
            cur.wait_event(up)
            y = compute(bufs[b][:hi - lo]).contiguous()
            done = torch.cuda.Event()
            done.record(cur)
            free_ev[b] = done
            if pend is not None:
                flush(*pend)                          # blocks the host while the GPU works on chunk c
            pend = (y, done, hi - lo)
        if pend is not None:
            flush(*pend)
        cur.wait_stream(self._s_in)
        self._s_out.synchronize()                     # the result is complete in host memory
        return state["out"]

    def for_each_chunk(self, x, chunk, consume, first=None):
        """The upload half of run_chunked for results that STAY on the device: x [N, ...] (numpy, pageable) is cut into
        chunks of `chunk` items, the upload of chunk c+1 (copy stream, pinned staging ring) overlaps consume(device
        chunk, lo, hi) of chunk c on the current stream.  consume must be done with its chunk in stream order (it may keep
        results, not the chunk: the two buffers are reused).  first: items of the first chunk (default `chunk`) -- nothing
        overlaps ITS upload, so a short first chunk starts the kernels sooner.  Returns when everything has been
        SUBMITTED; the current stream is ordered behind the last upload."""
        x = np.ascontiguousarray(x)
        n = x.shape[0]
        dev = self.device
        dt = self._torch_dtype_of(x)
        if dt is None:
            raise ValueError("for_each_chunk: unsupported array dtype %s" % x.dtype)
        cur = torch.cuda.current_stream(dev)
        chunk = max(1, min(int(chunk), n))
        first = chunk if first is None else max(1, min(int(first), chunk))
        bufs = [torch.empty((chunk,) + tuple(x.shape[1:]), dtype=dt, device=dev) for _ in range(2 if n > first else 1)]
        self._s_in.wait_stream(cur)                 # the buffers' memory may still be in use by earlier work of this stream
        free_ev = [None, None]
        starts = [0] + list(range(first, n, chunk))
        for c, lo in enumerate(starts):
            hi = min(lo + (first if c == 0 else chunk), n)
            b = c % len(bufs)
            if free_ev[b] is not None:
                self._s_in.wait_event(free_ev[b])     # chunk c-2's kernels have read this buffer
            self.upload(x[lo:hi], out=bufs[b][:hi - lo], stream=self._s_in)
            up = torch.cuda.Event()
            up.record(self._s_in)
            cur.wait_event(up)
            consume(bufs[b][:hi - lo], lo, hi)
            free_ev[b] = torch.cuda.Event()
            free_ev[b].record(cur)
        for b_ in bufs:
            b_.record_stream(cur)

    # ---- dense layers -----------------------------------------------------------
    def gemm_bias_act(self, a, b, bias=None, act=L.DLC_ACT_NONE, blayout=L.DLC_B_KN, out=None):
        """out[M,N] = act(a[M,K] . b + bias) in a's dtype (float64 / float32)."""
        if a.dim() != 2 or b.dim() != 2:
            raise ValueError("gemm_bias_act: operands must be 2-D")
        if a.dtype not in (torch.float64, torch.float32) or b.dtype != a.dtype:
            raise ValueError("gemm_bias_act: float64 or float32 operands of one dtype")
        a, b = a.contiguous(), b.contiguous()
        m, k = a.shape
        n = b.shape[1] if blayout == L.DLC_B_KN else b.shape[0]
        kb = b.shape[0] if blayout == L.DLC_B_KN else b.shape[1]
        if kb != k:
            raise ValueError("gemm_bias_act: inner dimensions differ (%d vs %d)" % (k, kb))
        if bias is not None:
            bias = bias.contiguous()
            if bias.numel() != n or bias.dtype != a.dtype:
                raise ValueError("gemm_bias_act: bias must be [N] of the operand dtype")
        if out is None:
            out = torch.empty((m, n), dtype=a.dtype, device=self.device)
        elif not isinstance(out, torch.Tensor) or tuple(out.shape) != (m, n) or out.dtype != a.dtype or \
                out.stride(1) != 1 or out.stride(0) < n or out.device != self.device:
            raise ValueError("gemm_bias_act: out must be a [%d, %d] %s tensor with unit column stride on %s"
                             % (m, n, a.dtype, self.device))
        self._check(self.lib.dlc_gemm_bias_act(self.ctx, _TORCH_TO_DLC[a.dtype], blayout, act, m, n, k, _ptr(a),
                                                a.stride(0), _ptr(b), b.stride(0), _ptr(bias), _ptr(out),
                                                out.stride(0), self._stream()))
        return out

    def bias_act(self, a, bias=None, act=L.DLC_ACT_NONE):
        a = a.contiguous()
        a2 = a.reshape(-1, a.shape[-1])
        out = torch.empty_like(a2)
        if bias is not None:
            bias = bias.to(a.dtype).contiguous()
            if bias.numel() != a2.shape[1]:
                raise ValueError("bias_act: bias must have the row width")
        self._check(self.lib.dlc_bias_act(self.ctx, _TORCH_TO_DLC[a.dtype], act, a2.shape[0], a2.shape[1], _ptr(a2),
                                           a2.stride(0), _ptr(bias), _ptr(out), out.stride(0), self._stream()))
        return out.reshape(a.shape)

    def _encode_out(self, out, rows, cols, dt, what):
        """The caller's result tensor for an encode ([rows, cols], contiguous, on this device), or a new one."""
        if out is None:
            return torch.empty((rows, cols), dtype=dt, device=self.device)
        if out.dtype != dt or tuple(out.shape) != (rows, cols) or not out.is_contiguous() or out.device != self.device:
            raise ValueError("%s: out must be a contiguous %s [%d, %d] tensor on %s" % (what, dt, rows, cols, self.device))
        return out

    def sdav_encode(self, x2d, weights, biases, out=None):
        """sigmoid chain on x2d [rows, K0]; weights[l] is [dims[l], dims[l+1]].  out: where the result goes (a caller that
        collects the chunks of a sequence hands its slice: no copy behind the call)."""
        dt = x2d.dtype
        if dt not in (torch.float64, torch.float32):
            raise ValueError("sdav_encode: float64 or float32")
        x2d = x2d.contiguous()
        n_layers = len(weights)
        dims = [x2d.shape[1]] + [w.shape[1] for w in weights]
        for l, w in enumerate(weights):
            if w.dtype != dt or w.shape[0] != dims[l] or not w.is_contiguous():
                raise ValueError("sdav_encode: W[%d] must be a contiguous [%d, *] %s tensor" % (l, dims[l], dt))
        rows = x2d.shape[0]
        dims_c = (C.c_int64 * (n_layers + 1))(*dims)
        w_c = (C.c_void_p * n_layers)(*[w.data_ptr() for w in weights])
        b_c = (C.c_void_p * n_layers)(*[(b.data_ptr() if b is not None else 0) for b in biases])
        need = self.lib.dlc_sdav_encode_workspace_bytes(rows, dims_c, n_layers, _TORCH_TO_DLC[dt])
        ws = self.workspace("sdav", need)
        out = self._encode_out(out, rows, dims[-1], dt, "sdav_encode")
        self._check(self.lib.dlc_sdav_encode(self.ctx, _TORCH_TO_DLC[dt], rows, n_layers, dims_c, _ptr(x2d), w_c, b_c,
                                              _ptr(out), _ptr(ws), ws.numel(), self._stream()))
        return out

    def sdav_split_panels(self, weights):
        """The tolerance-mode encoder's prepared weights (dlc_sdav_split_prepare): fp64 weights [dims[l], dims[l+1]] -> one
        uint8 device tensor holding, per layer, the two transposed fp16 pieces of W 2^s and 2^s."""
        n_layers = len(weights)
        dims = [weights[0].shape[0]] + [w.shape[1] for w in weights]
        for l, w in enumerate(weights):
            if w.dtype != torch.float64 or w.shape[0] != dims[l] or not w.is_contiguous() or w.device != self.device:
                raise ValueError("sdav_split_panels: W[%d] must be a contiguous float64 [%d, *] tensor on %s" % (l, dims[l], self.device))
        dims_c = (C.c_int64 * (n_layers + 1))(*dims)
        need = self.lib.dlc_sdav_split_panels_bytes(n_layers, dims_c)
        panels = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        w_c = (C.c_void_p * n_layers)(*[w.data_ptr() for w in weights])
        self._check(self.lib.dlc_sdav_split_prepare(self.ctx, n_layers, dims_c, w_c, _ptr(panels), panels.numel(), self._stream()))
        return panels

    def sdav_encode_split(self, x2d, dims, panels, biases, out=None):
        """The sigmoid chain on x2d [rows, dims[0]] (fp64) in the tolerance mode (three fp16 MFMA products per layer,
        include/dlc.h: dlc_sdav_encode_split) -> fp64 [rows, dims[-1]]."""
        if x2d.dtype != torch.float64 or x2d.dim() != 2 or x2d.shape[1] != dims[0]:
            raise ValueError("sdav_encode_split: x must be float64 [rows, %d]" % dims[0])
        x2d = x2d.contiguous()
        n_layers = len(dims) - 1
        rows = x2d.shape[0]
        dims_c = (C.c_int64 * (n_layers + 1))(*dims)
        b_c = (C.c_void_p * n_layers)(*[(b.data_ptr() if b is not None else 0) for b in biases])
        need = self.lib.dlc_sdav_encode_split_workspace_bytes(rows, dims_c, n_layers)
        ws = self.workspace("sdav_split", need)
        out = self._encode_out(out, rows, dims[-1], torch.float64, "sdav_encode_split")
        self._check(self.lib.dlc_sdav_encode_split(self.ctx, rows, n_layers, dims_c, _ptr(x2d), _ptr(panels), b_c, _ptr(out),
                                                    _ptr(ws), ws.numel(), self._stream()))
        return out

    def train_workspace(self, layer, batch, patches, weights):
        """A workspace tensor of dlc_sdav_train_step's size for this shape (for a caller that keeps its own)."""
        n_layers = len(weights)
        dims = [weights[0].shape[0]] + [w.shape[1] for w in weights]
        dims_c = (C.c_int64 * (n_layers + 1))(*dims)
        need = self.lib.dlc_sdav_train_workspace_bytes(batch, patches, dims_c, n_layers, layer)
        if need == 0:
            raise ValueError("sdav_train_step: batch must be >= 2 frames, layer in range")
        return torch.empty(int(need), dtype=torch.uint8, device=self.device)

    def sdav_train_step(self, layer, x2d, batch, patches, masks, weights, b_enc, b_dec, sparse_level, sparse_penalty,
                        consecutive_penalty, learning_rate, loss_out=None, ws=None):
        """One in-place SGD step of `layer` (fp64 tensors on this device); loss_out: 4 doubles; ws: the caller's own
        workspace (train_workspace) instead of the engine's."""
        n_layers = len(weights)
        dims = [weights[0].shape[0]] + [w.shape[1] for w in weights]
        dims_c = (C.c_int64 * (n_layers + 1))(*dims)
        for t in [x2d] + list(masks[:layer + 1]) + list(weights[:layer + 1]) + list(b_enc[:layer + 1]) + [b_dec]:
            if t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("sdav_train_step: contiguous float64 tensors expected")
        m_c = (C.c_void_p * n_layers)(*[(masks[l].data_ptr() if l <= layer else 0) for l in range(n_layers)])
        w_c = (C.c_void_p * n_layers)(*[w.data_ptr() for w in weights])
        b_c = (C.c_void_p * n_layers)(*[b.data_ptr() for b in b_enc])
        need = self.lib.dlc_sdav_train_workspace_bytes(batch, patches, dims_c, n_layers, layer)
        if need == 0:
            raise ValueError("sdav_train_step: batch must be >= 2 frames, layer in range")
        if ws is None:
            ws = self.workspace("train", need)
        else:
            self._check_ws(ws)
        self._check(self.lib.dlc_sdav_train_step(self.ctx, layer, batch, patches, n_layers, dims_c, _ptr(x2d), m_c, w_c,
                                                  b_c, _ptr(b_dec), float(sparse_level), float(sparse_penalty),
                                                  float(consecutive_penalty), float(learning_rate), _ptr(loss_out),
                                                  _ptr(ws), ws.numel(), self._stream()))

    def random_mask(self, out, n_zeros, seed, counter):
        """Exactly n_zeros zeros among ones, uniformly placed, into the fp64 tensor `out` (dlc_random_mask_f64)."""
        if out.dtype != torch.float64 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("random_mask: contiguous float64 tensor on %s expected" % (self.device,))
        self._check(self.lib.dlc_random_mask_f64(self.ctx, _ptr(out), out.numel(), int(n_zeros), int(seed) & (2 ** 64 - 1),
                                                  int(counter) & (2 ** 64 - 1), self._stream()))
        return out

    # ---- SDAV patch front-end --------------------------------------------------------------
    def rgb_to_gray(self, rgb):
        rgb = rgb.contiguous()
        if rgb.dtype != torch.uint8 or rgb.shape[-1] != 3:
            raise ValueError("rgb_to_gray: uint8 [..., 3] expected")
        gray = torch.empty(rgb.shape[:-1], dtype=torch.uint8, device=self.device)
        self._check(self.lib.dlc_rgb_to_gray_u8(self.ctx, _ptr(rgb), gray.numel(), _ptr(gray), self._stream()))
        return gray

    def harris_keypoints(self, gray, n):
        """gray uint8 [F,H,W] -> (points int32 [F,n,2] as cv2-style (x = column, y = row), responses int64
        [F,n], counts int32 [F]); (-1,-1) / 0 past a frame's count.  The build's stand-in for SURF."""
        gray = gray.contiguous()
        if gray.dtype != torch.uint8 or gray.dim() != 3:
            raise ValueError("harris_keypoints: uint8 [F, H, W] expected")
        f, h, w = gray.shape
        need = self.lib.dlc_harris_keypoints_workspace_bytes(f, h, w)
        if need == 0 or n < 1:
            raise ValueError("harris_keypoints: needs n >= 1 and frames of at least 7x7 pixels")
        ws = self.workspace("harris", need)
        pts = torch.empty((f, n, 2), dtype=torch.int32, device=self.device)
        resp = torch.empty((f, n), dtype=torch.int64, device=self.device)
        cnt = torch.empty((f,), dtype=torch.int32, device=self.device)
        self._check(self.lib.dlc_harris_keypoints_u8(self.ctx, _ptr(gray), f, h, w, n, _ptr(pts), _ptr(resp), _ptr(cnt),
                                                      _ptr(ws), ws.numel(), self._stream()))
        return pts, resp, cnt

    def extract_patches(self, gray, key_points, patch_size, dtype=torch.float64):
        """gray uint8 [F,H,W], key_points int32 [F,P,2] -> [F,P,patch_size^2] pixel/255."""
        gray, key_points = gray.contiguous(), key_points.contiguous()
        f, h, w = gray.shape
        p = key_points.shape[1]
        out = torch.empty((f, p, patch_size * patch_size), dtype=dtype, device=self.device)
        self._check(self.lib.dlc_extract_patches(self.ctx, _ptr(gray), f, h, w, _ptr(key_points), p, patch_size,
                                                  _TORCH_TO_DLC[dtype], _ptr(out), self._stream()))
        return out

    # ---- CnnVtl pieces ---------------------------------------------------------------
    def im2col(self, x, kh, kw, stride, pad_top, pad_left, oh, ow):
        n, h, w, c = x.shape
        cols = torch.empty((n * oh * ow, kh * kw * c), dtype=torch.float64, device=self.device)
        self._check(self.lib.dlc_im2col_nhwc_f64(self.ctx, _ptr(x), n, h, w, c, kh, kw, stride, pad_top, pad_left, oh,
                                                  ow, _ptr(cols), self._stream()))
        return cols

    def conv2d(self, x, kernel2d, bias, kh, kw, stride, pad_top, pad_left, oh, ow, act, frame_keys=None):
        """NHWC fp64 convolution + bias + activation as an implicit GEMM (no im2col matrix).
        kernel2d is the HWIO kernel reshaped [kh*kw*c, cout].  frame_keys ([n, 2] int64 from
        frame_minmax_keys()): the minimum / maximum of every frame's outputs is folded into it on the way."""
        x = x.contiguous()
        n, h, w, c = x.shape
        cout = kernel2d.shape[1]
        out = torch.empty((n, oh, ow, cout), dtype=torch.float64, device=self.device)
        if frame_keys is None:
            self._check(self.lib.dlc_conv2d_nhwc_f64(self.ctx, _ptr(x), n, h, w, c, _ptr(kernel2d), _ptr(bias), kh, kw, cout,
                                                      stride, pad_top, pad_left, oh, ow, act, _ptr(out), self._stream()))
        else:
            self._check_keys(frame_keys, n)
            self._check(self.lib.dlc_conv2d_nhwc_f64_stats(self.ctx, _ptr(x), n, h, w, c, _ptr(kernel2d), _ptr(bias), kh, kw,
                                                            cout, stride, pad_top, pad_left, oh, ow, act, _ptr(out),
                                                            _ptr(frame_keys), self._stream()))
        return out

    def _check_keys(self, keys, n):
        if not isinstance(keys, torch.Tensor) or keys.dtype != torch.int64 or tuple(keys.shape) != (n, 2) or \
                not keys.is_contiguous() or keys.device != self.device:
            raise ValueError("frame keys must be a contiguous [%d, 2] int64 tensor on %s" % (n, self.device))

    def frame_minmax_keys(self, n):
        """[n, 2] ordered keys, initialised to 'nothing seen', for conv2d(frame_keys=) / quant_gather()."""
        keys = torch.empty((n, 2), dtype=torch.int64, device=self.device)
        self._check(self.lib.dlc_cnnvtl_frame_minmax_init(self.ctx, _ptr(keys), n, self._stream()))
        return keys

    def quant_gather(self, segments, columns, frame_keys):
        """minmax_quant_gather with the per-frame range taken from frame_keys (folded by the convolutions)."""
        n = segments[0].shape[0]
        self._check_keys(frame_keys, n)
        segs = [s.reshape(n, -1).contiguous() for s in segments]
        ptrs = (C.c_void_p * len(segs))(*[s.data_ptr() for s in segs])
        sizes = (C.c_int64 * len(segs))(*[s.shape[1] for s in segs])
        minmax = torch.empty((n, 2), dtype=torch.float64, device=self.device)
        out = torch.empty((n, columns.numel()), dtype=torch.int8, device=self.device)
        self._check(self.lib.dlc_quant_gather_i8(self.ctx, ptrs, sizes, len(segs), n, _ptr(columns), columns.numel(),
                                                  _ptr(frame_keys), _ptr(minmax), _ptr(out), self._stream()))
        return out

    def space_to_depth(self, x, s):
        """NHWC fp64 [n,h,w,c] -> [n, h/s, w/s, s*s*c] (block s; dlc_space_to_depth_nhwc_f64)."""
        x = x.contiguous()
        n, h, w, c = x.shape
        y = torch.empty((n, h // s, w // s, s * s * c), dtype=torch.float64, device=self.device)
        self._check(self.lib.dlc_space_to_depth_nhwc_f64(self.ctx, _ptr(x), n, h, w, c, s, _ptr(y), self._stream()))
        return y

    def maxpool3x3s2(self, x):
        n, h, w, c = x.shape
        y = torch.empty((n, (h - 3) // 2 + 1, (w - 3) // 2 + 1, c), dtype=torch.float64, device=self.device)
        self._check(self.lib.dlc_maxpool3x3s2_nhwc_f64(self.ctx, _ptr(x), n, h, w, c, _ptr(y), self._stream()))
        return y

    def minmax_quant_gather(self, segments, columns):
        n = segments[0].shape[0]
        segs = [s.reshape(n, -1).contiguous() for s in segments]
        ptrs = (C.c_void_p * len(segs))(*[s.data_ptr() for s in segs])
        sizes = (C.c_int64 * len(segs))(*[s.shape[1] for s in segs])
        minmax = torch.empty((n, 2), dtype=torch.float64, device=self.device)
        out = torch.empty((n, columns.numel()), dtype=torch.int8, device=self.device)
        self._check(self.lib.dlc_minmax_quant_gather_i8(self.ctx, ptrs, sizes, len(segs), n, _ptr(columns),
                                                         columns.numel(), _ptr(minmax), _ptr(out), self._stream()))
        return out

    # ---- reference-semantics match ---------------------------------------------------
    def distinctive_score(self, dataset, mu, sigma, with_range=False):
        """Distinctive score [H] of a dataset [..., H]; with_range=True: (score, range) where range (3 + 2 H int64 words on
        the device) is what the pass learned about the columns' extremes -- pass it to sdav_similarity_matrix(range=) for
        the SAME tensor and that call does not read the descriptors again to find them."""
        d2 = dataset.reshape(-1, dataset.shape[-1]).contiguous()
        score = torch.empty(d2.shape[1], dtype=torch.float64, device=self.device)
        rng = torch.empty(self.lib.dlc_sdav_range_words(d2.shape[1]), dtype=torch.int64, device=self.device) if with_range else None
        self._check(self.lib.dlc_sdav_distinctive_score(self.ctx, _ptr(d2), d2.shape[0], d2.shape[1], float(mu),
                                                         float(sigma), _ptr(score), _ptr(rng), self._stream()))
        return (score, rng) if with_range else score

    def sdav_similarity_matrix(self, desc, score, a=10.0, b=-10.0, want_int64=True, force_f64=False, no_host_sync=False,
                               chunk_bytes=0, stats=None, direct_pairs=None, range=None):
        """All-vs-all SDAV similarity of desc [N,P,H] (fp64) -> (out fp64 [N,N], out int64 [N,N] or None).
        force_f64: the fp64 Gram form instead of the int8 arg-min filter (same matrix); no_host_sync: never read the
        flag word back (graph-capturable; a dataset with NaN / inf then yields NaN and stats[1] = 1, and no sample decides
        whether the filter is worth running); chunk_bytes: bound of one product block (0 = 8 GiB); stats: int64 [2] device
        tensor ([0] arg-mins evaluated directly, [1] why the fp64 form ran instead: bit 0 NaN / inf, bit 1 the sample's
        verdict), direct_pairs: uint8 [N,N] device tensor marking the frame pairs with a directly evaluated arg-min
        (include/dlc.h)."""
        desc = desc.contiguous()
        n, p, h = desc.shape
        flags = (L.DLC_SIM_FORCE_F64 if force_f64 else 0) | (L.DLC_SIM_NO_HOST_SYNC if no_host_sync else 0)
        if stats is not None:
            self._check_out("stats", stats, (2,), torch.int64)
        if direct_pairs is not None:
            self._check_out("direct_pairs", direct_pairs, (n, n), torch.uint8)
        if range is not None:
            self._check_out("range", range, (self.lib.dlc_sdav_range_words(h),), torch.int64)
        out = torch.empty((n, n), dtype=torch.float64, device=self.device)
        out_i = torch.empty((n, n), dtype=torch.int64, device=self.device) if want_int64 else None
        need = self.lib.dlc_sdav_similarity_workspace_bytes(n, p, h, flags, int(chunk_bytes))
        ws = self.workspace("sim", need)
        self._check(self.lib.dlc_sdav_similarity_matrix(self.ctx, _ptr(desc), n, p, h, _ptr(score), float(a), float(b),
                                                         _ptr(out), _ptr(out_i), flags, int(chunk_bytes), _ptr(range),
                                                         _ptr(stats), _ptr(direct_pairs), _ptr(ws), ws.numel(), self._stream()))
        return out, out_i

    # ---- streaming similarity: one new frame against the resident older ones (dlc_sdav_stream_*) ----
    def sdav_stream_state(self, capacity, p, h, lo=0.0, hi=1.0, col_centre=None):
        """State of a similarity stream whose values satisfy lo <= x - col_centre[k] <= hi (col_centre: [h] fp64 on the
        device, or None for zeros)."""
        need = self.lib.dlc_sdav_stream_state_bytes(int(capacity), int(p), int(h))
        if need == 0:
            raise ValueError("streaming similarity: P <= 32 patches and H <= 32768 (got %d, %d)" % (p, h))
        if col_centre is not None:
            self._check_out("col_centre", col_centre, (int(h),), torch.float64)
        state = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        self._check(self.lib.dlc_sdav_stream_init(self.ctx, _ptr(state), state.numel(), int(capacity), int(p), int(h), float(lo),
                                                   float(hi), _ptr(col_centre), self._stream()))
        return state

    def sdav_stream_append(self, state, desc, n_old, n_total, score):
        cap, p, h = desc.shape
        self._check(self.lib.dlc_sdav_stream_append(self.ctx, _ptr(state), state.numel(), cap, p, h, _ptr(desc), int(n_old),
                                                     int(n_total), _ptr(score), self._stream()))

    def sdav_stream_query(self, state, desc, f, score, a=10.0, b=-10.0, out=None, stats=None):
        cap, p, h = desc.shape
        if out is None:
            out = torch.empty((max(int(f), 0),), dtype=torch.float64, device=self.device)
        if stats is not None:
            self._check_out("stats", stats, (2,), torch.int64)
        self._check(self.lib.dlc_sdav_stream_query(self.ctx, _ptr(state), state.numel(), cap, p, h, _ptr(desc), int(f), _ptr(score),
                                                    float(a), float(b), _ptr(out), _ptr(stats), self._stream()))
        return out

    def sdav_stream_query_batch(self, state, desc, f_first, n_queries, score, a=10.0, b=-10.0, out=None, stats=None):
        """Rows of the resident frames f_first .. f_first + n_queries - 1 against the frames older than each, in one pair of
        launches: out [n_queries, ld >= f_first + n_queries - 1] fp64, row q valid in its first f_first + q entries."""
        cap, p, h = desc.shape
        ld = max(1, int(f_first) + int(n_queries) - 1)
        if out is None:
            out = torch.empty((int(n_queries), ld), dtype=torch.float64, device=self.device)
        else:
            self._check_out("out", out, (int(n_queries), out.shape[1]), torch.float64)
        if stats is not None:
            self._check_out("stats", stats, (2,), torch.int64)
        need = self.lib.dlc_sdav_stream_query_batch_workspace_bytes(cap, p, int(n_queries))
        ws = self.workspace("stream_batch", need)
        self._check(self.lib.dlc_sdav_stream_query_batch(self.ctx, _ptr(state), state.numel(), cap, p, h, _ptr(desc), int(f_first),
                                                          int(n_queries), _ptr(score), float(a), float(b), _ptr(out), out.shape[1],
                                                          _ptr(stats), _ptr(ws), ws.numel(), self._stream()))
        return out

    def sdav_stream_query_batch_staged(self, state, desc, f_first, n_queries, score, stage, out, ws, a=10.0, b=-10.0, stats=None,
                                       stream=None):
        """One half of the strip form of sdav_stream_query_batch (include/dlc.h): stage 1 = the product kernel into the
        caller's workspace ws, stage 2 = resolution + scores into out [n_queries, ld] -- for callers that run a batch's
        stage 2 beside the next batch's stage 1 (loop_closure.SdavLoopClosureDetector.submit).  stream: a torch stream
        (default: the current one)."""
        cap, p, h = desc.shape
        self._check_out("out", out, (int(n_queries), out.shape[1]), torch.float64)
        if stats is not None:
            self._check_out("stats", stats, (2,), torch.int64)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.dlc_sdav_stream_query_batch_staged(self.ctx, _ptr(state), state.numel(), cap, p, h, _ptr(desc),
                                                                 int(f_first), int(n_queries), _ptr(score), float(a), float(b),
                                                                 _ptr(out), out.shape[1], _ptr(stats), _ptr(ws), ws.numel(),
                                                                 int(stage), st))
        return out

    def topk_rows_f64(self, scores, limit0, limit_step, k, poison=None):
        """(scores [rows, k] fp64, idx [rows, k] int64) of the k best of the first limit0 + r * limit_step entries of row r of
        scores [rows, ld] (fp64): score descending, ties -> the lower index, NaN never; (-inf, -1) where fewer.
        poison: a device int64 [1] read by the kernel; non-zero -> every slot (NaN, -1)."""
        self._check_out("scores", scores, tuple(scores.shape), torch.float64)
        if poison is not None:
            self._check_out("poison", poison, (1,), torch.int64)
        rows, ld = scores.shape
        o_s = torch.empty((rows, k), dtype=torch.float64, device=self.device)
        o_i = torch.empty((rows, k), dtype=torch.int64, device=self.device)
        self._check(self.lib.dlc_topk_rows_f64(self.ctx, _ptr(scores), rows, ld, int(limit0), int(limit_step), int(k), _ptr(o_s),
                                                _ptr(o_i), _ptr(poison), self._stream()))
        return o_s, o_i

    def cnnvtl_distance_matrix(self, desc, d=None, out=None):
        """All-vs-all popcount(|a ^ b|) distances of int8 descriptors [N, D] -> int64 [N, N].  d: the descriptor length
        when desc's rows are padded (to a multiple of 4 bytes); out: a caller-kept [N, N] int64 tensor."""
        desc = desc.contiguous()
        n = desc.shape[0]
        if d is None:
            d = desc.shape[1]
            if d % 4:       # rows on 4-byte boundaries let the kernel load words (the bytes past d are masked there)
                padded = torch.zeros((n, (d + 15) // 16 * 16), dtype=torch.int8, device=self.device)
                padded[:, :d] = desc
                desc = padded
        if out is None:
            out = torch.empty((n, n), dtype=torch.int64, device=self.device)
        else:
            self._check_out("out", out, (n, n), torch.int64)
        self._check(self.lib.dlc_cnnvtl_distance_matrix(self.ctx, _ptr(desc), n, int(d), desc.stride(0), _ptr(out),
                                                         self._stream()))
        return out

    # ---- cosine + top-k -----------------------------------------------------------------
    @staticmethod
    def stored_width(d):
        """Width of a stored descriptor row: d zero-padded to a multiple of 64."""
        return (d + K_STEP - 1) // K_STEP * K_STEP

    def normalize(self, x, dtype="bf16", center=False, out=None):
        """Rows of x [n,d] (float32/float64) -> stored descriptors [n, d_pad]
        (bf16/fp16, L2-normalised, zero-padded to a multiple of 64), into `out` if given."""
        dt = torch_dtype(dtype)
        if dt not in (torch.bfloat16, torch.float16):
            raise ValueError("stored descriptor dtype must be bf16 or fp16")
        if x.dtype not in (torch.float32, torch.float64):
            raise ValueError("normalize: float32 or float64 input")
        x = x.contiguous()
        n, d = x.shape
        ldd = self.stored_width(d)
        if out is None:
            out = torch.empty((n, ldd), dtype=dt, device=self.device)
        elif out.shape != (n, ldd) or out.dtype != dt or not out.is_contiguous() or out.device != self.device:
            raise ValueError("normalize: out must be a contiguous [%d, %d] %s tensor on %s" % (n, ldd, dt, self.device))
        if n > 0:
            self._check(self.lib.dlc_l2_normalize_rows(self.ctx, _TORCH_TO_DLC[x.dtype], _ptr(x), n, d, x.stride(0),
                                                        1 if center else 0, _TORCH_TO_DLC[dt], _ptr(out), ldd,
                                                        self._stream()))
        # the mark unit_rows() looks for: THIS tensor object, at this version, holds rows of norm <= 1.005.  A view or a copy
        # does not carry it and an in-place write outdates it -- torch's own or the library's through a raw pointer
        # (_wrote: upload(out=), this call) -- then the match measures the norms instead of trusting
        self._wrote(out)
        out._dlc_unit_rows = out._version
        return out

    @staticmethod
    def unit_rows(t):
        """True when t is a tensor normalize() wrote and nothing has written to it since: its rows are what the cosine
        certificate's static tau is derived for (include/dlc.h, NORMS)."""
        return getattr(t, "_dlc_unit_rows", None) == t._version

    def max_row_norm(self, rows, out=None):
        """Device float [1]: max(out, the largest L2 norm of the stored rows [n, d]) rounded up (dlc_max_row_norm); out
        starts at 0 when not given.  What a database of rows normalize() did NOT write owes the certificate."""
        if rows.dim() != 2 or rows.dtype not in (torch.bfloat16, torch.float16) or rows.stride(1) != 1:
            raise ValueError("max_row_norm: stored rows [n, d] (bf16 / fp16, contiguous rows)")
        if out is None:
            out = torch.zeros((1,), dtype=torch.float32, device=self.device)
        else:
            self._check_out("out", out, (1,), torch.float32)
        if rows.shape[0] > 0:
            self._check(self.lib.dlc_max_row_norm(self.ctx, _TORCH_TO_DLC[rows.dtype], _ptr(rows), rows.shape[0],
                                                   rows.stride(0), rows.shape[1], _ptr(out), self._stream()))
        return out

    def cosine_tau_scale(self, q, db_max_norm=None, out=None, stream=None):
        """[Q] floats: what query i's certificate multiplies tau by, max(1, |q_i| R / 1.01) with R = db_max_norm (a device
        float [1] from max_row_norm; None: the database rows are normalize()'s) -- dlc_cosine_tau_scale."""
        if q.dim() != 2 or q.dtype not in (torch.bfloat16, torch.float16) or q.stride(1) != 1:
            raise ValueError("cosine_tau_scale: stored queries [Q, d] (bf16 / fp16, contiguous rows)")
        if out is None:
            out = torch.empty((q.shape[0],), dtype=torch.float32, device=self.device)
        else:
            self._check_out("out", out, (q.shape[0],), torch.float32)
        if db_max_norm is not None:
            self._check_out("db_max_norm", db_max_norm, (1,), torch.float32)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        if q.shape[0] > 0:
            self._check(self.lib.dlc_cosine_tau_scale(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), q.shape[0], q.stride(0),
                                                       q.shape[1], _ptr(db_max_norm), _ptr(out), st))
        return out

    def _check_stored(self, q, db):
        if q.dim() != 2 or db.dim() != 2 or q.shape[1] != db.shape[1]:
            raise ValueError("stored descriptors must be 2-D with one width")
        if q.dtype != db.dtype or q.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("stored descriptors must both be bf16 or both fp16")
        if q.stride(1) != 1 or db.stride(1) != 1:
            raise ValueError("stored descriptor rows must be contiguous")

    def match_topk(self, q, db, k, row_offset=0, out=None, details=False, older_than=None, tau_scale=None):
        """Top-k cosine match of stored queries q [Q,d] against stored db [N,d].
        Returns (scores [Q,k] float32, idx [Q,k] int64 with row_offset added); with details=True a
        TopK(scores, idx, scores_f64, status): the fp64 scores the order was decided on and, per query,
        0 = certified by the selection, 2 = resolved by the exhaustive pass (include/dlc.h).
        older_than = L: query i only sees db rows below L + i (dlc_cosine_topk_older; (-inf, -1) where it sees fewer than k).
        tau_scale: None states that q and db are normalize()'s rows; otherwise cosine_tau_scale(q, max_row_norm(db))
        (KeyframeDatabase.match_topk keeps track of that by itself)."""
        self._check_stored(q, db)
        nq, d = q.shape
        if tau_scale is not None:
            self._check_out("tau_scale", tau_scale, (nq,), torch.float32)
        n = db.shape[0]
        need = self.lib.dlc_cosine_topk_workspace_bytes(nq, n, d, k)
        if need == 0:
            raise ValueError("match_topk: k=%d outside 1..%d (or empty operand)" % (k, L.DLC_MAX_K))
        ws = self.workspace("topk", need)
        if out is None:
            scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            idx = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            scores = self._check_out("out[0] (scores)", out[0], (nq, k), torch.float32)
            idx = self._check_out("out[1] (idx)", out[1], (nq, k), torch.int64)
        s64 = torch.empty((nq, k), dtype=torch.float64, device=self.device) if details else None
        status = torch.empty((nq,), dtype=torch.int32, device=self.device) if details else None
        if older_than is None:
            self._check(self.lib.dlc_cosine_topk(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), nq, q.stride(0), _ptr(db), n,
                                                  db.stride(0), d, k, row_offset, _ptr(scores), _ptr(s64), _ptr(idx),
                                                  _ptr(status), _ptr(tau_scale), _ptr(ws), ws.numel(), self._stream()))
        else:
            self._check(self.lib.dlc_cosine_topk_older(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), nq, q.stride(0), _ptr(db), n,
                                                        db.stride(0), d, k, row_offset, int(older_than), _ptr(scores),
                                                        _ptr(s64), _ptr(idx), _ptr(status), _ptr(tau_scale), _ptr(ws),
                                                        ws.numel(), self._stream()))
        return TopK(scores, idx, s64, status) if details else (scores, idx)

    def topk_workspace_bytes(self, nq, n, d, k):
        need = self.lib.dlc_cosine_topk_workspace_bytes(nq, n, d, k)
        if need == 0:
            raise ValueError("k=%d outside 1..%d (or empty operand)" % (k, L.DLC_MAX_K))
        return need

    def score_error_bound(self, nq, n, d, k):
        """tau of the plan a [nq, d] x [n, d] top-k match takes: |fp32 score of the score pass - fp64 score| <= tau."""
        return float(self.lib.dlc_cosine_score_error_bound(nq, n, d, k))

    def score_error_bound_any_plan(self, d):
        """The largest tau any plan has for descriptors of (stored) width d: what a sharded merge certifies with -- the same
        number on every rank, whatever plan its shard's size picks."""
        return float(self.lib.dlc_cosine_score_error_bound_any_plan(d))

    def score_groups(self, q, db, k, ws, stream=None):
        """Stage 1 of match_topk (the MFMA score GEMM) into the caller's workspace tensor."""
        self._check_stored(q, db)
        self._check_ws(ws)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.dlc_cosine_score_groups(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), q.shape[0], q.stride(0),
                                                      _ptr(db), db.shape[0], db.stride(0), q.shape[1], k, _ptr(ws),
                                                      ws.numel(), st))

    def select_topk(self, q, db, k, ws, scores, idx, row_offset=0, coop=False, stream=None, scores_f64=None, status=None,
                    tau_scale=None):
        """Stage 2 of match_topk (selection, fp64 re-score, final top-k, certificate, exhaustive pass) from the workspace."""
        self._check_stored(q, db)
        self._check_ws(ws)
        self._check_out("scores", scores, (q.shape[0], k), torch.float32)
        self._check_out("idx", idx, (q.shape[0], k), torch.int64)
        if scores_f64 is not None:
            self._check_out("scores_f64", scores_f64, (q.shape[0], k), torch.float64)
        if status is not None:
            self._check_out("status", status, (q.shape[0],), torch.int32)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.dlc_cosine_select_topk(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), q.shape[0], q.stride(0),
                                                     _ptr(db), db.shape[0], db.stride(0), q.shape[1], k, row_offset,
                                                     _ptr(scores), _ptr(scores_f64), _ptr(idx), _ptr(status),
                                                     _ptr(tau_scale), _ptr(ws), ws.numel(),
                                                     L.DLC_SELECT_COOP if coop else 0, st))

    def _check_ws(self, ws):
        if not isinstance(ws, torch.Tensor) or ws.dtype != torch.uint8 or not ws.is_contiguous() or ws.device != self.device:
            raise ValueError("workspace must be a contiguous uint8 tensor on %s" % (self.device,))

    def groups_per_query(self, k):
        kg = self.lib.dlc_cosine_groups_per_query(k)
        if kg == 0:
            raise ValueError("k=%d outside 1..%d" % (k, L.DLC_MAX_K))
        return kg

    def select_groups(self, q, db, k, ws, grp_ids, grp_max, coop=False, stream=None):
        """Stage 2a: the kg best groups of every query: shard-local ids [Q,kg] and grp_max [Q,kg+1] = their maxima +
        (last column) the best maximum among the groups that are not listed."""
        self._check_stored(q, db)
        self._check_ws(ws)
        kg = self.groups_per_query(k)
        self._check_out("grp_ids", grp_ids, (q.shape[0], kg), torch.int32)
        self._check_out("grp_max", grp_max, (q.shape[0], kg + 1), torch.float32)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.dlc_cosine_select_groups(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), q.shape[0], q.stride(0),
                                                       _ptr(db), db.shape[0], db.stride(0), q.shape[1], k, _ptr(ws),
                                                       ws.numel(), _ptr(grp_ids), _ptr(grp_max),
                                                       L.DLC_SELECT_COOP if coop else 0, st))

    def rescore_topk(self, q, db, k, grp_ids, grp_max, scores_f64, idx, bound=None, all_max=None, row_offset=0, coop=False,
                     stream=None, tau_scale=None):
        """Stage 2b: fp64 re-score of the listed groups (filtered against all shards' maxima when all_max
        [parts, Q, kg+1] is given): the shard's part of the top-k (fp64 scores, rows) and bound [Q] = the best fp32
        score any row outside the surviving groups of all shards can have."""
        self._check_stored(q, db)
        kg = self.groups_per_query(k)
        self._check_out("grp_ids", grp_ids, (q.shape[0], kg), torch.int32)
        self._check_out("grp_max", grp_max, (q.shape[0], kg + 1), torch.float32)
        self._check_out("scores_f64", scores_f64, (q.shape[0], k), torch.float64)
        self._check_out("idx", idx, (q.shape[0], k), torch.int64)
        if bound is not None:
            self._check_out("bound", bound, (q.shape[0],), torch.float32)
        if all_max is not None:
            self._check_out("all_max", all_max, (all_max.shape[0], q.shape[0], kg + 1), torch.float32)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        parts = 0 if all_max is None else all_max.shape[0]
        self._check(self.lib.dlc_cosine_rescore_topk(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), q.shape[0], q.stride(0),
                                                      _ptr(db), db.shape[0], db.stride(0), q.shape[1], k, row_offset,
                                                      _ptr(grp_ids), _ptr(grp_max), _ptr(all_max), parts, _ptr(scores_f64),
                                                      _ptr(idx), _ptr(bound), _ptr(tau_scale),
                                                      L.DLC_SELECT_COOP if coop else 0, st))

    def exhaustive_topk(self, q, db, k, ws, lower, tau, status, scores_f64, idx, scores=None, row_offset=0, stream=None,
                        tau_scale=None):
        """The exhaustive pass of the sharded protocol for the queries with status == 1: lower [Q] fp64 = the k-th score
        found so far; this shard's exact top-k over every group whose maximum (in `ws`, the workspace of the score
        pass) is >= lower - tau replaces scores_f64 / idx (and scores) of those queries; their status becomes 2."""
        self._check_stored(q, db)
        self._check_ws(ws)
        nq = q.shape[0]
        self._check_out("lower", lower, (nq,), torch.float64)
        self._check_out("status", status, (nq,), torch.int32)
        self._check_out("scores_f64", scores_f64, (nq, k), torch.float64)
        self._check_out("idx", idx, (nq, k), torch.int64)
        if scores is not None:
            self._check_out("scores", scores, (nq, k), torch.float32)
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self._stream()
        self._check(self.lib.dlc_cosine_exhaustive_topk(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), nq, q.stride(0), _ptr(db),
                                                         db.shape[0], db.stride(0), q.shape[1], k, row_offset, _ptr(lower), 1,
                                                         float(tau), _ptr(tau_scale), _ptr(status), _ptr(scores),
                                                         _ptr(scores_f64), _ptr(idx), _ptr(ws), ws.numel(), st))

    def topk_merge_packed(self, gathered, nq, k, out, bound=None, tau=0.0, scores_f64=None, status=None, tau_scale=None):
        """Merge an all-gather of packed per-shard results: gathered is uint8 [parts, nq*k*16], each part = int64 idx
        [nq,k] followed by float64 scores [nq,k].  With bound [nq] / tau the merge certifies into status [nq]."""
        parts = gathered.shape[0]
        self._check_out("gathered", gathered, (parts, nq * k * 16), torch.uint8)
        base = gathered.data_ptr()
        o_s = self._check_out("out[0] (scores)", out[0], (nq, k), torch.float32)
        o_i = self._check_out("out[1] (idx)", out[1], (nq, k), torch.int64)
        if bound is not None:
            self._check_out("bound", bound, (nq,), torch.float32)
        if scores_f64 is not None:
            self._check_out("scores_f64", scores_f64, (nq, k), torch.float64)
        if status is not None:
            self._check_out("status", status, (nq,), torch.int32)
        self._check(self.lib.dlc_topk_merge_strided(self.ctx, C.c_void_p(base + nq * k * 8), nq * k * 2,
                                                     C.c_void_p(base), nq * k * 2, parts, nq, k, _ptr(bound), float(tau),
                                                     _ptr(tau_scale), _ptr(o_s), _ptr(scores_f64), _ptr(o_i), _ptr(status),
                                                     self._stream()))
        return o_s, o_i

    def topk_keep_older(self, scores, idx, limit0, k):
        """Row b of best-first (scores, idx) [B, kk] -> its first k entries with 0 <= id < limit0 + b, (-inf, -1) after."""
        scores, idx = scores.contiguous(), idx.contiguous()
        if scores.dim() != 2 or idx.shape != scores.shape or scores.dtype != torch.float32 or idx.dtype != torch.int64:
            raise ValueError("topk_keep_older: scores float32 / idx int64 of one shape [B, kk]")
        b, kk = scores.shape
        o_s = torch.empty((b, k), dtype=torch.float32, device=self.device)
        o_i = torch.empty((b, k), dtype=torch.int64, device=self.device)
        self._check(self.lib.dlc_topk_keep_older(self.ctx, _ptr(scores), _ptr(idx), b, kk, int(limit0), int(k), _ptr(o_s),
                                                  _ptr(o_i), self._stream()))
        return o_s, o_i

    def topk_merge(self, scores_f64, idx, out=None, details=False):
        """Merge [parts, Q, k] per-shard results (fp64 scores: the order is decided on them) into the global [Q, k]:
        (scores float32, idx); with details=True a TopK with the merged fp64 scores too."""
        scores_f64, idx = scores_f64.contiguous(), idx.contiguous()
        if scores_f64.dim() != 3 or idx.shape != scores_f64.shape or scores_f64.dtype != torch.float64 or \
                idx.dtype != torch.int64:
            raise ValueError("topk_merge: scores float64 / idx int64 of one shape [parts, Q, k]")
        parts, nq, k = scores_f64.shape
        if out is None:
            o_s = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            o_i = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            o_s = self._check_out("out[0] (scores)", out[0], (nq, k), torch.float32)
            o_i = self._check_out("out[1] (idx)", out[1], (nq, k), torch.int64)
        o_64 = torch.empty((nq, k), dtype=torch.float64, device=self.device) if details else None
        self._check(self.lib.dlc_topk_merge(self.ctx, _ptr(scores_f64), _ptr(idx), parts, nq, k, _ptr(o_s), _ptr(o_64),
                                             _ptr(o_i), self._stream()))
        return TopK(o_s, o_i, o_64, None) if details else (o_s, o_i)

    def cosine_scores(self, q, db):
        self._check_stored(q, db)
        nq, d = q.shape
        n = db.shape[0]
        s = torch.empty((nq, n), dtype=torch.float32, device=self.device)
        need = self.lib.dlc_cosine_scores_workspace_bytes(nq, n, d)          # split-K partials; 0 for most shapes
        ws = self.workspace("scores", need) if need else None
        self._check(self.lib.dlc_cosine_scores(self.ctx, _TORCH_TO_DLC[q.dtype], _ptr(q), nq, q.stride(0), _ptr(db), n,
                                                db.stride(0), d, _ptr(s), s.stride(0), _ptr(ws) if need else None,
                                                need, self._stream()))
        return s

    # ---- profiling hooks for bench.py ---------------------------------------------------
    def set_profiling(self, enabled):
        self._check(self.lib.dlc_set_profiling(self.ctx, 1 if enabled else 0))

    def profile_gemm_ms(self, capacity=256):
        """Durations (ms) of the score-GEMM launches of the last calls, from HIP events
        recorded on the launch stream (blocks until they complete)."""
        buf = (C.c_float * capacity)()
        n = self.lib.dlc_profile_gemm_ms(self.ctx, buf, capacity)
        if n < 0:
            self._check(n)
        return [float(buf[i]) for i in range(n)]


_default = {}


def default_engine(device=None):
    """Process-wide engine for the current (or given) device."""
    if device is None:
        if not torch.cuda.is_available():
            L.load()   # raises ImportError first if the library itself is missing
            raise RuntimeError("deeploopcloser_amd needs a visible MI355X; there is no CPU fallback")
        device = torch.cuda.current_device()
    elif not isinstance(device, int):
        device = torch.device(device).index or 0      # one engine per GPU, whatever names it
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]
