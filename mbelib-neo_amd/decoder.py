"""BatchDecoder -- host mirror of the reference's frame-level entry points for a batch of
streams.  One object owns the per-stream state ({cur, prev, prev_enhanced} + the per-stream
RNG state that the reference keeps thread-local) in HBM and feeds frames through the C-ABI
launcher.  torch is used for device memory and streams only.

Reference interface mirrored (include/mbelib-neo/mbelib.h):
  mbe_initMbeParms                      -> BatchDecoder(...)            (state defaults)
  mbe_setThreadRngSeed                  -> seeds=...                    (per stream)
  mbe_processImbe7200x4400Frame[f]      -> decode(frames, T)            codec=IMBE
  mbe_processAmbe3600x2450Frame[f]      -> decode(frames, T)            codec=AMBE
  mbe_decode*Frame                      -> fec(frames)
"""
import numpy as np

from . import _native
from .layout import (
    CODEC_IMBE7200X4400,
    FRAME_BYTES,
    PARMS_DTYPE,
    RECORD_DTYPE,
    RESULT_DTYPE,
    RNG_DTYPE,
    init_state,
    load_tables_blob,
    rng_default,
    rng_seeded,
)

_initialised = {}


def ensure_init(device_index, blob=None):
    """mbx_init once per (process, device) -- the native library keeps one context per device and every
    launcher uses the context of the calling thread's current device.  Returns the blob checksum."""
    device_index = int(device_index)
    if device_index not in _initialised:
        blob = blob if blob is not None else load_tables_blob()
        L = _native.lib()
        _native.check(L.mbx_init(device_index, blob, len(blob)), "mbx_init")   # also makes it the current device
        _initialised[device_index] = L.mbx_table_checksum()
    return _initialised[device_index]


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise _native.NativeLibraryError("no HIP device visible to torch; mbelib-neo_amd has no CPU fallback")
    return torch


class BatchDecoder:
    """resident=True: the decoder owns the state between launches and uses the resident form (mbx_process_batch_resident,
    include/mbx.h): prev_mp_enhanced elided while it equals cur_mp, prev_mp fetched lazily; state_numpy() materialises the
    ABI triplets first.  PCM, results and state are bit-identical to the default."""

    def __init__(self, codec, streams, device=0, seeds=None, tables_blob=None, resident=False):
        torch = _torch()
        self.codec = int(codec)
        self.streams = int(streams)
        self.device = torch.device("cuda", int(device))
        torch.cuda.set_device(self.device)
        ensure_init(self.device.index, tables_blob)
        st = init_state(self.streams)
        rg = rng_default(self.streams) if seeds is None else rng_seeded(seeds)
        self.state = torch.from_numpy(st.view(np.uint8).reshape(-1)).to(self.device)
        self.rng = torch.from_numpy(rg.view(np.uint8).reshape(-1)).to(self.device)
        self.resident = torch.zeros(self.streams, dtype=torch.int32, device=self.device) if resident else None

    # -- state access (host copies) --------------------------------------------------------
    def materialize(self):
        """resident decoders: write the elided prev_mp_enhanced structs out (no-op otherwise)"""
        if self.resident is not None:
            torch = _torch()
            with torch.cuda.device(self.device):
                _native.check(_native.lib().mbx_resident_materialize(self.streams, None, self.state.data_ptr(), self.resident.data_ptr(),
                                                                     torch.cuda.current_stream().cuda_stream), "mbx_resident_materialize")

    def state_numpy(self):
        self.materialize()
        return self.state.cpu().numpy().view(PARMS_DTYPE).reshape(self.streams, 3)

    def rng_numpy(self):
        return self.rng.cpu().numpy().view(RNG_DTYPE)

    def set_state(self, state, rng=None):
        torch = _torch()
        self.state.copy_(torch.from_numpy(np.ascontiguousarray(state).view(np.uint8).reshape(-1)))
        if self.resident is not None:
            self.resident.zero_()
        if rng is not None:
            self.rng.copy_(torch.from_numpy(np.ascontiguousarray(rng).view(np.uint8).reshape(-1)))

    def to_device(self, frames):
        torch = _torch()
        if isinstance(frames, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(frames).reshape(-1)).to(self.device)
        return frames

    # -- launches ----------------------------------------------------------------------------
    def fec(self, frames):
        """FEC stage only: wire frames -> parameter records (device tensor of uint32 [n, 4])."""
        torch = _torch()
        d_frames = self.to_device(frames)
        n = d_frames.numel() // FRAME_BYTES[self.codec]
        rec = torch.empty((n, 4), dtype=torch.int32, device=self.device)
        L = _native.lib()
        fn = {0: L.mbx_fec_imbe7200x4400, 1: L.mbx_fec_ambe3600x2450, 2: L.mbx_fec_imbe7100x4400,
              3: L.mbx_fec_ambe3600x2450}[self.codec]   # both AMBE codecs share the FEC front end
        with torch.cuda.device(self.device):
            _native.check(fn(d_frames.data_ptr(), n, rec.data_ptr(), torch.cuda.current_stream().cuda_stream), "mbx_fec")
        return rec

    def make_outputs(self, T, want_pcm16=True, want_float=False, want_results=True):
        torch = _torch()
        n = self.streams * T
        out = {"records": torch.empty((n, 4), dtype=torch.int32, device=self.device)}
        out["pcm16"] = torch.empty((n, 160), dtype=torch.int16, device=self.device) if want_pcm16 else None
        out["pcmf"] = torch.empty((n, 160), dtype=torch.float32, device=self.device) if want_float else None
        out["results"] = torch.empty((n, 5), dtype=torch.int32, device=self.device) if want_results else None
        return out

    def decode(self, frames, T, want_pcm16=True, want_float=False, want_results=True, out=None, staged=False):
        """S x T frames, stream-major (stream s, frame t at index s*T + t).  Asynchronous on the
        current torch stream; returns device tensors.
        staged=True: the stages as separate calls (mbx_fec_* then mbx_process_records) instead of mbx_process_batch -- the
        same results by contract; the IMBE codecs at T = 1 take one fused launch in mbx_process_batch and the FEC /
        expansion / stream launches here (tests compare the two)."""
        torch = _torch()
        d_frames = self.to_device(frames)
        n = self.streams * int(T)
        if d_frames.numel() != n * FRAME_BYTES[self.codec]:
            raise ValueError("frames must hold streams*T wire frames")
        if out is None:
            out = self.make_outputs(T, want_pcm16, want_float, want_results)

        def ptr(t):
            return t.data_ptr() if t is not None else None

        with torch.cuda.device(self.device):   # the launcher works on the current device's context
            if staged:
                if self.resident is not None:
                    raise ValueError("staged decoding is for the ABI-triplet form")
                L = _native.lib()
                strm = torch.cuda.current_stream().cuda_stream
                fn = {0: L.mbx_fec_imbe7200x4400, 1: L.mbx_fec_ambe3600x2450, 2: L.mbx_fec_imbe7100x4400, 3: L.mbx_fec_ambe3600x2450}[self.codec]
                _native.check(fn(d_frames.data_ptr(), n, out["records"].data_ptr(), strm), "mbx_fec")
                rc = L.mbx_process_records(0 if self.codec == 2 else self.codec, self.streams, int(T), out["records"].data_ptr(),
                                           self.state.data_ptr(), self.rng.data_ptr(), ptr(out["pcm16"]), ptr(out["pcmf"]),
                                           ptr(out["results"]), strm)
            elif self.resident is not None:
                rc = _native.lib().mbx_process_batch_resident(
                    self.codec, self.streams, int(T), None, d_frames.data_ptr(), self.state.data_ptr(), self.resident.data_ptr(),
                    self.rng.data_ptr(), ptr(out["pcm16"]), ptr(out["pcmf"]), ptr(out["results"]), out["records"].data_ptr(),
                    torch.cuda.current_stream().cuda_stream,
                )
            else:
                rc = _native.lib().mbx_process_batch(
                    self.codec, self.streams, int(T), d_frames.data_ptr(), self.state.data_ptr(), self.rng.data_ptr(),
                    ptr(out["pcm16"]), ptr(out["pcmf"]), ptr(out["results"]), out["records"].data_ptr(),
                    torch.cuda.current_stream().cuda_stream,
                )
        _native.check(rc, "mbx_process_batch")
        return out


def results_numpy(t):
    return t.cpu().numpy().view(RESULT_DTYPE).reshape(-1)


RESULT_HIST_FIELDS = ("frames", "soft_input", "c0_valid", "c4_valid", "flag3", "tone", "erasure", "repeat", "mute", "c0_errors",
                      "protected_errors", "c4_errors", "total_errors", "frames_with_errors")


def result_histogram(results, stream=None):
    """mbx_result_histogram over a device tensor of mbe_process_result (as the decoders return them): the per-batch tally of flags
    and error counts, formed on the device (include/mbx.h: mbx_result_hist).  Returns a dict of Python ints."""
    torch = _torch()
    L = _native.lib()
    n = results.numel() * results.element_size() // RESULT_DTYPE.itemsize
    hist = torch.zeros(len(RESULT_HIST_FIELDS), dtype=torch.int64, device=results.device)
    _native.check(L.mbx_result_histogram(results.data_ptr(), n, hist.data_ptr(), stream if stream is not None else torch.cuda.current_stream().cuda_stream),
                  "mbx_result_histogram")
    return dict(zip(RESULT_HIST_FIELDS, (int(v) for v in hist.cpu().tolist())))


def records_numpy(t):
    return t.cpu().numpy().view(RECORD_DTYPE).reshape(-1)


def synthesize_speech(cur, prev, rng, device=0, want_pcm16=False):
    """mbe_synthesizeSpeechf for S (cur, prev) pairs given as numpy PARMS arrays; returns
    (pcmf [S,160], cur', prev', rng', pcm16 or None).  Host buffers in, host buffers out."""
    ensure_init(device)
    cur = np.ascontiguousarray(cur).copy()
    prev = np.ascontiguousarray(prev).copy()
    rng = np.ascontiguousarray(rng).copy()
    S = cur.shape[0]
    pcmf = np.empty((S, 160), dtype=np.float32)
    pcm16 = np.empty((S, 160), dtype=np.int16) if want_pcm16 else None
    rc = _native.lib().mbx_synthesize_speech_host(
        S, cur.ctypes.data, prev.ctypes.data, rng.ctypes.data, pcmf.ctypes.data,
        pcm16.ctypes.data if pcm16 is not None else None,
    )
    _native.check(rc, "mbx_synthesize_speech_host")
    return pcmf, cur, prev, rng, pcm16


def floattoshort(pcmf, device=0):
    """mbe_floattoshort over [n,160] float frames (host in/out)."""
    ensure_init(device)
    pcmf = np.ascontiguousarray(pcmf, dtype=np.float32)
    out = np.empty(pcmf.shape, dtype=np.int16)
    _native.check(_native.lib().mbx_floattoshort_host(pcmf.ctypes.data, out.ctypes.data, pcmf.shape[0]), "mbx_floattoshort_host")
    return out


def process_batch_host(codec, S, T, frames, state, rng, device=0):
    """Whole pipeline on host buffers (numpy in, numpy out): returns dict."""
    ensure_init(device)
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    state = np.ascontiguousarray(state).copy()
    rng = np.ascontiguousarray(rng).copy()
    n = S * T
    pcm16 = np.empty((n, 160), dtype=np.int16)
    pcmf = np.empty((n, 160), dtype=np.float32)
    results = np.empty(n, dtype=RESULT_DTYPE)
    records = np.empty(n, dtype=RECORD_DTYPE)
    rc = _native.lib().mbx_process_batch_host(
        codec, S, T, frames.ctypes.data, state.ctypes.data, rng.ctypes.data, pcm16.ctypes.data, pcmf.ctypes.data,
        results.ctypes.data, records.ctypes.data,
    )
    _native.check(rc, "mbx_process_batch_host")
    return {"pcm16": pcm16, "pcmf": pcmf, "results": results, "records": records, "state": state, "rng": rng}


# ---- soft-decision front end (mbe_soft_bit arrays: uint8 [..., 2] = (bit, reliability)) ------------
SOFT_CELLS = {0: 184, 1: 96, 2: 168, 3: 96}


def _soft_array(codec, soft, n):
    soft = np.ascontiguousarray(soft, dtype=np.uint8)
    if soft.size != n * SOFT_CELLS[codec] * 2:
        raise ValueError(f"expected {n} soft frames of {SOFT_CELLS[codec]} (bit, reliability) pairs")
    return soft


def fec_soft_host(codec, soft, device=0):
    """mbe_decode*SoftFrame for a batch: soft [n, 184|96, 2] uint8 -> parameter records."""
    ensure_init(device)
    soft = np.ascontiguousarray(soft, dtype=np.uint8)
    n = soft.size // (SOFT_CELLS[codec] * 2)
    soft = _soft_array(codec, soft, n)
    records = np.empty(n, dtype=RECORD_DTYPE)
    _native.check(_native.lib().mbx_fec_soft_host(codec, soft.ctypes.data, n, records.ctypes.data), "mbx_fec_soft_host")
    return records


def process_batch_soft_host(codec, S, T, soft, state, rng, device=0):
    """mbe_process*SoftFramef for S streams x T frames on host buffers."""
    ensure_init(device)
    n = S * T
    soft = _soft_array(codec, soft, n)
    state = np.ascontiguousarray(state).copy()
    rng = np.ascontiguousarray(rng).copy()
    pcm16 = np.empty((n, 160), dtype=np.int16)
    pcmf = np.empty((n, 160), dtype=np.float32)
    results = np.empty(n, dtype=RESULT_DTYPE)
    records = np.empty(n, dtype=RECORD_DTYPE)
    rc = _native.lib().mbx_process_batch_soft_host(
        codec, S, T, soft.ctypes.data, state.ctypes.data, rng.ctypes.data, pcm16.ctypes.data, pcmf.ctypes.data,
        results.ctypes.data, records.ctypes.data,
    )
    _native.check(rc, "mbx_process_batch_soft_host")
    return {"pcm16": pcm16, "pcmf": pcmf, "results": results, "records": records, "state": state, "rng": rng}


def ecc_soft_words_host(kind, soft, device=0):
    """mbe_golay2312Soft (kind 0, soft [n, 23, 2]) / mbe_hamming1511Soft (kind 1, soft [n, 15, 2]) /
    mbe_7100x4400hamming1511Soft (kind 2):
    returns (corrected words, return values)."""
    ensure_init(device)
    width = 23 if kind == 0 else 15
    soft = np.ascontiguousarray(soft, dtype=np.uint8)
    n = soft.size // (width * 2)
    out = np.empty(n, dtype=np.uint32)
    errs = np.empty(n, dtype=np.int32)
    rc = _native.lib().mbx_ecc_soft_words_host(kind, soft.ctypes.data, n, out.ctypes.data, errs.ctypes.data)
    _native.check(rc, "mbx_ecc_soft_words_host")
    return out, errs


def soft_bits_from_llr(llr):
    llr = np.ascontiguousarray(llr, dtype=np.int16)
    soft = np.empty(llr.shape + (2,), dtype=np.uint8)
    _native.check(_native.lib().mbx_soft_bits_from_llr(llr.ctypes.data, soft.ctypes.data, llr.size), "mbx_soft_bits_from_llr")
    return soft


def records_from_bits(bits, total_errors=None, c0_errors=None, flags=0):
    """Parameter records from bit arrays [n, 88|49] (0/1) -- the mbe_process*Data entry of the batch API.
    total_errors goes into the protected-error field unless c0_errors is given (then flags should carry
    MBE_PROCESS_FLAG_C0_VALID = 2)."""
    bits = np.asarray(bits, dtype=np.uint8)
    n, nb = bits.shape
    padded = np.zeros((n, 96), dtype=np.uint8)
    padded[:, :nb] = bits
    words = np.packbits(padded, axis=1).reshape(n, 3, 4)
    rec = np.zeros(n, dtype=RECORD_DTYPE)
    rec["w"][:, :3] = (words[:, :, 0].astype(np.uint32) << 24) | (words[:, :, 1].astype(np.uint32) << 16) | (
        words[:, :, 2].astype(np.uint32) << 8) | words[:, :, 3].astype(np.uint32)
    tot = np.zeros(n, dtype=np.uint32) if total_errors is None else np.asarray(total_errors, dtype=np.uint32)
    c0 = np.zeros(n, dtype=np.uint32) if c0_errors is None else np.asarray(c0_errors, dtype=np.uint32)
    rec["w"][:, 3] = c0 | ((tot - c0) << 8) | (np.uint32(flags) << 24)
    return rec


def process_records_host(codec, S, T, records, state, rng, device=0):
    """mbx_process_records on host buffers: records [S*T] stream-major; returns dict like process_batch_host."""
    torch = _torch()
    ensure_init(device)
    dev = torch.device("cuda", int(device))
    n = S * T
    d_rec = torch.from_numpy(np.ascontiguousarray(records).view(np.uint8).reshape(-1)).to(dev)
    d_state = torch.from_numpy(np.ascontiguousarray(state).view(np.uint8).reshape(-1).copy()).to(dev)
    d_rng = torch.from_numpy(np.ascontiguousarray(rng).view(np.uint8).reshape(-1).copy()).to(dev)
    pcm16 = torch.empty((n, 160), dtype=torch.int16, device=dev)
    pcmf = torch.empty((n, 160), dtype=torch.float32, device=dev)
    results = torch.empty((n, 5), dtype=torch.int32, device=dev)
    rc = _native.lib().mbx_process_records(codec, S, T, d_rec.data_ptr(), d_state.data_ptr(), d_rng.data_ptr(), pcm16.data_ptr(),
                                           pcmf.data_ptr(), results.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "mbx_process_records")
    torch.cuda.synchronize()
    return {
        "pcm16": pcm16.cpu().numpy(), "pcmf": pcmf.cpu().numpy(), "results": results_numpy(results),
        "state": d_state.cpu().numpy().view(PARMS_DTYPE).reshape(S, 3), "rng": d_rng.cpu().numpy().view(RNG_DTYPE),
    }
