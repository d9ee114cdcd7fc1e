// mbx_session.hip -- the fan-in path of libmbx_hip.so: a session keeps the model state of S streams RESIDENT on the
// device and takes batches of wire frames from HOST memory, returning PCM to HOST memory (include/mbx.h, "sessions").
//
// What crosses PCIe per 20 ms frame is the wire frame in (18 B / 9 B) and the PCM out (320 B int16, + 20 B when the
// mbe_process_result is asked for) -- not the 15.6 KB of state the per-call conveniences (mbx_process_batch_host) move.
//
// Pipeline: kDepth slots of device buffers, two HIP streams.  Batch k uses slot k % kDepth:
//     compute stream   stage-in kernel (frames(k) read from pinned host memory), FEC + stream kernels -> event comp(k)
//     copy-out stream  wait comp(k); D2H PCM / results (DMA)                                          -> event done(k)
// Batches run in submission order (the state of batch k+1 depends on batch k); kernels(k+1) overlap D2H(k).  The frames
// are NOT brought in by a DMA copy: on this stack a host-to-device copy of batch k+1 queues behind the device-to-host copy
// of batch k on the copy engine, which put the next batch's kernels behind that transfer (65,536 frames: 0.257 ms of
// kernels + 0.379 ms of PCM back to back = 103 M frames/s; overlapped the PCM transfer alone bounds the rate).
// The host blocks only when it is kDepth batches ahead of the device.
// Host buffers that are pinned (mbx_host_alloc / hipHostMalloc / hipHostRegister) are DMA targets as they are; pageable
// buffers are staged through pinned memory owned by the slot (one extra host memcpy each way).
//
// Host code only; every number is computed by the kernels of mbx_fec.hip / mbx_stream.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "mbx.h"

void mbx_set_error_text(const char* text);   // mbx_api.hip: the per-thread text behind mbx_last_error()

namespace {

constexpr int kDepth = 3;

int sfail(const char* what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof(buf), "session: %s: %s", what, hipGetErrorString(e));
    mbx_set_error_text(buf);
    return MBX_ENODEVICE;
}

#define S_TRY(expr)                        \
    do {                                   \
        hipError_t e_ = (expr);            \
        if (e_ != hipSuccess) {            \
            return sfail(#expr, e_);       \
        }                                  \
    } while (0)

bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // an ordinary (pageable) pointer: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

struct CopyOut {   // a staged output still to be handed to the caller's pageable buffer
    void*       dst = nullptr;
    const void* src = nullptr;
    size_t      bytes = 0;
};

struct Slot {
    uint8_t*            d_frames = nullptr;
    mbx_param_record*   d_records = nullptr;
    int16_t*            d_pcm16 = nullptr;
    float*              d_pcmf = nullptr;
    mbe_process_result* d_results = nullptr;
    int32_t*            d_index = nullptr;
    void*               d_workspace = nullptr;
    // pinned staging, allocated on first need
    uint8_t*            h_in = nullptr;
    int32_t*            h_index = nullptr;
    int16_t*            h_pcm16 = nullptr;
    float*              h_pcmf = nullptr;
    mbe_process_result* h_results = nullptr;
    mbx_param_record*   h_records = nullptr;
    hipEvent_t          comp = nullptr, done = nullptr;
    bool                busy = false;
    CopyOut             out[4];
    int                 nout = 0;
};

}  // namespace

struct mbx_session {
    int      device = -1;
    int      codec = 0;
    int      streams = 0;
    size_t   max_frames = 0;
    unsigned outputs = 0;
    size_t   frame_bytes = 0;
    mbe_parms*      d_state = nullptr;   // [streams][3]
    mbx_stream_rng* d_rng = nullptr;     // [streams]
    uint32_t*       d_resident = nullptr;   // [streams]: prev_mp_enhanced of the stream is elided (mbx_process_batch_resident, include/mbx.h)
    hipStream_t s_comp = nullptr, s_out = nullptr;
    Slot     slot[kDepth];
    unsigned long long submitted = 0;
    std::vector<uint8_t> seen;   // submit_indexed: one mark per stream, to refuse an index that names a stream twice
};

namespace {

int finish_slot(Slot& sl) {   // wait for the slot's batch and hand staged outputs over
    if (!sl.busy) {
        return 0;
    }
    S_TRY(hipEventSynchronize(sl.done));
    for (int i = 0; i < sl.nout; ++i) {
        memcpy(sl.out[i].dst, sl.out[i].src, sl.out[i].bytes);
    }
    sl.nout = 0;
    sl.busy = false;
    return 0;
}

template <class U>
int pinned(U*& p, size_t count) {
    if (!p) {
        S_TRY(hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(U), hipHostMallocDefault));
    }
    return 0;
}

struct DeviceGuard {   // a session is bound to the device it was created on, whatever the caller's current device is
    int before = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&before);
        if (before != dev) {
            (void)hipSetDevice(dev);
        } else {
            before = -1;
        }
    }
    ~DeviceGuard() {
        if (before >= 0) {
            (void)hipSetDevice(before);
        }
    }
};

void default_parms(mbe_parms& p) {   // ref mbe_initMbeParms, src/core/mbelib.c:367-410 (defaults only)
    memset(&p, 0, sizeof(p));
    p.w0 = (float)((4.0 * 3.14159265358979323846) / (134.0 + 39.5));
    p.L = (int)(0.9254 * (int)((3.14159265358979323846 / p.w0) + 0.25));
    p.K = 12;
    for (int l = 0; l <= 56; ++l) {
        p.Ml[l] = 1.0f;
    }
    p.localEnergy = 75000.0f;
    p.amplitudeThreshold = 20480;
    p.mutingThreshold = MBE_MUTING_THRESHOLD_IMBE;
    p.noiseSeed = -1.0f;
}

bool range_ok(const mbx_session* s, int first, int count) {
    return s && first >= 0 && count >= 0 && (long long)first + count <= s->streams;
}

int enqueue(mbx_session* s, Slot& sl, int n, int T, const int32_t* index, const uint8_t* frames, int16_t* pcm16, float* pcmf,
            mbe_process_result* results, mbx_param_record* records);

int submit(mbx_session* s, int n, int T, const int32_t* index, const uint8_t* frames, int16_t* pcm16, float* pcmf,
           mbe_process_result* results, mbx_param_record* records) {
    if (!s || !frames || n < 0 || T < 0 || n > s->streams || (size_t)n * (size_t)T > s->max_frames) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if ((pcm16 && !(s->outputs & MBX_SESSION_PCM16)) || (pcmf && !(s->outputs & MBX_SESSION_PCMF))
        || (results && !(s->outputs & MBX_SESSION_RESULTS))) {
        return MBE_STATUS_INVALID_ARGUMENT;   // an output the session was not created for
    }
    if (n == 0 || T == 0) {
        return 0;
    }
    if (index) {   // every row must name its own stream: two rows on one stream would race on its state
        s->seen.assign((size_t)s->streams, 0);
        for (int i = 0; i < n; ++i) {
            if (index[i] < 0 || index[i] >= s->streams || s->seen[(size_t)index[i]]) {
                mbx_set_error_text("session: stream index out of range or listed twice");
                return MBE_STATUS_INVALID_ARGUMENT;
            }
            s->seen[(size_t)index[i]] = 1;
        }
    }
    DeviceGuard guard(s->device);
    Slot& sl = s->slot[s->submitted % kDepth];
    int rc = finish_slot(sl);   // the batch that used this slot kDepth submissions ago
    if (rc < 0) {
        return rc;
    }
    // From here on work is queued on the three streams against this slot's buffers.  If anything fails halfway the slot is
    // NOT marked busy, so before the error goes back the streams are drained: the next submit may reuse the slot at once.
    rc = enqueue(s, sl, n, T, index, frames, pcm16, pcmf, results, records);
    if (rc < 0) {
        (void)hipStreamSynchronize(s->s_comp);
        (void)hipStreamSynchronize(s->s_out);
        (void)hipGetLastError();
        sl.nout = 0;
        return rc;
    }
    sl.busy = true;
    ++s->submitted;
    return 0;
}

int enqueue(mbx_session* s, Slot& sl, int n, int T, const int32_t* index, const uint8_t* frames, int16_t* pcm16, float* pcmf,
            mbe_process_result* results, mbx_param_record* records) {
    int rc = 0;
    const size_t nf = (size_t)n * (size_t)T;
    // ---- in ----
    const uint8_t* src = frames;
    if (!is_pinned(frames)) {
        rc = pinned(sl.h_in, s->max_frames * s->frame_bytes);
        if (rc < 0) {
            return rc;
        }
        memcpy(sl.h_in, frames, nf * s->frame_bytes);
        src = sl.h_in;
    }
    // The frames (and the stream index) are fetched by a kernel on the COMPUTE stream, not by a DMA copy: see stage_in_kernel
    // (mbx_fec.hip).  A pinned buffer that is not 16-byte aligned goes through the slot's own pinned staging like a pageable one.
    if (reinterpret_cast<uintptr_t>(src) & 15u) {
        rc = pinned(sl.h_in, s->max_frames * s->frame_bytes);
        if (rc < 0) {
            return rc;
        }
        memcpy(sl.h_in, frames, nf * s->frame_bytes);
        src = sl.h_in;
    }
    const void* dev_view = src;   // what the GPU dereferences: the same address for hipHostMalloc memory, possibly another one for
    void* mapped = nullptr;       // memory the host registered itself (hipHostRegister)
    if (hipHostGetDevicePointer(&mapped, const_cast<uint8_t*>(src), 0) == hipSuccess && mapped) {
        dev_view = mapped;
    } else {
        (void)hipGetLastError();
    }
    rc = mbx_stage_in(sl.d_frames, dev_view, nf * s->frame_bytes, s->s_comp);
    if (rc < 0) {
        return rc;
    }
    if (index) {
        rc = pinned(sl.h_index, (size_t)s->streams);   // always staged: 4 B per stream, and the caller may reuse its array at once
        if (rc < 0) {
            return rc;
        }
        memcpy(sl.h_index, index, (size_t)n * sizeof(int32_t));
        rc = mbx_stage_in(sl.d_index, sl.h_index, (size_t)n * sizeof(int32_t), s->s_comp);
        if (rc < 0) {
            return rc;
        }
    }
    // ---- compute ----
    int16_t* d16 = pcm16 ? sl.d_pcm16 : nullptr;
    float*   dfl = pcmf ? sl.d_pcmf : nullptr;
    mbe_process_result* dres = results ? sl.d_results : nullptr;
    if (s->d_resident) {   // the session owns the state between submits: the resident form (no prev_mp_enhanced traffic, lazy prev_mp)
        rc = mbx_process_batch_resident(s->codec, n, T, index ? sl.d_index : nullptr, sl.d_frames, s->d_state, s->d_resident, s->d_rng,
                                        d16, dfl, dres, sl.d_records, s->s_comp);
    } else if (index) {
        rc = mbx_process_batch_indexed(s->codec, n, T, sl.d_index, sl.d_frames, s->d_state, s->d_rng, d16, dfl, dres, sl.d_records,
                                       s->s_comp);
    } else {
        rc = mbx_process_batch_ws(s->codec, n, T, sl.d_frames, s->d_state, s->d_rng, d16, dfl, dres, sl.d_records, sl.d_workspace,
                                  mbx_workspace_bytes(s->max_frames), s->s_comp);
    }
    if (rc < 0) {
        return rc;
    }
    S_TRY(hipEventRecord(sl.comp, s->s_comp));
    // ---- out ----
    S_TRY(hipStreamWaitEvent(s->s_out, sl.comp, 0));
    sl.nout = 0;
    auto out = [&](void* user, const void* dev, auto*& stage, size_t per_frame) -> int {   // per_frame elements of *stage per frame
        if (!user) {
            return 0;
        }
        const size_t bytes = nf * per_frame * sizeof(*stage);
        void* dst = user;
        if (!is_pinned(user)) {
            int r = pinned(stage, s->max_frames * per_frame);
            if (r < 0) {
                return r;
            }
            dst = stage;
            sl.out[sl.nout++] = CopyOut{user, stage, bytes};
        }
        hipError_t e = hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, s->s_out);
        return e == hipSuccess ? 0 : sfail("hipMemcpyAsync (device to host)", e);
    };
    if ((rc = out(pcm16, sl.d_pcm16, sl.h_pcm16, 160)) < 0 || (rc = out(pcmf, sl.d_pcmf, sl.h_pcmf, 160)) < 0
        || (rc = out(results, sl.d_results, sl.h_results, 1)) < 0 || (rc = out(records, sl.d_records, sl.h_records, 1)) < 0) {
        return rc;
    }
    S_TRY(hipEventRecord(sl.done, s->s_out));
    return 0;
}

}  // namespace

extern "C" {

void* mbx_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void mbx_host_free(void* p) {
    if (p) {
        (void)hipHostFree(p);
    }
}

int mbx_session_create(mbx_session** out, int codec, int streams, size_t max_frames_per_submit, unsigned outputs) {
    if (!out || streams <= 0 || max_frames_per_submit == 0 || codec < MBX_CODEC_IMBE7200X4400 || codec > MBX_CODEC_AMBE3600X2400
        || (outputs & ~(MBX_SESSION_PCM16 | MBX_SESSION_PCMF | MBX_SESSION_RESULTS)) != 0u) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int dev = -1;
    S_TRY(hipGetDevice(&dev));
    if (!mbx_device_ready(dev)) {
        mbx_set_error_text("mbx_session_create: mbx_init() has not been called for the current device");
        return MBX_ENOTINIT;
    }
    mbx_session* s = new (std::nothrow) mbx_session();
    if (!s) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    s->device = dev;
    s->codec = codec;
    s->streams = streams;
    s->max_frames = max_frames_per_submit;
    s->outputs = outputs ? outputs : MBX_SESSION_PCM16;
    s->frame_bytes = (codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400) ? MBX_AMBE_FRAME_BYTES : MBX_IMBE_FRAME_BYTES;
    *out = s;
    auto bail = [&](int rc) {
        mbx_session_destroy(s);
        *out = nullptr;
        return rc;
    };
#define C_TRY(expr)                          \
    do {                                     \
        hipError_t e_ = (expr);              \
        if (e_ != hipSuccess) {              \
            return bail(sfail(#expr, e_));   \
        }                                    \
    } while (0)
    C_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_state), (size_t)streams * 3 * sizeof(mbe_parms)));
    C_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_rng), (size_t)streams * sizeof(mbx_stream_rng)));
    if (!getenv("MBX_SESSION_ABI_STATE")) {   // (development switch: keep the ABI triplets whole after every submit, for A/B timing)
        C_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_resident), (size_t)streams * sizeof(uint32_t)));
        C_TRY(hipMemset(s->d_resident, 0, (size_t)streams * sizeof(uint32_t)));
    }
    C_TRY(hipStreamCreateWithFlags(&s->s_comp, hipStreamNonBlocking));
    C_TRY(hipStreamCreateWithFlags(&s->s_out, hipStreamNonBlocking));
    const size_t mf = s->max_frames;
    for (Slot& sl : s->slot) {
        C_TRY(hipMalloc(reinterpret_cast<void**>(&sl.d_frames), mf * s->frame_bytes));
        C_TRY(hipMalloc(reinterpret_cast<void**>(&sl.d_records), mf * sizeof(mbx_param_record)));
        C_TRY(hipMalloc(reinterpret_cast<void**>(&sl.d_index), (size_t)streams * sizeof(int32_t)));
        if (!s->d_resident) {   // (the resident entry point uses the workspace the launcher keeps for the compute stream, reserved below)
            C_TRY(hipMalloc(&sl.d_workspace, mbx_workspace_bytes(mf)));
        }
        if (s->outputs & MBX_SESSION_PCM16) {
            C_TRY(hipMalloc(reinterpret_cast<void**>(&sl.d_pcm16), mf * 160 * sizeof(int16_t)));
        }
        if (s->outputs & MBX_SESSION_PCMF) {
            C_TRY(hipMalloc(reinterpret_cast<void**>(&sl.d_pcmf), mf * 160 * sizeof(float)));
        }
        if (s->outputs & MBX_SESSION_RESULTS) {
            C_TRY(hipMalloc(reinterpret_cast<void**>(&sl.d_results), mf * sizeof(mbe_process_result)));
        }
        C_TRY(hipEventCreateWithFlags(&sl.comp, hipEventDisableTiming));
        // (a blocking wait -- hipEventBlockingSync -- was measured here and costs more than it saves: the wake-up latency of every
        // submit that has to wait is exposed, 155 -> 113 M frames/s with two session threads, 117 -> 78 M with four)
        C_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
#undef C_TRY
    // the indexed launches size the compute stream's own workspace once, here, not in the first submit
    int rc = mbx_reserve_stream(s->s_comp, mf);
    if (rc < 0) {
        return bail(rc);
    }
    rc = mbx_session_reset(s, 0, streams);
    return rc < 0 ? bail(rc) : 0;
}

int mbx_session_destroy(mbx_session* s) {
    if (!s) {
        return 0;
    }
    DeviceGuard guard(s->device);
    if (s->s_comp) {
        (void)hipStreamSynchronize(s->s_comp);
        (void)mbx_release_stream(s->s_comp);
    }
    if (s->s_out) {
        (void)hipStreamSynchronize(s->s_out);
    }
    for (Slot& sl : s->slot) {
        (void)hipFree(sl.d_frames);
        (void)hipFree(sl.d_records);
        (void)hipFree(sl.d_pcm16);
        (void)hipFree(sl.d_pcmf);
        (void)hipFree(sl.d_results);
        (void)hipFree(sl.d_index);
        (void)hipFree(sl.d_workspace);
        (void)hipHostFree(sl.h_in);
        (void)hipHostFree(sl.h_index);
        (void)hipHostFree(sl.h_pcm16);
        (void)hipHostFree(sl.h_pcmf);
        (void)hipHostFree(sl.h_results);
        (void)hipHostFree(sl.h_records);
        if (sl.comp) {
            (void)hipEventDestroy(sl.comp);
        }
        if (sl.done) {
            (void)hipEventDestroy(sl.done);
        }
    }
    (void)hipFree(s->d_state);
    (void)hipFree(s->d_resident);
    (void)hipFree(s->d_rng);
    if (s->s_comp) {
        (void)hipStreamDestroy(s->s_comp);
    }
    if (s->s_out) {
        (void)hipStreamDestroy(s->s_out);
    }
    (void)hipGetLastError();
    delete s;
    return 0;
}

int mbx_session_streams(const mbx_session* s) { return s ? s->streams : 0; }

int mbx_session_wait(mbx_session* s) {
    if (!s) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    DeviceGuard guard(s->device);
    for (int i = 0; i < kDepth; ++i) {   // oldest first: staged outputs are handed over in submission order
        int rc = finish_slot(s->slot[(s->submitted + i) % kDepth]);
        if (rc < 0) {
            return rc;
        }
    }
    return 0;
}

int mbx_session_set_state(mbx_session* s, int first, int count, const mbe_parms* state, const mbx_stream_rng* rng) {
    if (!range_ok(s, first, count) || (!state && !rng)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = mbx_session_wait(s);
    if (rc < 0) {
        return rc;
    }
    DeviceGuard guard(s->device);
    S_TRY(hipStreamSynchronize(s->s_comp));   // (a submit that failed halfway leaves no busy slot to wait on)
    if (state) {
        S_TRY(hipMemcpy(s->d_state + 3 * (size_t)first, state, (size_t)count * 3 * sizeof(mbe_parms), hipMemcpyHostToDevice));
        if (s->d_resident) {   // the caller's triplets are whole: nothing is elided any more
            S_TRY(hipMemset(s->d_resident + first, 0, (size_t)count * sizeof(uint32_t)));
        }
    }
    if (rng) {
        S_TRY(hipMemcpy(s->d_rng + first, rng, (size_t)count * sizeof(mbx_stream_rng), hipMemcpyHostToDevice));
    }
    return 0;
}

int mbx_session_get_state(mbx_session* s, int first, int count, mbe_parms* state, mbx_stream_rng* rng) {
    if (!range_ok(s, first, count) || (!state && !rng)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = mbx_session_wait(s);
    if (rc < 0) {
        return rc;
    }
    DeviceGuard guard(s->device);
    S_TRY(hipStreamSynchronize(s->s_comp));
    if (state) {
        if (s->d_resident) {   // write the elided prev_mp_enhanced structs out: the triplets leave in their ABI form
            rc = mbx_resident_materialize(count, nullptr, s->d_state + 3 * (size_t)first, s->d_resident + first, s->s_comp);
            if (rc < 0) {
                return rc;
            }
            S_TRY(hipStreamSynchronize(s->s_comp));
        }
        S_TRY(hipMemcpy(state, s->d_state + 3 * (size_t)first, (size_t)count * 3 * sizeof(mbe_parms), hipMemcpyDeviceToHost));
    }
    if (rng) {
        S_TRY(hipMemcpy(rng, s->d_rng + first, (size_t)count * sizeof(mbx_stream_rng), hipMemcpyDeviceToHost));
    }
    return 0;
}

int mbx_session_reset(mbx_session* s, int first, int count) {
    if (!range_ok(s, first, count)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (count == 0) {
        return 0;
    }
    mbe_parms p;
    default_parms(p);
    std::vector<mbe_parms> st((size_t)count * 3, p);
    std::vector<mbx_stream_rng> rg((size_t)count);
    for (auto& r : rg) {
        mbx_rng_default(&r);
    }
    return mbx_session_set_state(s, first, count, st.data(), rg.data());
}

int mbx_session_seed(mbx_session* s, int first, int count, const uint32_t* seeds) {
    if (!range_ok(s, first, count) || !seeds) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    std::vector<mbx_stream_rng> rg((size_t)count);
    for (int i = 0; i < count; ++i) {
        mbx_rng_default(&rg[(size_t)i]);
        mbx_rng_seed(&rg[(size_t)i], seeds[i]);
    }
    return mbx_session_set_state(s, first, count, nullptr, rg.data());
}

int mbx_session_submit(mbx_session* s, int T, const uint8_t* frames, int16_t* pcm16, float* pcmf, mbe_process_result* results) {
    return submit(s, s ? s->streams : 0, T, nullptr, frames, pcm16, pcmf, results, nullptr);
}

int mbx_session_submit_indexed(mbx_session* s, int n, int T, const int32_t* stream_index, const uint8_t* frames, int16_t* pcm16,
                               float* pcmf, mbe_process_result* results, mbx_param_record* records) {
    if (!stream_index) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    return submit(s, n, T, stream_index, frames, pcm16, pcmf, results, records);
}

}  // extern "C"
