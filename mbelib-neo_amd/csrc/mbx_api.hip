// mbx_api.hip -- the C-ABI of libmbx_hip.so (declared in include/mbx.h): table upload,
// launchers, host-buffer conveniences.  No CPU compute path exists here: if HIP is not
// usable every entry point fails with MBX_ENODEVICE.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "mbx.h"
#include "mbx_device.h"

namespace mbx {
// kernels (mbx_fec.hip, mbx_stream.hip)
__global__ void fec_imbe7200x4400_kernel(const uint8_t*, size_t, mbx_param_record*, DeviceTables);
__global__ void fec_ambe3600x2450_kernel(const uint8_t*, size_t, mbx_param_record*, DeviceTables);
__global__ void fec_imbe7100x4400_kernel(const uint8_t*, size_t, mbx_param_record*, DeviceTables);
__global__ void floattoshort_kernel(const float*, int16_t*, size_t);
__global__ void result_histogram_kernel(const mbe_process_result*, size_t, unsigned long long*);
__global__ void stage_in_kernel(const uint8_t*, uint8_t*, size_t);
__global__ void expand_imbe_kernel(const mbx_param_record*, size_t, FrameParams*, DeviceTables);
__global__ void expand_ambe_kernel(const mbx_param_record*, size_t, FrameParams*, DeviceTables);
__global__ void expand_ambe2400_kernel(const mbx_param_record*, size_t, FrameParams*, DeviceTables);
__global__ void imbe_stream_kernel(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                   int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void frame_server_kernel(mbx_frame_mailbox*, unsigned, mbe_parms*, mbx_stream_rng*, int16_t*, float*, mbe_process_result*,
                                    mbx_param_record*, DeviceTables, FrameShadow);
__global__ void imbe_stream_kernel_one(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                       int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe_stream_kernel_one(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                       int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe2400_stream_kernel_one(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                           int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void imbe_stream_kernel_one_fused(int, int, const uint8_t*, mbx_param_record*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                             mbe_process_result*, DeviceTables);
__global__ void imbe_stream_kernel_res1_fused(int, int, const uint8_t*, mbx_param_record*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                              mbe_process_result*, DeviceTables);
#ifdef MBX_EXP_PAIR
__global__ void imbe_stream_kernel_lds_pairexp(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                               mbe_process_result*, DeviceTables);
#endif
__global__ void imbe_stream_kernel_lds_slice(int, int, int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                             mbe_process_result*, DeviceTables);
__global__ void ambe_stream_kernel_lds_slice(int, int, int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                             mbe_process_result*, DeviceTables);
__global__ void ambe2400_stream_kernel_lds_slice(int, int, int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*, int16_t*,
                                                 float*, mbe_process_result*, DeviceTables);
__global__ void imbe_one_launch_kernel(int, int, const uint8_t*, mbx_param_record*, FrameParams*, uint32_t*, uint32_t*, uint32_t, mbe_parms*,
                                       mbx_stream_rng*, int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void imbe_one_launch_kernel_res(int, int, const uint8_t*, mbx_param_record*, FrameParams*, uint32_t*, uint32_t*, uint32_t, mbe_parms*,
                                           mbx_stream_rng*, int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe_one_launch_kernel(int, const uint8_t*, mbx_param_record*, FrameParams*, uint32_t*, uint32_t*, uint32_t, mbe_parms*, mbx_stream_rng*,
                                       int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe_one_launch_kernel_res(int, const uint8_t*, mbx_param_record*, FrameParams*, uint32_t*, uint32_t*, uint32_t, mbe_parms*,
                                           mbx_stream_rng*, int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe2400_one_launch_kernel_res(int, const uint8_t*, mbx_param_record*, FrameParams*, uint32_t*, uint32_t*, uint32_t, mbe_parms*,
                                               mbx_stream_rng*, int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe2400_one_launch_kernel(int, const uint8_t*, mbx_param_record*, FrameParams*, uint32_t*, uint32_t*, uint32_t, mbe_parms*,
                                           mbx_stream_rng*, int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void imbe7100_stream_kernel_one_fused(int, int, const uint8_t*, mbx_param_record*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                                 mbe_process_result*, DeviceTables);
__global__ void imbe7100_stream_kernel_res1_fused(int, int, const uint8_t*, mbx_param_record*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                                  mbe_process_result*, DeviceTables);
__global__ void imbe_stream_kernel_res1(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                        int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe_stream_kernel_res1(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                        int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe2400_stream_kernel_res1(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                            int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void imbe_stream_kernel_res(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                       int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe_stream_kernel_res(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                       int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe2400_stream_kernel_res(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                           int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void imbe_stream_kernel_lds(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                       int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe_stream_kernel_lds(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                       int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe2400_stream_kernel_lds(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                           int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe_stream_kernel(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                   int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void ambe2400_stream_kernel(int, int, const mbx_param_record*, const FrameParams*, mbe_parms*, mbx_stream_rng*,
                                   int16_t*, float*, mbe_process_result*, DeviceTables);
__global__ void imbe_frame_kernel(int, const uint8_t*, mbx_param_record*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                  mbe_process_result*, uint32_t*, uint32_t, DeviceTables, FrameShadow);
__global__ void ambe_frame_kernel(const uint8_t*, mbx_param_record*, mbe_parms*, mbx_stream_rng*, int16_t*, float*, mbe_process_result*,
                                  uint32_t*, uint32_t, DeviceTables, FrameShadow);
__global__ void ambe2400_frame_kernel(const uint8_t*, mbx_param_record*, mbe_parms*, mbx_stream_rng*, int16_t*, float*,
                                      mbe_process_result*, uint32_t*, uint32_t, DeviceTables, FrameShadow);
__global__ void synth_speech_kernel(int, mbe_parms*, mbe_parms*, mbx_stream_rng*, float*, int16_t*, DeviceTables);
__global__ void enhance_kernel(int, mbe_parms*);
__global__ void smoothing_kernel(int, mbe_parms*, const mbe_parms*);
__global__ void comfort_noise_kernel(int, mbx_stream_rng*, float*, int16_t*);
__global__ void state_copy_kernel(int, mbe_parms*);
__global__ void tone_kernel(int, const mbx_param_record*, const int32_t*, mbe_parms*, float*, int16_t*, int);
__global__ void ecc_words_kernel(int, const uint32_t*, size_t, uint32_t*, int32_t*, DeviceTables);
__global__ void pack_cells_kernel(int, const char*, size_t, uint8_t*, int32_t*);
__global__ void fec_stage_kernel(int, int, const uint8_t*, size_t, uint8_t*, mbx_param_record*, DeviceTables);
__global__ void decode_parms_kernel(int, int, const FrameParams*, mbe_parms*, mbe_parms*, int32_t*, DeviceTables);
__global__ void fec_imbe7200x4400_soft_kernel(const mbe_soft_bit*, size_t, mbx_param_record*, DeviceTables);
__global__ void fec_ambe3600x2450_soft_kernel(const mbe_soft_bit*, size_t, mbx_param_record*, DeviceTables);
__global__ void fec_imbe7100x4400_soft_kernel(const mbe_soft_bit*, size_t, mbx_param_record*, DeviceTables);
__global__ void ecc_soft_words_kernel(int, const mbe_soft_bit*, size_t, uint32_t*, int32_t*, DeviceTables);
}  // namespace mbx

namespace {

// ---- per-device contexts ---------------------------------------------------------------------------------------
// One Context per HIP device, created by mbx_init(device, ...).  Every launcher works on the context of the calling
// thread's CURRENT device (hipGetDevice), like any other HIP call, so one process can drive all eight GPUs of a node:
// mbx_init() each device once, then hipSetDevice(d) before launching on d.  The reference keeps its mutable helper
// state thread-local and is re-entrant per stream (ref include/mbelib-neo/mbelib.h:28-30); here the only mutable
// launcher state is the per-(device, hipStream_t) StreamSlot below, guarded by the context's mutex, so any number of
// host threads may launch concurrently on their own streams (or on a shared one: launches of one call stay together).
constexpr int kMaxDevices = 32;

struct StreamSlot {                      // what a hipStream_t owns inside a context
    mbx::FrameParams* workspace = nullptr;   // expand-stage output of launches on this stream; grow-only
    size_t            frames = 0;
    uint32_t*         flags = nullptr;       // one-launch T = 1 step: ready word per chunk of eight rows (behind the rows, same allocation)
    uint32_t          epoch = 0;             // ... == the epoch of the launch that wrote them; a new value every launch
    hipStream_t       side[4] = {nullptr, nullptr, nullptr, nullptr};   // sliced launches: the internal streams the groups' slices are issued on ...
    hipEvent_t        fork = nullptr, join[4] = {nullptr, nullptr, nullptr, nullptr};   // ... and the events that hang them between the caller's launches
    unsigned          launches = 0;          // parity = direction in which the next stream-kernel launch walks the streams
    int               exp_codec = -1;        // what mbx_expand_records() last left in the workspace
    size_t            exp_n = 0;
    const void*       exp_records = nullptr;
};

struct Context {
    std::mutex         mu;                   // guards slots / reserve_frames; held across the launches of one call
    std::atomic<bool>  ready{false};
    int                device = -1;
    mbx::DeviceTables  tabs{nullptr, nullptr, 0, 0};
    void*              d_blob = nullptr;
    void*              d_derived = nullptr;
    uint32_t           checksum = 0;
    int                simds = 0;            // 4 per CU
    size_t             reserve_frames = 0;   // mbx_reserve(): minimum workspace of every slot
    std::unordered_map<void*, StreamSlot> slots;
};
Context    g_ctx[kMaxDevices];
std::mutex g_init_mu;                        // serialises mbx_init() / mbx_shutdown()

thread_local char t_err[256] = "";           // mbx_last_error() is per thread, like errno

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    if (e != hipSuccess) {
        snprintf(t_err, sizeof(t_err), "%s: %s", what, hipGetErrorString(e));
    } else {
        snprintf(t_err, sizeof(t_err), "%s", what);
    }
    return code;
}

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) {                        \
            return fail(MBX_ENODEVICE, #expr, e_);     \
        }                                              \
    } while (0)

// context of the calling thread's current device, or nullptr with *rc set
Context* current_ctx(int* rc) {
    int dev = -1;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        *rc = fail(MBX_ENODEVICE, "hipGetDevice", e);
        return nullptr;
    }
    Context& c = g_ctx[dev];
    if (!c.ready.load(std::memory_order_acquire)) {
        *rc = fail(MBX_ENOTINIT, "mbx_init() has not been called for the current device");
        return nullptr;
    }
    return &c;
}

#define REQUIRE_CTX(c)                 \
    int rc_ctx_ = 0;                   \
    Context* c = current_ctx(&rc_ctx_); \
    if (!c) {                          \
        return rc_ctx_;                \
    }

void release_side_streams(StreamSlot& slot) {
    for (int g = 0; g < 4; ++g) {
        if (slot.side[g]) {
            (void)hipStreamSynchronize(slot.side[g]);
            (void)hipStreamDestroy(slot.side[g]);
            slot.side[g] = nullptr;
        }
        if (slot.join[g]) {
            (void)hipEventDestroy(slot.join[g]);
            slot.join[g] = nullptr;
        }
    }
    if (slot.fork) {
        (void)hipEventDestroy(slot.fork);
        slot.fork = nullptr;
    }
}

void free_context(Context& c) {   // caller holds g_init_mu and c.mu
    (void)hipFree(c.d_blob);
    (void)hipFree(c.d_derived);
    for (auto& kv : c.slots) {
        (void)hipFree(kv.second.workspace);
        release_side_streams(kv.second);
    }
    c.slots.clear();
    c.d_blob = c.d_derived = nullptr;
    c.tabs = mbx::DeviceTables{};   // (every field zero: mbx_init sets what it needs)
    c.reserve_frames = 0;
    c.checksum = 0;
    c.device = -1;
}

int lds_min_frames() {   // frames per stream from which the LDS-resident instances are used (MBX_LDS_MIN_FRAMES: A/B timing)
    static const int v = [] {
        const char* e = getenv("MBX_LDS_MIN_FRAMES");
        const int n = e ? atoi(e) : 4;
        return (n >= 1 && n <= 1 << 20) ? n : 4;   // anything else (0, negative, not a number) is ignored
    }();
    return v;
}
#define kLdsResidentMinFrames lds_min_frames()
bool res1_enabled() {   // MBX_NO_RES1 (A/B timing): resident one-frame launches through the LDS-resident instance; read once
    static const bool on = getenv("MBX_NO_RES1") == nullptr;
    return on;
}

bool lds_resident_enabled() {
    static const bool on = getenv("MBX_NO_LDS_RESIDENT") == nullptr;   // development switch for A/B timing; read once
    return on;
}

// Walking order of successive stream-kernel launches (see launch_stream): alternating by default; MBX_NO_REVERSE=1 in the
// environment or mbx_set_stream_order(0) fixes it (every launch walks the streams forward).
std::atomic<int>& stream_order_flag() {
    static std::atomic<int> flag{getenv("MBX_NO_REVERSE") == nullptr ? 1 : 0};   // thread-safe static initialisation
    return flag;
}
bool reverse_enabled() { return stream_order_flag().load(std::memory_order_relaxed) != 0; }
// 1: AMBE tone frames are synthesised (the reference's default build); MBX_DISABLE_TONES=1 in the environment starts with 0
std::atomic<int>& tones_flag() {
    static std::atomic<int> flag{[] {
        const char* e = getenv("MBX_DISABLE_TONES");
        return (e && e[0] == '1') ? 0 : 1;
    }()};
    return flag;
}

uint32_t fnv1a(const uint8_t* p, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; ++i) {
        h = (h ^ p[i]) * 16777619u;
    }
    return h;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(MBX_ENODEVICE, what, e);
}

int validate_bits(const char* bits, size_t count) {   // ref: src/internal/mbe_result.h:18-29
    if (!bits) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    // eight cells at a time: a cell is valid iff no bit but bit 0 is set
    uint64_t acc = 0;
    size_t i = 0;
    for (; i + 8 <= count; i += 8) {
        uint64_t v;
        memcpy(&v, bits + i, 8);
        acc |= v;
    }
    for (; i < count; ++i) {
        acc |= (uint8_t)bits[i];
    }
    return (acc & 0xfefefefefefefefeULL) ? MBE_STATUS_INVALID_BITS : 0;
}

// rows of 0/1 cells -> wire bits: row r contributes cells width[r]-1 .. 0.  Eight cells become eight bits with one
// multiplication (byte i of x lands on bit 56 + i of x * 0x0102040810204080), three of those cover a row of up to 24 cells.
void pack_rows(const char* cells, int rows, int stride, const int* width, uint8_t* out, int nbytes) {
    char pad[200];   // the last row is read eight cells at a time: give it room (frames are at most 184 cells)
    const int ncell = rows * stride;
    memcpy(pad, cells, (size_t)ncell);
    memset(pad + ncell, 0, sizeof(pad) - (size_t)ncell);
    uint64_t acc = 0;   // bit accumulator, filled from the top
    int have = 0, o = 0;
    for (int r = 0; r < rows; ++r) {
        const char* row = pad + r * stride;
        uint32_t v = 0;   // bit j = cell j
        for (int k = 0; k < 3; ++k) {
            uint64_t x;
            memcpy(&x, row + 8 * k, 8);
            v |= (uint32_t)(((x & 0x0101010101010101ULL) * 0x0102040810204080ULL) >> 56) << (8 * k);
        }
        const int w = width[r];
        v &= (w >= 32) ? 0xffffffffu : ((1u << w) - 1u);
        acc |= (uint64_t)v << (64 - have - w);   // cell w-1 first
        have += w;
        while (have >= 8) {
            out[o++] = (uint8_t)(acc >> 56);
            acc <<= 8;
            have -= 8;
        }
    }
    if (have > 0 && o < nbytes) {
        out[o++] = (uint8_t)(acc >> 56);
    }
    while (o < nbytes) {
        out[o++] = 0;
    }
}

}  // namespace

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) {
            (void)hipFree(p);
        }
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class U> U* as() { return static_cast<U*>(p); }
};
}  // namespace

void mbx_set_error_text(const char* text) { snprintf(t_err, sizeof(t_err), "%s", text); }   // for mbx_session.hip

extern "C" {

const char* mbx_last_error(void) { return t_err; }

int mbx_init(int device, const void* table_blob, size_t table_bytes) {
    if (!table_blob || table_bytes != sizeof(mbx_tables)) {
        return fail(MBX_EBADTABLE, "table blob: wrong size");
    }
    const mbx_tables* host = static_cast<const mbx_tables*>(table_blob);
    if (host->magic != MBX_TABLES_MAGIC || host->version != MBX_TABLES_VERSION || host->total_bytes != sizeof(mbx_tables)) {
        return fail(MBX_EBADTABLE, "table blob: wrong magic/version");
    }
    const uint32_t sum = fnv1a(reinterpret_cast<const uint8_t*>(&host->checksum) + 4, sizeof(mbx_tables) - 16);
    if (sum != host->checksum) {
        return fail(MBX_EBADTABLE, "table blob: checksum mismatch");
    }
    // the voiced-bank kernel relies on the shape of the synthesis window (mbx_stream.hip): zero / ramp / one / ramp / zero
    for (int k = 0; k < 321; ++k) {
        const float v = host->ws[k];
        const bool ok = (k <= 55 || k >= 265) ? (v == 0.0f) : ((k >= 105 && k <= 215) ? (v == 1.0f) : (v > 0.0f && v < 1.0f));
        if (!ok) {
            return fail(MBX_EBADTABLE, "table blob: unexpected synthesis window shape");
        }
    }
    // the IMBE expansion scatters payload bits without looking at the entries again (mbx_expand_imbe.h): word 0..57, bit 0..11
    for (int l9 = 0; l9 < 48; ++l9) {
        for (int i = 0; i < 79; ++i) {
            if (host->imbe_bo[l9][i][0] >= 58 || host->imbe_bo[l9][i][1] >= 12) {
                return fail(MBX_EBADTABLE, "table blob: IMBE bit-layout entry out of range");
            }
        }
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        return fail(MBX_ENODEVICE, "no HIP device");
    }
    if (device < 0 || device >= count || device >= kMaxDevices) {
        return fail(MBX_ENODEVICE, "device index out of range");
    }
    std::lock_guard<std::mutex> init_lock(g_init_mu);
    HIP_TRY(hipSetDevice(device));   // the calling thread's current device from here on, like hipSetDevice itself
    Context& ctx = g_ctx[device];
    if (ctx.ready.load(std::memory_order_acquire)) {
        if (ctx.checksum == host->checksum) {
            return 0;   // same tables already resident: nothing to do (any thread may call mbx_init again)
        }
        // different tables: the caller must have no launches in flight on this device
        HIP_TRY(hipDeviceSynchronize());
        std::lock_guard<std::mutex> lock(ctx.mu);
        ctx.ready.store(false, std::memory_order_release);
        const int keep = ctx.tabs.ablate;
        free_context(ctx);
        ctx.tabs.ablate = keep;
    }

    // derived tables
    std::vector<mbx::DerivedTables> dv(1);
    mbx::DerivedTables& d = dv[0];
    memset(&d, 0, sizeof(d));
    uint32_t a = 1, c = 0;   // x_k = a*x_0 + c (mod 53125)
    for (int k = 0; k <= 160; ++k) {
        d.lcg_mul[k] = a;
        d.lcg_add[k] = c;
        d.lcg_pack[k] = a | (c << 16);
        a = (uint32_t)(((uint64_t)a * 171u) % 53125u);
        c = (uint32_t)(((uint64_t)c * 171u + 11213u) % 53125u);
    }
    for (int L = 1; L < 64; ++L) {
        d.log2_int[L] = log2f((float)L);
    }
    for (int i = 0; i < 12; ++i) {
        const uint32_t g = host->golay_gen[i];
        d.golay_rot[i] = ((g & 0x3fu) << 1) | ((g >> 6) & 1u) | (g & 0x780u);
    }
    for (int p = 0; p < 57; ++p) {
        for (int c = 1; c < 57; ++c) {
            d.l_ratio[p][c] = (float)p / (float)c;
        }
    }
    for (int L = 1; L < 57; ++L) {
        const float rho = (L <= 15) ? 0.4f : ((L <= 24) ? ((0.03f * (float)L) - 0.05f) : 0.7f);
        d.imbe_rho_over_l[L] = rho / (float)L;
        d.ambe_pred_over_l[L] = (float)0.65 / (float)L;
    }
    for (int n = 0; n < 192; ++n) {
        d.nfrac[n] = (float)n / (float)160;
    }
    for (int l9 = 0; l9 < 48; ++l9) {   // who owns what in the IMBE parameter expansion (ref src/imbe/imbe7200x4400.c:233-270)
        const uint8_t* J = host->imbe_ji[l9];
        const int L = l9 + 9;
        for (int lane = 0; lane < 64; ++lane) {
            int hblk = 1, first = 0, ji = J[0];   // higher-order coefficient of word lane + 8
            for (int q = 1; q < 6; ++q) {
                if (lane >= first + (ji - 1)) {
                    first += ji - 1;
                    hblk = q + 1;
                    ji = J[q];
                }
            }
            const int hk = lane - first + 2;
            int iblk = 1, ifirst = 1, iji = J[0];   // harmonic `lane`
            for (int q = 1; q < 6; ++q) {
                if (lane >= ifirst + iji) {
                    ifirst += iji;
                    iblk = q + 1;
                    iji = J[q];
                }
            }
            const int ij = lane - ifirst + 1;
            const bool harm = lane >= 1 && lane <= L && iji >= 1 && iji <= 10 && ij >= 1 && ij <= 10;
            d.imbe_lane_map[l9][lane] = (uint32_t)hblk | ((uint32_t)(hk & 15) << 3) | ((uint32_t)iblk << 7)
                                        | ((uint32_t)(iji & 15) << 10) | ((uint32_t)(ij & 15) << 14);
            d.imbe_hoc_sd[l9][lane] = (hk >= 2 && hk <= 10) ? host->imbe_standdev[hk - 2] : 0.0f;
            for (int k = 1; k <= 10; ++k) {
                d.imbe_idct_rows[l9][lane][k - 1] = harm ? host->imbe_idct_cos[iji][ij][k] : 0.0f;
            }
        }
    }
    for (int b0 = 0; b0 < 208; ++b0) {
        uint32_t wbits;
        memcpy(&wbits, &host->imbe_w0[b0], 4);
        d.imbe_b0[b0] = make_uint2(wbits, (uint32_t)host->imbe_L[b0] | ((uint32_t)host->imbe_K[b0] << 8));
    }
    for (int j = 0; j < 64; ++j) {
        uint32_t hi = 0, lo = 0;
        for (int i = 0; i < 6; ++i) {
            if ((j >> (5 - i)) & 1) {
                hi ^= host->golay_gen[i];
                lo ^= host->golay_gen[6 + i];
            }
        }
        d.golay_half_syn[j] = (hi << 16) | lo;
        uint32_t ac[2];
        for (int half = 0; half < 2; ++half) {   // x -> 173 x + 13849 mod 2^16, k = j + 1 + 64 half times
            uint32_t a = 1, c = 0;
            for (int k = 0; k < j + 1 + 64 * half; ++k) {
                a = (a * 173u) & 0xffffu;
                c = (c * 173u + 13849u) & 0xffffu;
            }
            ac[half] = a | (c << 16);
        }
        d.pr_lane[j] = make_uint2(ac[0], ac[1]);
    }
    memset(d.pr_bits, 0, sizeof(d.pr_bits));
    for (uint32_t seed = 0; seed < 4096; ++seed) {
        uint32_t x = (16u * seed) & 0xffffu;
        for (int k = 0; k < 114; ++k) {
            x = (173u * x + 13849u) & 0xffffu;
            d.pr_bits[seed][k >> 5] |= (x >> 15) << (31 - (k & 31));
        }
    }
    memset(d.imbe_L_lanes, 0, sizeof(d.imbe_L_lanes));
    for (int b0 = 0; b0 < 208; ++b0) {
        d.imbe_L_lanes[b0 & 63] |= (uint32_t)host->imbe_L[b0] << (8 * (b0 >> 6));
    }
    memset(d.imbe_len_rows, 0, sizeof(d.imbe_len_rows));
    for (int ji = 1; ji <= 10; ++ji) {
        for (int j = 1; j <= ji; ++j) {
            for (int k = 1; k <= ji; ++k) {
                d.imbe_len_rows[ji][j - 1][k - 1] = host->imbe_idct_cos[ji][j][k];
            }
        }
    }
    memset(d.imbe_blk_info, 0, sizeof(d.imbe_blk_info));
    memset(d.imbe_blk_bm, 0, sizeof(d.imbe_blk_bm));
    memset(d.imbe_blk_step, 0, sizeof(d.imbe_blk_step));
    for (int l9 = 0; l9 < 48; ++l9) {   // per block: where its words / harmonics start and how its coefficients are quantised
        int m = 8, l = 1;
        for (int blk = 1; blk <= 6; ++blk) {
            const int ji = host->imbe_ji[l9][blk - 1];
            d.imbe_blk_info[l9][blk] = (uint32_t)m | ((uint32_t)l << 8) | ((uint32_t)ji << 16);
            for (int k = 2; k <= ji && k <= 10; ++k) {
                const int Bm = (m - 8 < 50) ? host->imbe_hoba[l9][m - 8] : 0;
                d.imbe_blk_bm[l9][blk][k] = (uint8_t)Bm;
                d.imbe_blk_step[l9][blk][k] = (Bm > 0 && Bm <= 11) ? (host->imbe_quantstep[Bm - 1] * host->imbe_standdev[k - 2]) : 0.0f;
                ++m;
            }
            l += ji;
        }
    }
    for (int n = 0; n < 160; ++n) {
        d.wola_inv[n] = (host->wola_denom[n] > 1e-10f) ? (1.0f / host->wola_denom[n]) : 0.0f;
    }
    for (int b0 = 0; b0 < 128; ++b0) {   // ref src/ambe/ambe3600x2400.c:238 (same expression, host libm)
        d.ambep_f0[b0] = exp2f(-4.311767578125f - (2.1336e-2f * ((float)b0 + 0.5f)));
    }
    {   // x_k = 173 x_{k-1} + 13849 (mod 2^16)  =>  x_k = pr_mul[k] x_0 + pr_add[k]
        uint32_t m = 1u, a = 0u;
        for (int k = 0; k < 116; ++k) {
            d.pr_mul[k] = m;
            d.pr_add[k] = a;
            m = (173u * m) & 0xffffu;
            a = (173u * a + 13849u) & 0xffffu;
        }
    }
    for (int variant = 0; variant < 2; ++variant) {
        // code word of data bit i: data bits at positions {2,4,5,6,8..14} (7100x4400 mapping: {4..14}), parity
        // at {0,1,3,7} ({0,1,2,3}) chosen for a zero syndrome (ref src/ecc/ecc.c:128-155)
        static const int data_pos[2][11] = {{2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14}, {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14}};
        static const int parity_pos[2][4] = {{0, 1, 3, 7}, {0, 1, 2, 3}};
        const uint16_t* gen = variant ? host->hamming7100_gen : host->hamming_gen;
        for (int i = 0; i < 11; ++i) {
            uint32_t found = 0xffffffffu;
            for (uint32_t p = 0; p < 16u && found == 0xffffffffu; ++p) {
                uint32_t c = 1u << data_pos[variant][i];
                for (int q = 0; q < 4; ++q) {
                    c |= ((p >> q) & 1u) << parity_pos[variant][q];
                }
                int syndrome = 0;
                for (int q = 0; q < 4; ++q) {
                    syndrome |= (__builtin_popcount(c & gen[q]) & 1) << q;
                }
                if (syndrome == 0) {
                    found = c;
                }
            }
            if (found == 0xffffffffu) {
                return fail(MBX_EBADTABLE, "mbx_init: Hamming generator rows admit no code word for a data bit");
            }
            (variant ? d.ham7100_basis : d.ham_basis)[i] = found;
        }
    }

    std::lock_guard<std::mutex> lock(ctx.mu);
    HIP_TRY(hipMalloc(&ctx.d_blob, sizeof(mbx_tables)));
    HIP_TRY(hipMalloc(&ctx.d_derived, sizeof(mbx::DerivedTables)));
    HIP_TRY(hipMemcpy(ctx.d_blob, table_blob, sizeof(mbx_tables), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx.d_derived, &d, sizeof(d), hipMemcpyHostToDevice));
    ctx.tabs.t = static_cast<const mbx_tables*>(ctx.d_blob);
    ctx.tabs.tones_off = tones_flag().load(std::memory_order_relaxed) ? 0 : 1;
    ctx.tabs.d = static_cast<const mbx::DerivedTables*>(ctx.d_derived);
    ctx.device = device;
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    ctx.simds = 4 * cus;
    ctx.checksum = host->checksum;
    ctx.ready.store(true, std::memory_order_release);
    return 0;
}

void mbx_shutdown(void) {
    std::lock_guard<std::mutex> init_lock(g_init_mu);
    int before = -1;
    (void)hipGetDevice(&before);
    for (int dev = 0; dev < kMaxDevices; ++dev) {
        Context& ctx = g_ctx[dev];
        if (!ctx.ready.load(std::memory_order_acquire)) {
            continue;
        }
        (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lock(ctx.mu);
        ctx.ready.store(false, std::memory_order_release);
        free_context(ctx);
    }
    if (before >= 0) {
        (void)hipSetDevice(before);
    }
}

int mbx_set_stream_order(int alternate) { return stream_order_flag().exchange(alternate ? 1 : 0, std::memory_order_relaxed); }

// Tone synthesis on / off, process-wide: the run-time form of the reference's one compile-time option that changes what this
// path computes (NOTONES = -DDISABLE_AMBE_TONES, ref CMakeLists.txt:41,330-337; src/core/mbelib.c:747-751,815-819).  The flag
// lives in every context's DeviceTables (a kernel argument): launches issued after the call see the new value.
int mbx_set_tone_synthesis(int enabled) {
    const int before = tones_flag().exchange(enabled ? 1 : 0, std::memory_order_relaxed);
    for (int dev = 0; dev < kMaxDevices; ++dev) {
        std::lock_guard<std::mutex> lock(g_ctx[dev].mu);
        g_ctx[dev].tabs.tones_off = enabled ? 0 : 1;
    }
    return before;
}

int mbx_device_ready(int device) {
    return device >= 0 && device < kMaxDevices && g_ctx[device].ready.load(std::memory_order_acquire) ? 1 : 0;
}

uint32_t mbx_table_checksum(void) {
    int rc = 0;
    Context* c = current_ctx(&rc);
    return c ? c->checksum : 0u;
}

// Where cell [row][col] of the reference's frame array sits in the packed wire frame: bit index from the frame's first bit (bit 7 of
// byte 0 = index 0), -1 for a cell the codec does not use.  The row widths are the packers' (pack_rows: row after row, inside a row
// from the highest cell down to cell 0).
static const int* wire_row_widths(int codec, int* rows) {
    static const int imbe7200[8] = {23, 23, 23, 23, 15, 15, 15, 7}, imbe7100[7] = {19, 24, 23, 23, 15, 15, 23}, ambe[4] = {24, 23, 11, 14};
    switch (codec) {
        case MBX_CODEC_IMBE7200X4400: *rows = 8; return imbe7200;
        case MBX_CODEC_IMBE7100X4400: *rows = 7; return imbe7100;
        case MBX_CODEC_AMBE3600X2450:
        case MBX_CODEC_AMBE3600X2400: *rows = 4; return ambe;
        default: *rows = 0; return nullptr;
    }
}
int mbx_wire_bit_of_cell(int codec, int row, int col) {
    int rows = 0;
    const int* w = wire_row_widths(codec, &rows);
    if (!w || row < 0 || row >= rows || col < 0 || col >= w[row]) {
        return -1;
    }
    int pos = 0;
    for (int r = 0; r < row; ++r) {
        pos += w[r];
    }
    return pos + (w[row] - 1 - col);
}

// Folds a caller's deinterleave schedule into one table: the caller knows, for each of the n channel bits of a voice burst in the order
// it receives them, the cell (cell_row[i], cell_col[i]) its air-interface tables send that bit to; wire_bit[i] is where the same bit
// goes in the packed wire frame.  Refuses a schedule that is not a bijection onto the codec's cells (a wrong table is caught here,
// once, not as noise in the audio).
int mbx_wire_permutation(int codec, const int* cell_row, const int* cell_col, int n, int* wire_bit) {
    int rows = 0;
    const int* w = wire_row_widths(codec, &rows);
    if (!w || !cell_row || !cell_col || !wire_bit) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int total = 0;
    for (int r = 0; r < rows; ++r) {
        total += w[r];
    }
    if (n != total) {
        return fail(MBE_STATUS_INVALID_ARGUMENT, "mbx_wire_permutation: the schedule must name every channel bit of the codec once");
    }
    bool seen[192] = {false};
    for (int i = 0; i < n; ++i) {
        const int b = mbx_wire_bit_of_cell(codec, cell_row[i], cell_col[i]);
        if (b < 0 || seen[b]) {
            return fail(MBE_STATUS_INVALID_ARGUMENT, "mbx_wire_permutation: a cell outside the codec's frame, or named twice");
        }
        seen[b] = true;
        wire_bit[i] = b;
    }
    return 0;
}

int mbx_pack_imbe7200x4400(const char* frames, size_t n, uint8_t* packed) {
    static const int width[8] = {23, 23, 23, 23, 15, 15, 15, 7};
    int rc = validate_bits(frames, n * 184u);
    if (rc < 0) {
        return rc;
    }
    if (!packed) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < n; ++i) {
        pack_rows(frames + i * 184u, 8, 23, width, packed + i * MBX_IMBE_FRAME_BYTES, MBX_IMBE_FRAME_BYTES);
    }
    return 0;
}

int mbx_pack_imbe7100x4400(const char* frames, size_t n, uint8_t* packed) {
    static const int width[7] = {19, 24, 23, 23, 15, 15, 23};
    int rc = validate_bits(frames, n * 168u);
    if (rc < 0) {
        return rc;
    }
    if (!packed) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < n; ++i) {
        pack_rows(frames + i * 168u, 7, 24, width, packed + i * MBX_IMBE7100_FRAME_BYTES, MBX_IMBE7100_FRAME_BYTES);
    }
    return 0;
}

int mbx_pack_ambe3600x2450(const char* frames, size_t n, uint8_t* packed) {
    static const int width[4] = {24, 23, 11, 14};
    int rc = validate_bits(frames, n * 96u);
    if (rc < 0) {
        return rc;
    }
    if (!packed) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < n; ++i) {
        pack_rows(frames + i * 96u, 4, 24, width, packed + i * MBX_AMBE_FRAME_BYTES, MBX_AMBE_FRAME_BYTES);
    }
    return 0;
}

void mbx_unpack_records(const mbx_param_record* rec, size_t n, int nbits, char* bits, mbe_process_result* results) {
    for (size_t i = 0; i < n; ++i) {
        if (bits) {
            for (int b = 0; b < nbits; ++b) {
                bits[i * (size_t)nbits + b] = (char)((rec[i].w[b >> 5] >> (31 - (b & 31))) & 1u);
            }
        }
        if (results) {
            mbe_process_result& r = results[i];
            r.c0_errors = (int)(rec[i].w[3] & 0xffu);
            r.protected_errors = (int)((rec[i].w[3] >> 8) & 0xffu);
            r.c4_errors = (int)((rec[i].w[3] >> 16) & 0xffu);
            r.total_errors = r.c0_errors + r.protected_errors;
            r.flags = (rec[i].w[3] >> 24) & 0xffu;
        }
    }
}

void mbx_rng_default(mbx_stream_rng* rng) {
    memset(rng, 0, sizeof(*rng));
    rng->unvoiced_seed_state = 3147u;
}

void mbx_rng_seed(mbx_stream_rng* rng, uint32_t seed) {   // ref: src/core/mbelib.c:173-181
    if (seed == 0u) {
        seed = 0x6d25357bu;
    }
    rng->cn_seed48 = (((uint64_t)seed) ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1ULL);
    rng->cn_seeded = 1;
    rng->unvoiced_seed_state = seed % 53125u;
    rng->unvoiced_seed_override = 1;
}

// ---- workspace of the stream stage ------------------------------------------------------------------------------
// caller holds c->mu
static int ensure_workspace(Context* c, StreamSlot& slot, size_t frames, void* stream) {
    if (frames <= slot.frames) {
        return 0;
    }
    const size_t want = frames > c->reserve_frames ? frames : c->reserve_frames;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        return fail(MBE_STATUS_INVALID_ARGUMENT,
                    "the expand workspace of this stream would have to grow during stream capture: call mbx_reserve_stream() first");
    }
    (void)hipGetLastError();
    if (slot.workspace) {   // earlier launches on this stream may still read it
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipFree(slot.workspace));
        slot.workspace = nullptr;
        slot.flags = nullptr;
        slot.frames = 0;
        slot.exp_codec = -1;
    }
    const size_t flag_bytes = (((want + 7) / 8 + 1) * sizeof(uint32_t) + 255) & ~(size_t)255;   // (+ 1: the fall-back counter)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&slot.workspace), want * sizeof(mbx::FrameParams) + flag_bytes));
    slot.frames = want;
    slot.flags = reinterpret_cast<uint32_t*>(slot.workspace + want);
    slot.epoch = 0;
    // (no launch has epoch 0.)  On the LAUNCH stream: a non-blocking stream is not ordered against the null stream, so a null-stream
    // memset could land after the first launch's front blocks have published their flags.  (Not capturing: checked above.)
    HIP_TRY(hipMemsetAsync(slot.flags, 0, flag_bytes, (hipStream_t)stream));
    return 0;
}

int mbx_reserve_stream(void* stream, size_t max_frames) {
    REQUIRE_CTX(c);
    std::lock_guard<std::mutex> lock(c->mu);
    return ensure_workspace(c, c->slots[stream], max_frames, stream);
}

int mbx_reserve(size_t max_frames) {
    REQUIRE_CTX(c);
    std::lock_guard<std::mutex> lock(c->mu);
    if (max_frames > c->reserve_frames) {
        c->reserve_frames = max_frames;   // every slot grows to at least this much the next time it has to grow at all
    }
    // Only the default stream's slot is sized here.  The slots of other streams are NOT walked: the library cannot know
    // whether a hipStream_t it saw earlier still exists, and synchronising a destroyed handle is undefined -- a stream
    // that must not allocate at its next launch (stream capture) is sized with mbx_reserve_stream() by its owner.
    return ensure_workspace(c, c->slots[nullptr], max_frames, nullptr);
}

int mbx_release_stream(void* stream) {
    REQUIRE_CTX(c);
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->slots.find(stream);
    if (it == c->slots.end()) {
        return 0;
    }
    if (it->second.workspace) {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipFree(it->second.workspace));
    }
    release_side_streams(it->second);
    c->slots.erase(it);
    return 0;
}

size_t mbx_workspace_bytes(size_t max_frames) { return max_frames * sizeof(mbx::FrameParams); }

int mbx_fec_imbe7200x4400(const uint8_t* d_frames, size_t n, mbx_param_record* d_records, void* stream) {
    REQUIRE_CTX(c);
    if (!d_frames || !d_records) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(mbx::fec_imbe7200x4400_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_frames, n,
                       d_records, c->tabs);
    return check_launch("fec_imbe7200x4400_kernel");
}

int mbx_fec_ambe3600x2450(const uint8_t* d_frames, size_t n, mbx_param_record* d_records, void* stream) {
    REQUIRE_CTX(c);
    if (!d_frames || !d_records) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(mbx::fec_ambe3600x2450_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_frames, n,
                       d_records, c->tabs);
    return check_launch("fec_ambe3600x2450_kernel");
}

int mbx_fec_imbe7100x4400(const uint8_t* d_frames, size_t n, mbx_param_record* d_records, void* stream) {
    REQUIRE_CTX(c);
    if (!d_frames || !d_records) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(mbx::fec_imbe7100x4400_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_frames, n,
                       d_records, c->tabs);
    return check_launch("fec_imbe7100x4400_kernel");
}

int mbx_fec_soft(int codec, const mbe_soft_bit* d_soft, size_t n, mbx_param_record* d_records, void* stream) {
    REQUIRE_CTX(c);
    if (!d_soft || !d_records || codec < MBX_CODEC_IMBE7200X4400 || codec > MBX_CODEC_AMBE3600X2400) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    if (n > 0x7fffffffu) {
        return fail(MBE_STATUS_INVALID_ARGUMENT, "mbx_fec_soft: more than 2^31-1 frames in one launch");
    }
    if (codec == MBX_CODEC_IMBE7200X4400) {   // one wavefront per frame
        hipLaunchKernelGGL(mbx::fec_imbe7200x4400_soft_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_soft, n,
                           d_records, c->tabs);
    } else if (codec == MBX_CODEC_IMBE7100X4400) {
        hipLaunchKernelGGL(mbx::fec_imbe7100x4400_soft_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_soft, n,
                           d_records, c->tabs);
    } else {
        hipLaunchKernelGGL(mbx::fec_ambe3600x2450_soft_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_soft, n,
                           d_records, c->tabs);
    }
    return check_launch("fec_soft_kernel");
}

int mbx_ecc_soft_words(int kind, const mbe_soft_bit* d_in, size_t n, uint32_t* d_out, int32_t* d_errs, void* stream) {
    REQUIRE_CTX(c);
    if (!d_in || !d_out || kind < 0 || kind > 2 || n > 0x7fffffffu) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::ecc_soft_words_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, kind, d_in, n, d_out,
                       d_errs, c->tabs);
    return check_launch("ecc_soft_words_kernel");
}

int mbx_validate_soft_bits(const mbe_soft_bit* soft, size_t count) {
    if (!soft) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < count; ++i) {
        if (soft[i].bit > 1u) {
            return MBE_STATUS_INVALID_BITS;
        }
    }
    return 0;
}

int mbx_soft_bits_from_hard(const char* bits, mbe_soft_bit* soft, size_t count, uint8_t reliability) {
    if (!soft || !bits) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < count; ++i) {
        if ((unsigned char)bits[i] > 1u) {
            return MBE_STATUS_INVALID_BITS;
        }
    }
    for (size_t i = 0; i < count; ++i) {
        soft[i].bit = (uint8_t)(bits[i] ? 1u : 0u);
        soft[i].reliability = reliability;
    }
    return 0;
}

int mbx_soft_bits_from_llr(const int16_t* llr, mbe_soft_bit* soft, size_t count) {
    if (!llr || !soft) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < count; ++i) {
        const int v = llr[i], mag = v < 0 ? -v : v;
        soft[i].bit = (uint8_t)(v > 0 ? 1u : 0u);
        soft[i].reliability = (uint8_t)(mag > 255 ? 255 : mag);
    }
    return 0;
}

static bool expand_codec_ok(int codec) {
    return codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400;
}

// frame-parallel half of the parameter decode: records -> FrameParams rows at `out`
static int launch_expand(Context* c, int codec, const mbx_param_record* d_records, size_t n, mbx::FrameParams* out, void* stream) {
    const unsigned egrid = (unsigned)((n + 31) / 32);   // 8 frames per wave, 4 waves per workgroup (mbx_expand.hip)
    if (codec == MBX_CODEC_IMBE7200X4400) {
        hipLaunchKernelGGL(mbx::expand_imbe_kernel, dim3(egrid), dim3(256), 0, (hipStream_t)stream, d_records, n, out, c->tabs);
    } else if (codec == MBX_CODEC_AMBE3600X2400) {
        hipLaunchKernelGGL(mbx::expand_ambe2400_kernel, dim3(egrid), dim3(256), 0, (hipStream_t)stream, d_records, n, out, c->tabs);
    } else {
        hipLaunchKernelGGL(mbx::expand_ambe_kernel, dim3(egrid), dim3(256), 0, (hipStream_t)stream, d_records, n, out, c->tabs);
    }
    return check_launch("expand_kernel");
}

// ---- sliced launches (mbx_stream.hip, *_stream_kernel_lds_slice) ------------------------------------------------------------------
// A launch of S streams x T frames is S workgroups of equal length; when S does not fill the device's resident wave slots evenly
// the last round runs part-empty -- BASELINE configs[4]'s shard, 8,192 AMBE+2 streams on 5,120 slots, is 1.6 rounds: the second
// holds 3,072 waves.  Frames of a stream are sequential, but nothing says they must be ONE launch's: the streams are split into
// THREE groups and the frames into slices of 16, and the slices of each group are issued in order on an internal HIP stream of its
// own (forked from and joined to the caller's stream with events).  Measured, 8,192 x 128 AMBE+2, interleaved A/B on one box: plain
// 2.609 ms; two groups 2.518; three 2.481 (-4.7 %); FOUR 3.185 (+22 %: with the caller's stream that is five streams on HIP's four
// hardware queues, and two groups then share one); slices of 8 / 16 / 32 frames with two groups: 2.512 / 2.524 / 2.585.  A slice IS a launch of 16 frames per stream (state in from HBM,
// state out: results bit-identical by construction); the streams' kernels share the device, so the slots one group's slice
// leaves empty are taken by another group's next one (ideal: work / slots = 2.29 ms for that shape instead of ceil(S / slots) rounds).
// Kernel boundaries do the ordering: no in-kernel waiting, no assumption about dispatch.  (Built first as ONE grid of K x S
// workgroups that waited for their stream's previous slice on a progress word: the agent-scope release / acquire pair every slice
// then needs -- an L2 write-back with the PCM of 5,120 waves dirty in it -- made 8,192 x 128 in eight slices 34 % SLOWER.)
// MBX_SLICE=0 switches it off, MBX_SLICE=n sets the slice length in frames (A/B timing; read once).
static int slice_override() {
    static const int v = [] {
        const char* e = getenv("MBX_SLICE");
        return e ? atoi(e) : -1;
    }();
    return v;
}
// frames per slice for a launch of S streams x T frames on `slots` resident waves, or 0: the plain launch
static int choose_slice_frames(int S, int T, int slots) {
    const int forced = slice_override();
    if (forced == 0 || slots <= 0 || S < 2) {
        return 0;
    }
    const int Tc = forced > 0 ? ((forced + 7) & ~7) : 16;   // multiples of eight: the AMBE bodies expand eight frames at a time
    if (T < 2 * Tc) {
        return 0;
    }
    if (forced > 0) {
        return Tc;
    }
    if (S <= slots) {   // every stream has a slot of its own: nothing to balance
        return 0;
    }
    const double ideal = (double)S / (double)slots;
    const double plain = (double)((S + slots - 1) / slots);
    return (plain >= 1.06 * ideal) ? Tc : 0;   // what the part-empty last round costs must be worth the extra launches
}
// resident waves per SIMD of the kernel a long launch of `codec` takes (the kernels' own launch bounds: mbx_device.h)
static int lds_kernel_waves_per_simd(int codec) {
    return (codec == MBX_CODEC_IMBE7200X4400) ? MBX_IMBE_LDS_WAVES_PER_SIMD : MBX_AMBE_LDS_WAVES_PER_SIMD;
}
// caller holds c->mu.  0: issued (*rc); 1: not applicable (take the plain launch)
static int try_sliced_launch(Context* c, StreamSlot& slot, mbx::DeviceTables tabs, int codec, int S, int T, const mbx_param_record* d_records,
                             const mbx::FrameParams* params, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf,
                             mbe_process_result* d_results, void* stream, int* rc) {
    const int Tc = choose_slice_frames(S, T, lds_kernel_waves_per_simd(codec) * c->simds);
    if (Tc <= 0) {
        return 1;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = stream && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    if (capturing) {
        return 1;
    }
    static const int groups = [] {
        const char* e = getenv("MBX_SLICE_GROUPS");
        const int v = e ? atoi(e) : 3;
        return v < 2 ? 2 : (v > 4 ? 4 : v);
    }();
    if (!slot.fork) {   // first sliced launch on this stream: the internal streams of the groups (at most four, MBX_SLICE_GROUPS) and their events
        bool ok = hipEventCreateWithFlags(&slot.fork, hipEventDisableTiming) == hipSuccess;
        for (int g = 0; g < groups && ok; ++g) {
            ok = hipStreamCreateWithFlags(&slot.side[g], hipStreamNonBlocking) == hipSuccess
                 && hipEventCreateWithFlags(&slot.join[g], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            release_side_streams(slot);
            return 1;
        }
    }
    if (hipEventRecord(slot.fork, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    int result = 0;
    // Group 0 runs on the caller's stream itself -- its kernels are queued before the stream waits for the other groups --, so g
    // groups occupy g hardware queues, not g + 1, and one group needs no hand-over at all: 2.51 -> 2.42 ms on 8,192 x 128 AMBE+2
    // (interleaved A/B; four groups are +26 % either way).  MBX_SLICE_OWN=0 puts every group on an internal stream (A/B timing; read once).
    static const bool own = [] {
        const char* e = getenv("MBX_SLICE_OWN");
        return !(e && e[0] == '0');
    }();
    // Every fork-side wait is queued BEFORE any group's kernels: should one fail, nothing has been launched yet and the state is
    // untouched (the caller then gets the plain launch).  A wait queued on an internal stream and never followed by work is harmless.
    for (int g = (own ? 1 : 0); g < groups; ++g) {
        if (hipStreamWaitEvent(slot.side[g], slot.fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
    }
    for (int g = 0; g < groups; ++g) {
        const int s0 = (int)((long long)S * g / groups), Sg = (int)((long long)S * (g + 1) / groups) - s0;
        const bool on_caller = own && g == 0;
        hipStream_t st = on_caller ? (hipStream_t)stream : slot.side[g];
        mbx::DeviceTables tg = tabs;
        if (tg.stream_map) {
            tg.stream_map += s0;
        }
        const size_t f0 = (size_t)s0 * (size_t)T;
        const mbx_param_record* rg = d_records + f0;
        const mbx::FrameParams* pg = params ? params + f0 : nullptr;
        // (state and RNG state are addressed through the stream map when there is one: then only the batch rows are offset)
        mbe_parms* sg = tabs.stream_map ? d_state : d_state + 3 * (size_t)s0;
        mbx_stream_rng* ng = tabs.stream_map ? d_rng : d_rng + s0;
        int16_t* p16 = d_pcm16 ? d_pcm16 + f0 * 160 : nullptr;
        float* pf = d_pcmf ? d_pcmf + f0 * 160 : nullptr;
        mbe_process_result* rs = d_results ? d_results + f0 : nullptr;
        for (int t0 = 0; t0 < T; t0 += Tc) {
            const int n = (T - t0) < Tc ? (T - t0) : Tc;
            if (codec == MBX_CODEC_IMBE7200X4400) {
                hipLaunchKernelGGL(mbx::imbe_stream_kernel_lds_slice, dim3((unsigned)Sg), dim3(64), 0, st, Sg, T, t0, n, rg, pg, sg, ng, p16, pf, rs, tg);
            } else if (codec == MBX_CODEC_AMBE3600X2400) {
                hipLaunchKernelGGL(mbx::ambe2400_stream_kernel_lds_slice, dim3((unsigned)Sg), dim3(64), 0, st, Sg, T, t0, n, rg, pg, sg, ng, p16, pf, rs, tg);
            } else {
                hipLaunchKernelGGL(mbx::ambe_stream_kernel_lds_slice, dim3((unsigned)Sg), dim3(64), 0, st, Sg, T, t0, n, rg, pg, sg, ng, p16, pf, rs, tg);
            }
        }
        const int lrc = check_launch("stream_kernel_lds_slice");
        if (lrc < 0) {
            result = lrc;
        }
        if (!on_caller && (hipEventRecord(slot.join[g], st) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, slot.join[g], 0) != hipSuccess)) {
            (void)hipGetLastError();
            result = fail(MBX_ENODEVICE, "sliced launch: join");
        }
    }
    *rc = result;
    return 0;
}

// Stream-stage launch.  `params` = FrameParams rows written by the expand stage, or nullptr: the IMBE stream kernel
// then expands each record itself (one launch less, no workspace traffic).
//
// `reverse`: successive launches over the same streams walk them in opposite directions.  A decoder is called for the
// same streams every 20 ms; the state of 65,536 of them (512 MB) does not fit the 256 MB Infinity Cache, so in a fixed
// order every launch finds none of it there, while a launch that starts where the previous one ended finds its last
// quarter-gigabyte.  The results do not depend on the order.  The alternation is kept per (device, hipStream_t) slot
// and per session -- whoever re-walks the same state -- not process-wide.
static int launch_stream(Context* c, bool reverse, int codec, int S, int T, const mbx_param_record* d_records,
                         const mbx::FrameParams* params, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16,
                         float* d_pcmf, mbe_process_result* d_results, void* stream, const int32_t* d_stream_index = nullptr,
                         uint32_t* d_resident = nullptr, StreamSlot* slot = nullptr /* caller holds c->mu: time-sliced launches allowed */) {
    mbx::DeviceTables tabs = c->tabs;
    tabs.reverse = (reverse && reverse_enabled()) ? 1 : 0;
    tabs.stream_map = d_stream_index;
    tabs.resident = d_resident;
    // With several frames per stream prev_mp / prev_mp_enhanced stay in LDS for the whole launch (the *_lds instances,
    // four waves per SIMD) instead of being parked in their HBM slots every frame: mbx_stream.hip, ParkedState.
    // Resident state (d_resident) is understood by those instances only, whatever T is.
    const bool lds_resident = T >= kLdsResidentMinFrames && lds_resident_enabled();
    if (slot && lds_resident && !d_resident) {
        int rc = 0;
        if (try_sliced_launch(c, *slot, tabs, codec, S, T, d_records, params, d_state, d_rng, d_pcm16, d_pcmf, d_results, stream, &rc) == 0) {
            return rc;
        }
    }
    if (d_resident) {
        if (codec == MBX_CODEC_IMBE7200X4400 && T == 1 && res1_enabled()) {
            hipLaunchKernelGGL(mbx::imbe_stream_kernel_res1, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        } else if (codec == MBX_CODEC_IMBE7200X4400) {
            hipLaunchKernelGGL(mbx::imbe_stream_kernel_res, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        } else if (T == 1 && params && res1_enabled()) {   // the AMBE codecs: one frame per stream on rows from the expand launch
            auto* const k1 = codec == MBX_CODEC_AMBE3600X2400 ? mbx::ambe2400_stream_kernel_res1 : mbx::ambe_stream_kernel_res1;
            hipLaunchKernelGGL(k1, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records, params, d_state, d_rng, d_pcm16, d_pcmf,
                               d_results, tabs);
        } else if (codec == MBX_CODEC_AMBE3600X2400) {
            hipLaunchKernelGGL(mbx::ambe2400_stream_kernel_res, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        } else {
            hipLaunchKernelGGL(mbx::ambe_stream_kernel_res, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        }
        return check_launch("stream_kernel_res");
    }
    if (codec == MBX_CODEC_IMBE7200X4400) {
        if (lds_resident) {
#ifdef MBX_EXP_PAIR   // experiment builds only (mbx_stream.hip, imbe_stream_kernel_lds_pairexp)
            if ((S & 1) == 0 && !tabs.stream_map) {
                hipLaunchKernelGGL(mbx::imbe_stream_kernel_lds_pairexp, dim3((unsigned)S / 2), dim3(128), 0, (hipStream_t)stream, S, T, d_records,
                                   params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
                return check_launch("imbe_stream_kernel_lds_pairexp");
            }
#endif
            hipLaunchKernelGGL(mbx::imbe_stream_kernel_lds, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
            return check_launch("imbe_stream_kernel_lds");
        }
        if (T == 1) {
            hipLaunchKernelGGL(mbx::imbe_stream_kernel_one, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
            return check_launch("imbe_stream_kernel_one");
        }
        hipLaunchKernelGGL(mbx::imbe_stream_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                           params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        return check_launch("imbe_stream_kernel");
    }
    if (codec == MBX_CODEC_AMBE3600X2400) {
        if (lds_resident) {
            hipLaunchKernelGGL(mbx::ambe2400_stream_kernel_lds, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
            return check_launch("ambe2400_stream_kernel_lds");
        }
        if (T == 1) {
            hipLaunchKernelGGL(mbx::ambe2400_stream_kernel_one, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                               params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
            return check_launch("ambe2400_stream_kernel_one");
        }
        hipLaunchKernelGGL(mbx::ambe2400_stream_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                           params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        return check_launch("ambe2400_stream_kernel");
    }
    // (Long AMBE+2 launches used to need a second, register-padded instance to even out their rounds of waves -- config 5's
    // shard is 8 waves per SIMD on 6 slots: 6 + 2.  The LDS-resident instance runs 16 waves per CU: 2 x 16, and
    // 8,192 streams x T = 128 went from 3.51 ms to 3.15 ms.)
    if (lds_resident) {
        hipLaunchKernelGGL(mbx::ambe_stream_kernel_lds, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                           params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        return check_launch("ambe_stream_kernel_lds");
    }
    if (T == 1) {
        hipLaunchKernelGGL(mbx::ambe_stream_kernel_one, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                           params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        return check_launch("ambe_stream_kernel_one");
    }
    hipLaunchKernelGGL(mbx::ambe_stream_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, T, d_records,
                       params, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
    return check_launch("ambe_stream_kernel");
}

static bool stream_args_ok(int codec, int S, int T, const void* d_records, const void* d_state, const void* d_rng) {
    return d_records && d_state && d_rng && S >= 0 && T >= 0 && expand_codec_ok(codec);
}

// IMBE with several frames per stream: the stream kernel expands the records itself, which saves the workspace round
// trip and a launch (+4 % at T = 16).  With ONE frame per stream the whole-job rate is the same either way (measured
// 0.296 vs 0.292 ms per 65,536 frames): the table look-ups of the expansion are a latency chain a one-frame wave cannot
// hide, and the 8-lanes-per-frame expand kernel costs as much as it saves -- there the expansion stays a separate
// launch, which keeps the dominant kernel to the stream stage proper.  The AMBE HBM-slot instances always read rows.
// ... except for small batches (the synchronous per-frame API is S = T = 1): there a launch less is worth more than the
// last few per cent of kernel efficiency, and the IMBE stream kernel expands the record itself.
constexpr int kSmallBatchFrames = 256;
static bool needs_workspace(int codec, int S, int T) {
    if (codec == MBX_CODEC_IMBE7200X4400) {
        return T == 1 && S > kSmallBatchFrames;
    }
    // AMBE codecs: the LDS-resident instances (T >= 4) expand eight frames of their stream at a time into LDS rows;
    // the HBM-slot instances read rows from the workspace
    return !(T >= kLdsResidentMinFrames && lds_resident_enabled());
}

// The whole T = 1 step of the IMBE codecs as ONE launch (mbx_stream.hip).  Taken by the mbx_process_batch* entry points (which have
// the frames); the records-based entry points keep the expand + stream pair.  Two forms:
//   2 (default, 7200x4400): imbe_one_launch_kernel -- front blocks (FEC + expansion of eight frames per wave) and stream blocks in one
//     grid, rows handed over through the stream's workspace (mbx_front_imbe.h);
//   1 (7100x4400; 7200x4400 with MBX_FUSE_ONE=1): imbe_stream_kernel_one_fused -- the front end in the stream's own wave.
// MBX_FUSE_ONE=0 switches both off (A/B timing; read once).  MBX_FRONT_LEAD: by how many chunks of eight streams a front block
// runs ahead of its stream blocks in the grid.  Default: all front blocks first.  Measured (65,536 x 1, interleaved A/B, one box):
// lead 0 / 128 / 512: 0.37 / 0.36 / 0.35 ms (stream blocks start before their rows exist and wait); 1024 / 2048 / 4096: 0.2245 /
// 0.2245 / 0.2230; all first: 0.2229 -- and with the front blocks at a raised wave priority 0.2259 / 0.2253 against 0.2194: what a
// front block costs is the wave SLOT it holds for the ~10 us of its table-read chain, not its instructions, and slots are what an
// interleaved front block takes away from stream blocks that could use them.
static int fused_one_mode() {
    static const int mode = [] {
        const char* e = getenv("MBX_FUSE_ONE");
        return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
    }();
    return mode;
}
static int front_lead_chunks() {
    static const int lead = [] {
        const char* e = getenv("MBX_FRONT_LEAD");
        const int v = e ? atoi(e) : 0x7fffffff;
        return v < 0 ? 0 : v;
    }();
    return lead;
}
static bool ambe_codec(int codec) { return codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400; }
static bool fused_one_ok(int codec, int S, int T, const void* d_frames, bool resident = false) {
    (void)resident;
    if (T != 1 || S <= kSmallBatchFrames || fused_one_mode() == 0) {
        return false;
    }
    if (ambe_codec(codec)) {
        return fused_one_mode() == 2;   // 9-byte frames: byte loads, any alignment
    }
    return (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400) && (reinterpret_cast<uintptr_t>(d_frames) & 3u) == 0;
}
static bool one_launch_form(int codec) { return (codec == MBX_CODEC_IMBE7200X4400 || ambe_codec(codec)) && fused_one_mode() == 2; }
static int launch_fused_one(Context* c, bool reverse, int codec, int S, const uint8_t* d_frames, mbx_param_record* d_records,
                            mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                            void* stream, const int32_t* d_stream_index, uint32_t* d_resident) {
    mbx::DeviceTables tabs = c->tabs;
    tabs.reverse = (reverse && reverse_enabled()) ? 1 : 0;
    tabs.stream_map = d_stream_index;
    tabs.resident = d_resident;
    const bool v7100 = codec == MBX_CODEC_IMBE7100X4400;
    auto* const kernel = d_resident ? (v7100 ? mbx::imbe7100_stream_kernel_res1_fused : mbx::imbe_stream_kernel_res1_fused)
                                    : (v7100 ? mbx::imbe7100_stream_kernel_one_fused : mbx::imbe_stream_kernel_one_fused);
    hipLaunchKernelGGL(kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, codec, d_frames, d_records, d_state, d_rng, d_pcm16,
                       d_pcmf, d_results, tabs);
    return check_launch("imbe_stream_kernel_one_fused");
}
// caller holds c->mu; the slot's workspace holds S rows and its flags
static int launch_one_launch(Context* c, StreamSlot& slot, bool reverse, int codec, int S, const uint8_t* d_frames, mbx_param_record* d_records,
                             mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                             void* stream, const int32_t* d_stream_index, uint32_t* d_resident) {
    mbx::DeviceTables tabs = c->tabs;
    tabs.reverse = (reverse && reverse_enabled()) ? 1 : 0;
    tabs.stream_map = d_stream_index;
    tabs.resident = d_resident;
    if (++slot.epoch == 0u) {
        slot.epoch = 1u;
    }
    slot.exp_codec = -1;   // the rows of an earlier mbx_expand_records() are being replaced
    const unsigned chunks = (unsigned)((S + 7) / 8);
    if (ambe_codec(codec)) {
        auto* const akernel = d_resident ? (codec == MBX_CODEC_AMBE3600X2400 ? mbx::ambe2400_one_launch_kernel_res : mbx::ambe_one_launch_kernel_res)
                                         : (codec == MBX_CODEC_AMBE3600X2400 ? mbx::ambe2400_one_launch_kernel : mbx::ambe_one_launch_kernel);
        hipLaunchKernelGGL(akernel, dim3(9u * chunks), dim3(64), 0, (hipStream_t)stream, S, d_frames, d_records, slot.workspace, slot.flags,
                           slot.flags + (slot.frames + 7) / 8, slot.epoch, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
        return check_launch("ambe_one_launch_kernel");
    }
    auto* const kernel = d_resident ? mbx::imbe_one_launch_kernel_res : mbx::imbe_one_launch_kernel;
    hipLaunchKernelGGL(kernel, dim3(9u * chunks), dim3(64), 0, (hipStream_t)stream, S, front_lead_chunks(), d_frames, d_records, slot.workspace,
                       slot.flags, slot.flags + (slot.frames + 7) / 8, slot.epoch, d_state, d_rng, d_pcm16, d_pcmf, d_results, tabs);
    return check_launch("imbe_one_launch_kernel");
}
// one fused launch if the shape allows it: returns 1 when it was issued (*rc = its status), 0 when the caller goes on with the stages
static int try_fused_one(int codec, int S, int T, const uint8_t* d_frames, mbx_param_record* d_records, mbe_parms* d_state,
                         mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, void* stream,
                         const int32_t* d_stream_index, uint32_t* d_resident, int* rc, bool own_workspace = true) {
    if (!fused_one_ok(codec, S, T, d_frames, d_resident != nullptr)) {
        return 0;
    }
    int crc;
    Context* c = current_ctx(&crc);
    if (!c) {
        *rc = crc;
        return 1;
    }
    if (!d_state || !d_rng) {
        *rc = MBE_STATUS_INVALID_ARGUMENT;
        return 1;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    StreamSlot& slot = c->slots[stream];
    if (one_launch_form(codec) && own_workspace) {
        // A launch that is being CAPTURED into a graph would be replayed with the same epoch, and a replay would find the flags of
        // the replay before it: captured launches take the staged kernels.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool capturing = stream && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
        (void)hipGetLastError();
        if (capturing) {
            return 0;
        }
        const int wrc = ensure_workspace(c, slot, (size_t)S, stream);
        if (wrc < 0) {
            *rc = wrc;
            return 1;
        }
        *rc = launch_one_launch(c, slot, (slot.launches++ & 1u) != 0u, codec, S, d_frames, d_records, d_state, d_rng, d_pcm16, d_pcmf, d_results,
                                stream, d_stream_index, d_resident);
        return 1;
    }
    if (ambe_codec(codec)) {   // (no in-wave form for the AMBE codecs: a caller-owned workspace or a captured launch takes the staged kernels)
        return 0;
    }
    *rc = launch_fused_one(c, (slot.launches++ & 1u) != 0u, codec, S, d_frames, d_records, d_state, d_rng, d_pcm16, d_pcmf, d_results, stream,
                           d_stream_index, d_resident);
    return 1;
}

extern "C" int mbx_uses_expand_launch(int codec, int S, int T) { return needs_workspace(codec == MBX_CODEC_IMBE7100X4400 ? MBX_CODEC_IMBE7200X4400 : codec, S, T) ? 1 : 0; }

// expand (where needed) + stream kernel with the workspace at `ws` (nullptr when none is needed); `order` = the launch
// counter that decides the walking direction
static int run_stream_stage(Context* c, unsigned order, int codec, int S, int T, const mbx_param_record* d_records,
                            mbx::FrameParams* ws, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf,
                            mbe_process_result* d_results, void* stream, const int32_t* d_stream_index = nullptr,
                            uint32_t* d_resident = nullptr, StreamSlot* slot = nullptr) {
    if (needs_workspace(codec, S, T)) {
        int rc = launch_expand(c, codec, d_records, (size_t)S * (size_t)T, ws, stream);
        if (rc < 0) {
            return rc;
        }
    } else {
        ws = nullptr;
    }
    return launch_stream(c, (order & 1u) != 0u, codec, S, T, d_records, ws, d_state, d_rng, d_pcm16, d_pcmf, d_results, stream,
                         d_stream_index, d_resident, slot);
}

int mbx_expand_records(int codec, const mbx_param_record* d_records, size_t n, void* stream) {
    REQUIRE_CTX(c);
    if (!d_records || !expand_codec_ok(codec)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    StreamSlot& slot = c->slots[stream];
    int rc = ensure_workspace(c, slot, n, stream);
    if (rc < 0) {
        return rc;
    }
    rc = launch_expand(c, codec, d_records, n, slot.workspace, stream);
    slot.exp_codec = rc < 0 ? -1 : codec;
    slot.exp_n = n;
    slot.exp_records = d_records;
    return rc;
}

static int stream_expanded(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state, uint32_t* d_resident,
                           mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, void* stream);

// The two halves with a CALLER-OWNED workspace, so that they can run on DIFFERENT streams: the frame-parallel front end
// (FEC + expansion) of batch k + 1 does not depend on the stream stage of batch k, only on the frames -- a host that decodes
// batch after batch (recorded traffic, many sites) MAY let them overlap and order them with events of its own.  Measured on
// MI355X at T = 1 (bench.py --overlap-front-end): slower than everything on one stream (255 against 265 M frames/s) -- the
// HBM-bound stream kernel slows down by what it shares, and two cross-stream hand-overs per step cost what the overlap saves.
int mbx_expand_records_ws(int codec, const mbx_param_record* d_records, size_t n, void* d_workspace, size_t workspace_bytes,
                          void* stream) {
    REQUIRE_CTX(c);
    if (!d_records || !expand_codec_ok(codec) || !d_workspace || workspace_bytes < mbx_workspace_bytes(n)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    return launch_expand(c, codec, d_records, n, static_cast<mbx::FrameParams*>(d_workspace), stream);
}

int mbx_stream_expanded_ws(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state, uint32_t* d_resident,
                           mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                           const void* d_workspace, size_t workspace_bytes, void* stream) {
    REQUIRE_CTX(c);
    if (!stream_args_ok(codec, S, T, d_records, d_state, d_rng) || !d_workspace
        || workspace_bytes < mbx_workspace_bytes((size_t)S * (size_t)T)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0 || T == 0) {
        return 0;
    }
    unsigned order;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        order = c->slots[stream].launches++;
    }
    return launch_stream(c, (order & 1u) != 0u, codec, S, T, d_records, static_cast<const mbx::FrameParams*>(d_workspace), d_state,
                         d_rng, d_pcm16, d_pcmf, d_results, stream, nullptr, d_resident);
}

int mbx_stream_expanded(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state,
                        mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                        void* stream) {
    return stream_expanded(codec, S, T, d_records, d_state, nullptr, d_rng, d_pcm16, d_pcmf, d_results, stream);
}

// the same on resident state (mbx_process_batch_resident): bench.py brackets the stream kernel alone with it
int mbx_stream_expanded_resident(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state, uint32_t* d_resident,
                                 mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                                 void* stream) {
    if (!d_resident) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    return stream_expanded(codec, S, T, d_records, d_state, d_resident, d_rng, d_pcm16, d_pcmf, d_results, stream);
}

static int stream_expanded(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state, uint32_t* d_resident,
                           mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, void* stream) {
    REQUIRE_CTX(c);
    if (!stream_args_ok(codec, S, T, d_records, d_state, d_rng)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0 || T == 0) {
        return 0;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    StreamSlot& slot = c->slots[stream];
    if (slot.exp_codec != codec || slot.exp_n != (size_t)S * (size_t)T || slot.exp_records != d_records) {
        return fail(MBE_STATUS_INVALID_ARGUMENT,
                    "mbx_stream_expanded: the last mbx_expand_records() on this stream was not for this codec / batch / record array");
    }
    return launch_stream(c, (slot.launches++ & 1u) != 0u, codec, S, T, d_records, slot.workspace, d_state, d_rng, d_pcm16, d_pcmf,
                         d_results, stream, nullptr, d_resident);
}

int mbx_process_records(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state,
                        mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                        void* stream) {
    REQUIRE_CTX(c);
    if (!stream_args_ok(codec, S, T, d_records, d_state, d_rng)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0 || T == 0) {
        return 0;
    }
    // The lock is held across the launches of this call: the workspace of `stream` cannot be grown (freed) by another
    // thread between the expand launch and the stream launch, and two threads that share one hipStream_t cannot
    // interleave their expand / stream pairs.
    std::lock_guard<std::mutex> lock(c->mu);
    StreamSlot& slot = c->slots[stream];
    if (needs_workspace(codec, S, T)) {
        int rc = ensure_workspace(c, slot, (size_t)S * (size_t)T, stream);
        if (rc < 0) {
            return rc;
        }
        slot.exp_codec = -1;   // the rows are about to be replaced
    }
    return run_stream_stage(c, slot.launches++, codec, S, T, d_records, slot.workspace, d_state, d_rng, d_pcm16, d_pcmf, d_results,
                            stream, nullptr, nullptr, &slot);
}

int mbx_process_records_ws(int codec, int S, int T, const mbx_param_record* d_records, mbe_parms* d_state,
                           mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results,
                           void* d_workspace, size_t workspace_bytes, void* stream) {
    REQUIRE_CTX(c);
    if (!stream_args_ok(codec, S, T, d_records, d_state, d_rng)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0 || T == 0) {
        return 0;
    }
    if (needs_workspace(codec, S, T) && (!d_workspace || workspace_bytes < mbx_workspace_bytes((size_t)S * (size_t)T))) {
        return fail(MBE_STATUS_INVALID_ARGUMENT, "mbx_process_records_ws: workspace missing or smaller than mbx_workspace_bytes(S*T)");
    }
    unsigned order;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        order = c->slots[stream].launches++;
    }
    return run_stream_stage(c, order, codec, S, T, d_records, static_cast<mbx::FrameParams*>(d_workspace), d_state, d_rng, d_pcm16,
                            d_pcmf, d_results, stream);
}

// FEC stage of a hard-decision batch; *stream_codec = the codec of the stream stage that follows
static int launch_fec(int codec, const uint8_t* d_frames, size_t n, mbx_param_record* d_records, void* stream, int* stream_codec) {
    *stream_codec = codec;
    if (codec == MBX_CODEC_IMBE7200X4400) {
        return mbx_fec_imbe7200x4400(d_frames, n, d_records, stream);
    }
    if (codec == MBX_CODEC_IMBE7100X4400) {   // own front end; the records are in 7200x4400 order
        *stream_codec = MBX_CODEC_IMBE7200X4400;
        return mbx_fec_imbe7100x4400(d_frames, n, d_records, stream);
    }
    if (codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400) {   // shared AMBE FEC front end
        return mbx_fec_ambe3600x2450(d_frames, n, d_records, stream);
    }
    return MBE_STATUS_INVALID_ARGUMENT;
}

int mbx_process_batch(int codec, int S, int T, const uint8_t* d_frames, mbe_parms* d_state, mbx_stream_rng* d_rng,
                      int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records,
                      void* stream) {
    if (!d_frames || !d_records || S < 0 || T < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int stream_codec;
    int rc;
    if (try_fused_one(codec, S, T, d_frames, d_records, d_state, d_rng, d_pcm16, d_pcmf, d_results, stream, nullptr, nullptr, &rc)) {
        return rc;
    }
    rc = launch_fec(codec, d_frames, (size_t)S * (size_t)T, d_records, stream, &stream_codec);
    if (rc < 0) {
        return rc;
    }
    return mbx_process_records(stream_codec, S, T, d_records, d_state, d_rng, d_pcm16, d_pcmf, d_results, stream);
}

int mbx_stage_in(void* d_dst, const void* pinned_src, size_t bytes, void* stream) {
    if (!d_dst || !pinned_src) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (bytes == 0) {
        return 0;
    }
    if ((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(pinned_src)) & 15u) {
        return fail(MBE_STATUS_INVALID_ARGUMENT, "mbx_stage_in: both buffers must be 16-byte aligned");
    }
    const size_t threads = (bytes >> 4) + 1;
    hipLaunchKernelGGL(mbx::stage_in_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint8_t*>(pinned_src), static_cast<uint8_t*>(d_dst), bytes);
    return check_launch("stage_in_kernel");
}

static int launch_frame(int codec, const uint8_t* d_frame, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf,
                        mbe_process_result* d_result, mbx_param_record* d_record, uint32_t* d_done, uint32_t token, void* stream,
                        const mbx::FrameShadow& shadow) {
    REQUIRE_CTX(c);
    if (!d_frame || !d_state || !d_rng || !d_record || codec < MBX_CODEC_IMBE7200X4400 || codec > MBX_CODEC_AMBE3600X2400) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbx::DeviceTables tabs = c->tabs;
    tabs.reverse = 0;
    tabs.stream_map = nullptr;
    tabs.resident = nullptr;
    if (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400) {
        hipLaunchKernelGGL(mbx::imbe_frame_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, codec, d_frame, d_record, d_state, d_rng,
                           d_pcm16, d_pcmf, d_result, d_done, token, tabs, shadow);
    } else if (codec == MBX_CODEC_AMBE3600X2400) {
        hipLaunchKernelGGL(mbx::ambe2400_frame_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_frame, d_record, d_state, d_rng,
                           d_pcm16, d_pcmf, d_result, d_done, token, tabs, shadow);
    } else {
        hipLaunchKernelGGL(mbx::ambe_frame_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_frame, d_record, d_state, d_rng, d_pcm16,
                           d_pcmf, d_result, d_done, token, tabs, shadow);
    }
    return check_launch("frame_kernel");
}

int mbx_process_frame(int codec, const uint8_t* d_frame, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf,
                      mbe_process_result* d_result, mbx_param_record* d_record, uint32_t* d_done, uint32_t token, void* stream) {
    return launch_frame(codec, d_frame, d_state, d_rng, d_pcm16, d_pcmf, d_result, d_record, d_done, token, stream, mbx::FrameShadow{});
}

int mbx_process_frame_shadow(int codec, const uint8_t* d_frame, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16, float* d_pcmf,
                             mbe_process_result* d_result, mbx_param_record* d_record, uint32_t* d_done, uint32_t token,
                             mbe_parms* d_shadow_state, mbx_stream_rng* d_shadow_rng, uint32_t* d_shadow_ok, int use_shadow,
                             const uint8_t* h_frame, void* stream) {
    if (!d_shadow_state || !d_shadow_rng || !d_shadow_ok) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbx::FrameShadow shadow;
    shadow.state = d_shadow_state;
    shadow.rng = d_shadow_rng;
    shadow.ok = d_shadow_ok;
    shadow.use = use_shadow ? 1u : 0u;
    if (h_frame) {   // the frame's bytes travel with the launch
        const size_t nb = (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400) ? 18u : 9u;
        memcpy(shadow.frame_words, h_frame, nb);
        shadow.have_frame = 1u;
    }
    return launch_frame(codec, d_frame, d_state, d_rng, d_pcm16, d_pcmf, d_result, d_record, d_done, token, stream, shadow);
}

int mbx_frame_server_start(mbx_frame_mailbox* mailbox, unsigned idle_us, mbe_parms* d_state, mbx_stream_rng* d_rng, int16_t* d_pcm16,
                           float* d_pcmf, mbe_process_result* d_result, mbx_param_record* d_record, mbe_parms* d_shadow_state,
                           mbx_stream_rng* d_shadow_rng, uint32_t* d_shadow_ok, void* stream) {
    REQUIRE_CTX(c);
    if (!mailbox || (reinterpret_cast<uintptr_t>(mailbox) & 63u) || idle_us == 0 || idle_us > 1000000u || !d_state || !d_rng || !d_record
        || ((d_shadow_state != nullptr) != (d_shadow_rng != nullptr)) || ((d_shadow_state != nullptr) != (d_shadow_ok != nullptr))) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbx::FrameShadow shadow;
    shadow.state = d_shadow_state;
    shadow.rng = d_shadow_rng;
    shadow.ok = d_shadow_ok;
    mbx::DeviceTables tabs = c->tabs;
    tabs.reverse = 0;
    tabs.stream_map = nullptr;
    tabs.resident = nullptr;
    hipLaunchKernelGGL(mbx::frame_server_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mailbox, idle_us * 100u, d_state, d_rng,
                       d_pcm16, d_pcmf, d_result, d_record, tabs, shadow);
    return check_launch("frame_server_kernel");
}

int mbx_process_batch_ws(int codec, int S, int T, const uint8_t* d_frames, mbe_parms* d_state, mbx_stream_rng* d_rng,
                         int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records,
                         void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!d_frames || !d_records || S < 0 || T < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int stream_codec;
    int rc;
    if (try_fused_one(codec, S, T, d_frames, d_records, d_state, d_rng, d_pcm16, d_pcmf, d_results, stream, nullptr, nullptr, &rc, false)) {
        return rc;   // (the form that needs no workspace: the caller's has no flag words)
    }
    rc = launch_fec(codec, d_frames, (size_t)S * (size_t)T, d_records, stream, &stream_codec);
    if (rc < 0) {
        return rc;
    }
    return mbx_process_records_ws(stream_codec, S, T, d_records, d_state, d_rng, d_pcm16, d_pcmf, d_results, d_workspace,
                                  workspace_bytes, stream);
}

int mbx_process_batch_indexed(int codec, int S, int T, const int32_t* d_stream_index, const uint8_t* d_frames,
                              mbe_parms* d_state_pool, mbx_stream_rng* d_rng_pool, int16_t* d_pcm16, float* d_pcmf,
                              mbe_process_result* d_results, mbx_param_record* d_records, void* stream) {
    REQUIRE_CTX(c);
    if (!d_frames || !d_records || !d_stream_index || !d_state_pool || !d_rng_pool || S < 0 || T < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0 || T == 0) {
        return 0;
    }
    int stream_codec;
    int rc;
    if (try_fused_one(codec, S, T, d_frames, d_records, d_state_pool, d_rng_pool, d_pcm16, d_pcmf, d_results, stream, d_stream_index, nullptr,
                      &rc)) {
        return rc;
    }
    rc = launch_fec(codec, d_frames, (size_t)S * (size_t)T, d_records, stream, &stream_codec);
    if (rc < 0) {
        return rc;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    StreamSlot& slot = c->slots[stream];
    if (needs_workspace(stream_codec, S, T)) {
        rc = ensure_workspace(c, slot, (size_t)S * (size_t)T, stream);
        if (rc < 0) {
            return rc;
        }
        slot.exp_codec = -1;
    }
    return run_stream_stage(c, slot.launches++, stream_codec, S, T, d_records, slot.workspace, d_state_pool, d_rng_pool, d_pcm16,
                            d_pcmf, d_results, stream, d_stream_index, nullptr, &slot);
}

// ---- resident state (sessions, queue mode) ------------------------------------------------------------------------------
// The drop-in batch calls above keep the reference's three structs per stream whole after every launch: a caller may read
// them at any time.  A caller that OWNS the state for a while (a session; the queue mode's device pool) does not need that:
// after every ordinary frame prev_mp_enhanced is a field-for-field copy of cur_mp (ref src/imbe/imbe7200x4400.c:842-856,
// src/ambe/ambe3600x2450.c:790-800), so d_resident[slot] != 0 says "elided" and the kernels neither write nor read that struct;
// and of prev_mp a launch fetches only what the decode reads.  At T = 1 that is 8.9 KB of state traffic per frame instead of
// 14.1 KB.  mbx_resident_materialize() brings the triplets back to the ABI form (bit-identical to what mbx_process_batch
// would have left), after which d_resident is zero again.
int mbx_process_batch_resident(int codec, int S, int T, const int32_t* d_stream_index, const uint8_t* d_frames,
                               mbe_parms* d_state_pool, uint32_t* d_resident, mbx_stream_rng* d_rng_pool, int16_t* d_pcm16,
                               float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records, void* stream) {
    REQUIRE_CTX(c);
    if (!d_frames || !d_records || !d_state_pool || !d_resident || !d_rng_pool || S < 0 || T < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0 || T == 0) {
        return 0;
    }
    int stream_codec;
    int rc;
    if (try_fused_one(codec, S, T, d_frames, d_records, d_state_pool, d_rng_pool, d_pcm16, d_pcmf, d_results, stream, d_stream_index,
                      d_resident, &rc)) {
        return rc;
    }
    rc = launch_fec(codec, d_frames, (size_t)S * (size_t)T, d_records, stream, &stream_codec);
    if (rc < 0) {
        return rc;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    StreamSlot& slot = c->slots[stream];
    if (needs_workspace(stream_codec, S, T)) {
        rc = ensure_workspace(c, slot, (size_t)S * (size_t)T, stream);
        if (rc < 0) {
            return rc;
        }
        slot.exp_codec = -1;
    }
    return run_stream_stage(c, slot.launches++, stream_codec, S, T, d_records, slot.workspace, d_state_pool, d_rng_pool, d_pcm16,
                            d_pcmf, d_results, stream, d_stream_index, d_resident);
}

namespace mbx {
// prev_mp_enhanced := cur_mp for every listed slot whose struct is elided; one wavefront per slot
__global__ void __launch_bounds__(64)
resident_materialize_kernel(int n, const int32_t* __restrict__ index, mbe_parms* __restrict__ state, uint32_t* __restrict__ resident) {
    const int i = blockIdx.x;
    if (i >= n) {
        return;
    }
    const size_t slot = index ? (size_t)index[i] : (size_t)i;
    if (resident[slot] == 0u) {
        return;
    }
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&state[3 * slot + 0]);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&state[3 * slot + 2]);
    for (int k = threadIdx.x; k < (int)(sizeof(mbe_parms) / 4); k += 64) {
        dst[k] = src[k];
    }
    if (threadIdx.x == 0) {
        resident[slot] = 0u;
    }
}
}  // namespace mbx

int mbx_resident_materialize(int n, const int32_t* d_stream_index, mbe_parms* d_state_pool, uint32_t* d_resident, void* stream) {
    REQUIRE_CTX(c);
    (void)c;
    if (!d_state_pool || !d_resident || n < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::resident_materialize_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, n, d_stream_index,
                       d_state_pool, d_resident);
    return check_launch("resident_materialize_kernel");
}

int mbx_process_batch_soft(int codec, int S, int T, const mbe_soft_bit* d_soft, mbe_parms* d_state, mbx_stream_rng* d_rng,
                           int16_t* d_pcm16, float* d_pcmf, mbe_process_result* d_results, mbx_param_record* d_records,
                           void* stream) {
    if (!d_soft || !d_records || S < 0 || T < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = mbx_fec_soft(codec, d_soft, (size_t)S * (size_t)T, d_records, stream);
    if (rc < 0) {
        return rc;
    }
    if (codec == MBX_CODEC_IMBE7100X4400) {   // the records are in 7200x4400 order
        codec = MBX_CODEC_IMBE7200X4400;
    }
    return mbx_process_records(codec, S, T, d_records, d_state, d_rng, d_pcm16, d_pcmf, d_results, stream);
}

int mbx_synthesize_speech(int S, mbe_parms* d_cur, mbe_parms* d_prev, mbx_stream_rng* d_rng, float* d_pcmf,
                          int16_t* d_pcm16, void* stream) {
    REQUIRE_CTX(c);
    if (!d_cur || !d_prev || !d_rng || S < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::synth_speech_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, d_cur, d_prev,
                       d_rng, d_pcmf, d_pcm16, c->tabs);
    return check_launch("synth_speech_kernel");
}

int mbx_floattoshort(const float* d_in, int16_t* d_out, size_t nframes, void* stream) {
    REQUIRE_CTX(c);
    if (!d_in || !d_out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (nframes == 0) {
        return 0;
    }
    const size_t nsamples = nframes * 160u;
    const unsigned grid = (unsigned)((nsamples / 2 + 255) / 256);
    hipLaunchKernelGGL(mbx::floattoshort_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_in, d_out, nsamples);
    return check_launch("floattoshort_kernel");
}

int mbx_result_histogram(const mbe_process_result* d_results, size_t n, mbx_result_hist* d_hist, void* stream) {
    static_assert(sizeof(mbx_result_hist) == mbx::kResultHistWords * sizeof(unsigned long long), "mbx_result_hist is 14 64-bit counters");
    REQUIRE_CTX(c);
    if (!d_results || !d_hist) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    const size_t blocks = (n + 1023) / 1024;   // four frames per thread, at most 4,096 workgroups
    hipLaunchKernelGGL(mbx::result_histogram_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, d_results, n,
                       reinterpret_cast<unsigned long long*>(d_hist));
    return check_launch("result_histogram_kernel");
}

int mbx_spectral_amp_enhance(int S, mbe_parms* d_parms, void* stream) {
    REQUIRE_CTX(c);
    if (!d_parms || S < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::enhance_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, d_parms);
    return check_launch("enhance_kernel");
}

int mbx_adaptive_smoothing(int S, mbe_parms* d_cur, const mbe_parms* d_prev, void* stream) {
    REQUIRE_CTX(c);
    if (!d_cur || !d_prev || S < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::smoothing_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, d_cur, d_prev);
    return check_launch("smoothing_kernel");
}

int mbx_comfort_noise(int S, mbx_stream_rng* d_rng, float* d_pcmf, int16_t* d_pcm16, void* stream) {
    REQUIRE_CTX(c);
    if (!d_rng || S < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::comfort_noise_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, d_rng, d_pcmf,
                       d_pcm16);
    return check_launch("comfort_noise_kernel");
}

int mbx_synthesize_tone(int S, const mbx_param_record* d_records, const int32_t* d_dstar_ids, mbe_parms* d_cur, float* d_pcmf,
                        int16_t* d_pcm16, void* stream) {
    REQUIRE_CTX(c);
    if (!d_cur || S < 0 || (!d_records && !d_dstar_ids)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::tone_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, d_records, d_dstar_ids, d_cur,
                       d_pcmf, d_pcm16, c->tabs.tones_off);
    return check_launch("tone_kernel");
}

int mbx_state_copy(int S, mbe_parms* d_state, void* stream) {
    REQUIRE_CTX(c);
    if (!d_state || S < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (S == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::state_copy_kernel, dim3((unsigned)S), dim3(64), 0, (hipStream_t)stream, S, d_state);
    return check_launch("state_copy_kernel");
}

int mbx_ecc_words(int kind, const uint32_t* d_in, size_t n, uint32_t* d_out, int32_t* d_errs, void* stream) {
    REQUIRE_CTX(c);
    if (!d_in || !d_out || kind < 0 || kind > 2) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::ecc_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kind,
                       d_in, n, d_out, d_errs, c->tabs);
    return check_launch("ecc_words_kernel");
}

#ifdef MBX_ABLATE
// development build only (make ablate -> libmbx_hip_ablate.so): timing-only stage mask for every initialised device
void mbx_debug_set_ablation(int mask) {
    for (int dev = 0; dev < kMaxDevices; ++dev) {
        g_ctx[dev].tabs.ablate = mask;
    }
}
#endif

int mbx_pack_cells(int codec, const char* d_cells, size_t n, uint8_t* d_packed, int32_t* d_status, void* stream) {
    REQUIRE_CTX(c);
    (void)c;
    if (!d_cells || !d_packed || codec < MBX_CODEC_IMBE7200X4400 || codec > MBX_CODEC_AMBE3600X2400) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::pack_cells_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, codec, d_cells, n,
                       d_packed, d_status);
    return check_launch("pack_cells_kernel");
}

int mbx_fec_stage(int codec, int stage, const void* d_in, size_t n, uint8_t* d_frames_out, mbx_param_record* d_out, void* stream) {
    REQUIRE_CTX(c);
    if (!d_in || codec < MBX_CODEC_IMBE7200X4400 || codec > MBX_CODEC_AMBE3600X2400
        || !(stage == MBX_STAGE_C0 || stage == MBX_STAGE_DEMODULATE || stage == MBX_STAGE_DATA || stage == MBX_STAGE_CONVERT7100)
        || ((stage == MBX_STAGE_DATA || stage == MBX_STAGE_CONVERT7100) && !d_out)
        || ((stage == MBX_STAGE_C0 || stage == MBX_STAGE_DEMODULATE) && !d_frames_out)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    hipLaunchKernelGGL(mbx::fec_stage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codec, stage,
                       static_cast<const uint8_t*>(d_in), n, d_frames_out, d_out, c->tabs);
    return check_launch("fec_stage_kernel");
}

int mbx_decode_parms(int codec, const mbx_param_record* d_records, size_t n, mbe_parms* d_cur, mbe_parms* d_prev, int32_t* d_rc,
                     void* stream) {
    REQUIRE_CTX(c);
    if (!d_records || !d_cur || !d_prev || !d_rc || !expand_codec_ok(codec) || n > 0x7fffffffu) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (n == 0) {
        return 0;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    StreamSlot& slot = c->slots[stream];
    int rc = ensure_workspace(c, slot, n, stream);
    if (rc < 0) {
        return rc;
    }
    slot.exp_codec = -1;
    rc = launch_expand(c, codec, d_records, n, slot.workspace, stream);
    if (rc < 0) {
        return rc;
    }
    hipLaunchKernelGGL(mbx::decode_parms_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, codec, (int)n, slot.workspace,
                       d_cur, d_prev, d_rc, c->tabs);
    return check_launch("decode_parms_kernel");
}

#ifdef MBX_TESTING
// libmbx_hip_testing.so only (make testing; the tests load it through MBX_HIP_LIBRARY in a child process): fault injection for the
// fall-back path of the one-launch kernels, which no ordinary launch has ever taken.  every = 2^k > 0 makes the front blocks of chunks
// 0, 2^k, 2 * 2^k, ... do nothing, so that their stream blocks wait out their ~40 us and decode their own frames; 0 switches it off.
extern "C" int mbx_testing_set_front_skip(int every) {
    REQUIRE_CTX(c);
    if (every < 0 || (every & (every - 1)) != 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    c->tabs.front_skip = every;
    return 0;
}
#endif

// diagnostics: how many stream blocks of the one-launch kernel have expanded their own frame since the stream's workspace was
// allocated (a stream block does that when its front block's rows are not there in time); synchronises the stream.  -1: no workspace.
long long mbx_front_fallbacks(void* stream) {
    int crc;
    Context* c = current_ctx(&crc);
    if (!c) {
        return crc;
    }
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->slots.find(stream);
    if (it == c->slots.end() || !it->second.flags) {
        return -1;
    }
    uint32_t v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess
        || hipMemcpy(&v, it->second.flags + (it->second.frames + 7) / 8, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return MBX_ENODEVICE;
    }
    return (long long)v;
}

// 0, or the slice length in frames a launch of this shape is cut into (mbx_process_records and the batch calls on top of it)
int mbx_launch_slices(int codec, int S, int T) {
    int crc;
    Context* c = current_ctx(&crc);
    if (!c || !(T >= kLdsResidentMinFrames && lds_resident_enabled())) {
        return 0;
    }
    const int sc = codec == MBX_CODEC_IMBE7100X4400 ? MBX_CODEC_IMBE7200X4400 : codec;
    return choose_slice_frames(S, T, lds_kernel_waves_per_simd(sc) * c->simds);
}

// the dominant kernel of mbx_process_batch / _resident for a batch shape (frames 4-byte aligned, as device allocations are)
const char* mbx_batch_kernel_name(int codec, int S, int T, int resident) {
    if (fused_one_ok(codec, S, T, nullptr, resident != 0)) {
        if (ambe_codec(codec)) {
            return codec == MBX_CODEC_AMBE3600X2400 ? (resident ? "ambe2400_one_launch_kernel_res" : "ambe2400_one_launch_kernel")
                                                    : (resident ? "ambe_one_launch_kernel_res" : "ambe_one_launch_kernel");
        }
        if (codec == MBX_CODEC_IMBE7100X4400) {
            return resident ? "imbe7100_stream_kernel_res1_fused" : "imbe7100_stream_kernel_one_fused";
        }
        if (one_launch_form(codec)) {
            return resident ? "imbe_one_launch_kernel_res" : "imbe_one_launch_kernel";
        }
        return resident ? "imbe_stream_kernel_res1_fused" : "imbe_stream_kernel_one_fused";
    }
    if (!resident && mbx_launch_slices(codec, S, T) > 0) {
        return (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400) ? "imbe_stream_kernel_lds_slice"
               : (codec == MBX_CODEC_AMBE3600X2400)                                    ? "ambe2400_stream_kernel_lds_slice"
                                                                                        : "ambe_stream_kernel_lds_slice";
    }
    return mbx_stream_kernel_name(codec, resident ? -T : T);
}

const char* mbx_stream_kernel_name(int codec, int T) {
    if (T < 0) {   // the instances of the resident launches (mbx_process_batch_resident) with -T frames per stream
        const bool one = T == -1 && res1_enabled();
        return (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400) ? (one ? "imbe_stream_kernel_res1" : "imbe_stream_kernel_res")
               : (codec == MBX_CODEC_AMBE3600X2400)                                    ? (one ? "ambe2400_stream_kernel_res1" : "ambe2400_stream_kernel_res")
                                                                                        : (one ? "ambe_stream_kernel_res1" : "ambe_stream_kernel_res");
    }
    const bool lds = T >= kLdsResidentMinFrames && lds_resident_enabled();
    if (codec == MBX_CODEC_IMBE7200X4400 || codec == MBX_CODEC_IMBE7100X4400) {
        return lds ? "imbe_stream_kernel_lds" : (T == 1 ? "imbe_stream_kernel_one" : "imbe_stream_kernel");
    }
    if (codec == MBX_CODEC_AMBE3600X2400) {
        return lds ? "ambe2400_stream_kernel_lds" : (T == 1 ? "ambe2400_stream_kernel_one" : "ambe2400_stream_kernel");
    }
    return lds ? "ambe_stream_kernel_lds" : (T == 1 ? "ambe_stream_kernel_one" : "ambe_stream_kernel");
}

// ---- host-buffer conveniences ------------------------------------------------------------


int mbx_fec_host(int codec, const uint8_t* frames, size_t n, mbx_param_record* records) {
    REQUIRE_CTX(c);
    if (!frames || !records) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const size_t fb = (codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400) ? MBX_AMBE_FRAME_BYTES : MBX_IMBE_FRAME_BYTES;
    DevBuf df, dr;
    HIP_TRY(df.alloc(n * fb));
    HIP_TRY(dr.alloc(n * sizeof(mbx_param_record)));
    HIP_TRY(hipMemcpy(df.p, frames, n * fb, hipMemcpyHostToDevice));
    int rc = (codec == MBX_CODEC_IMBE7200X4400)   ? mbx_fec_imbe7200x4400(df.as<uint8_t>(), n, dr.as<mbx_param_record>(), nullptr)
             : (codec == MBX_CODEC_IMBE7100X4400) ? mbx_fec_imbe7100x4400(df.as<uint8_t>(), n, dr.as<mbx_param_record>(), nullptr)
             : (codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400)
                 ? mbx_fec_ambe3600x2450(df.as<uint8_t>(), n, dr.as<mbx_param_record>(), nullptr)
                 : MBE_STATUS_INVALID_ARGUMENT;
    if (rc < 0) {
        return rc;
    }
    HIP_TRY(hipMemcpy(records, dr.p, n * sizeof(mbx_param_record), hipMemcpyDeviceToHost));
    return 0;
}

static size_t soft_cells(int codec) {
    return codec == MBX_CODEC_IMBE7200X4400 ? MBX_IMBE_SOFT_BITS : (codec == MBX_CODEC_IMBE7100X4400 ? MBX_IMBE7100_SOFT_BITS : MBX_AMBE_SOFT_BITS);   // both AMBE codecs: 96
}

// frames in (hard: packed bytes, soft: mbe_soft_bit arrays), everything else as mbx_process_batch
static int process_batch_host_impl(int codec, int S, int T, const void* frames, size_t frame_bytes, bool soft, mbe_parms* state,
                                   mbx_stream_rng* rng, int16_t* pcm16, float* pcmf, mbe_process_result* results,
                                   mbx_param_record* records) {
    const size_t n = (size_t)S * (size_t)T;
    DevBuf df, ds, dg, d16, dfl, dres, drec;
    HIP_TRY(df.alloc(n * frame_bytes));
    HIP_TRY(ds.alloc((size_t)S * 3 * sizeof(mbe_parms)));
    HIP_TRY(dg.alloc((size_t)S * sizeof(mbx_stream_rng)));
    HIP_TRY(d16.alloc(n * 160 * sizeof(int16_t)));
    HIP_TRY(dfl.alloc(n * 160 * sizeof(float)));
    HIP_TRY(dres.alloc(n * sizeof(mbe_process_result)));
    HIP_TRY(drec.alloc(n * sizeof(mbx_param_record)));
    HIP_TRY(hipMemcpy(df.p, frames, n * frame_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ds.p, state, (size_t)S * 3 * sizeof(mbe_parms), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dg.p, rng, (size_t)S * sizeof(mbx_stream_rng), hipMemcpyHostToDevice));
    int rc = soft ? mbx_process_batch_soft(codec, S, T, df.as<mbe_soft_bit>(), ds.as<mbe_parms>(), dg.as<mbx_stream_rng>(),
                                           pcm16 ? d16.as<int16_t>() : nullptr, pcmf ? dfl.as<float>() : nullptr,
                                           results ? dres.as<mbe_process_result>() : nullptr, drec.as<mbx_param_record>(), nullptr)
                  : mbx_process_batch(codec, S, T, df.as<uint8_t>(), ds.as<mbe_parms>(), dg.as<mbx_stream_rng>(),
                                      pcm16 ? d16.as<int16_t>() : nullptr, pcmf ? dfl.as<float>() : nullptr,
                                      results ? dres.as<mbe_process_result>() : nullptr, drec.as<mbx_param_record>(), nullptr);
    if (rc < 0) {
        return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(state, ds.p, (size_t)S * 3 * sizeof(mbe_parms), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rng, dg.p, (size_t)S * sizeof(mbx_stream_rng), hipMemcpyDeviceToHost));
    if (pcm16) {
        HIP_TRY(hipMemcpy(pcm16, d16.p, n * 160 * sizeof(int16_t), hipMemcpyDeviceToHost));
    }
    if (pcmf) {
        HIP_TRY(hipMemcpy(pcmf, dfl.p, n * 160 * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (results) {
        HIP_TRY(hipMemcpy(results, dres.p, n * sizeof(mbe_process_result), hipMemcpyDeviceToHost));
    }
    if (records) {
        HIP_TRY(hipMemcpy(records, drec.p, n * sizeof(mbx_param_record), hipMemcpyDeviceToHost));
    }
    return 0;
}

int mbx_process_batch_host(int codec, int S, int T, const uint8_t* frames, mbe_parms* state, mbx_stream_rng* rng,
                           int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records) {
    REQUIRE_CTX(c);
    if (!frames || !state || !rng || S < 0 || T < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const size_t fb = (codec == MBX_CODEC_AMBE3600X2450 || codec == MBX_CODEC_AMBE3600X2400) ? MBX_AMBE_FRAME_BYTES : MBX_IMBE_FRAME_BYTES;
    return process_batch_host_impl(codec, S, T, frames, fb, false, state, rng, pcm16, pcmf, results, records);
}

int mbx_process_batch_soft_host(int codec, int S, int T, const mbe_soft_bit* soft, mbe_parms* state, mbx_stream_rng* rng,
                                int16_t* pcm16, float* pcmf, mbe_process_result* results, mbx_param_record* records) {
    REQUIRE_CTX(c);
    if (!soft || !state || !rng || S < 0 || T < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const size_t cells = soft_cells(codec);
    int rc = mbx_validate_soft_bits(soft, (size_t)S * (size_t)T * cells);
    if (rc < 0) {
        return rc;
    }
    return process_batch_host_impl(codec, S, T, soft, cells * sizeof(mbe_soft_bit), true, state, rng, pcm16, pcmf, results,
                                   records);
}

int mbx_fec_soft_host(int codec, const mbe_soft_bit* soft, size_t n, mbx_param_record* records) {
    REQUIRE_CTX(c);
    if (!soft || !records) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const size_t cells = soft_cells(codec);
    int rc = mbx_validate_soft_bits(soft, n * cells);
    if (rc < 0) {
        return rc;
    }
    DevBuf df, dr;
    HIP_TRY(df.alloc(n * cells * sizeof(mbe_soft_bit)));
    HIP_TRY(dr.alloc(n * sizeof(mbx_param_record)));
    HIP_TRY(hipMemcpy(df.p, soft, n * cells * sizeof(mbe_soft_bit), hipMemcpyHostToDevice));
    rc = mbx_fec_soft(codec, df.as<mbe_soft_bit>(), n, dr.as<mbx_param_record>(), nullptr);
    if (rc < 0) {
        return rc;
    }
    HIP_TRY(hipMemcpy(records, dr.p, n * sizeof(mbx_param_record), hipMemcpyDeviceToHost));
    return 0;
}

int mbx_ecc_soft_words_host(int kind, const mbe_soft_bit* in, size_t n, uint32_t* out, int32_t* errs) {
    REQUIRE_CTX(c);
    if (!in || !out || kind < 0 || kind > 2) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const size_t width = kind == 0 ? 23 : 15;
    int rc = mbx_validate_soft_bits(in, n * width);
    if (rc < 0) {
        return rc;
    }
    DevBuf di, dout, de;
    HIP_TRY(di.alloc(n * width * sizeof(mbe_soft_bit) + 2));
    HIP_TRY(dout.alloc(n * sizeof(uint32_t)));
    HIP_TRY(de.alloc(n * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(di.p, in, n * width * sizeof(mbe_soft_bit), hipMemcpyHostToDevice));
    rc = mbx_ecc_soft_words(kind, di.as<mbe_soft_bit>(), n, dout.as<uint32_t>(), de.as<int32_t>(), nullptr);
    if (rc < 0) {
        return rc;
    }
    HIP_TRY(hipMemcpy(out, dout.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (errs) {
        HIP_TRY(hipMemcpy(errs, de.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int mbx_synthesize_speech_host(int S, mbe_parms* cur, mbe_parms* prev, mbx_stream_rng* rng, float* pcmf, int16_t* pcm16) {
    REQUIRE_CTX(c);
    if (!cur || !prev || !rng || S < 0) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    DevBuf dc, dp, dg, dfl, d16;
    HIP_TRY(dc.alloc((size_t)S * sizeof(mbe_parms)));
    HIP_TRY(dp.alloc((size_t)S * sizeof(mbe_parms)));
    HIP_TRY(dg.alloc((size_t)S * sizeof(mbx_stream_rng)));
    HIP_TRY(dfl.alloc((size_t)S * 160 * sizeof(float)));
    HIP_TRY(d16.alloc((size_t)S * 160 * sizeof(int16_t)));
    HIP_TRY(hipMemcpy(dc.p, cur, (size_t)S * sizeof(mbe_parms), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dp.p, prev, (size_t)S * sizeof(mbe_parms), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dg.p, rng, (size_t)S * sizeof(mbx_stream_rng), hipMemcpyHostToDevice));
    int rc = mbx_synthesize_speech(S, dc.as<mbe_parms>(), dp.as<mbe_parms>(), dg.as<mbx_stream_rng>(),
                                   pcmf ? dfl.as<float>() : nullptr, pcm16 ? d16.as<int16_t>() : nullptr, nullptr);
    if (rc < 0) {
        return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cur, dc.p, (size_t)S * sizeof(mbe_parms), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(prev, dp.p, (size_t)S * sizeof(mbe_parms), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rng, dg.p, (size_t)S * sizeof(mbx_stream_rng), hipMemcpyDeviceToHost));
    if (pcmf) {
        HIP_TRY(hipMemcpy(pcmf, dfl.p, (size_t)S * 160 * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (pcm16) {
        HIP_TRY(hipMemcpy(pcm16, d16.p, (size_t)S * 160 * sizeof(int16_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int mbx_floattoshort_host(const float* in, int16_t* out, size_t nframes) {
    REQUIRE_CTX(c);
    if (!in || !out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    DevBuf di, dout;
    HIP_TRY(di.alloc(nframes * 160 * sizeof(float)));
    HIP_TRY(dout.alloc(nframes * 160 * sizeof(int16_t)));
    HIP_TRY(hipMemcpy(di.p, in, nframes * 160 * sizeof(float), hipMemcpyHostToDevice));
    int rc = mbx_floattoshort(di.as<float>(), dout.as<int16_t>(), nframes, nullptr);
    if (rc < 0) {
        return rc;
    }
    HIP_TRY(hipMemcpy(out, dout.p, nframes * 160 * sizeof(int16_t), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
