// mbe_shim.cpp -- libmbe_neo_amd.so: the reference's per-frame `mbe_*` C API (include/mbe_neo_amd.h)
// on top of the HIP launcher (include/mbx.h).  Host logic only: argument/result validation and
// bookkeeping follow the reference call for call; every number that the reference computes is
// computed by a kernel.  No CPU compute fallback -- if the GPU cannot be initialised the process
// aborts with a message.
//
// Per call: one frame is staged through a small per-thread device scratch area (allocated once
// per thread), the kernels run on a per-thread stream, outputs are copied back.  The reference's
// thread-local synthesis RNG state (src/core/mbe_adaptive.c:29-30, src/core/mbe_unvoiced_fft.c:29-30)
// is a thread_local mbx_stream_rng here.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "mbe_neo_amd.h"
#include "mbx.h"

namespace {

[[noreturn]] void die(const char* what) {
    fprintf(stderr, "libmbe_neo_amd: %s: %s\nlibmbe_neo_amd has no CPU fallback; aborting.\n", what, mbx_last_error());
    abort();
}

std::once_flag g_once;
int g_device = 0;   // MBX_DEVICE: the one device the per-frame API runs on

void init_once() {
    std::string path;
    if (const char* env = getenv("MBX_TABLES")) {
        path = env;
    } else {
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&init_once), &info) && info.dli_fname) {
            path = info.dli_fname;
            const size_t slash = path.find_last_of('/');
            path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/data/mbx_tables.bin";
        }
    }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "libmbe_neo_amd: cannot open table blob '%s' (set MBX_TABLES)\n", path.c_str());
        abort();
    }
    std::vector<unsigned char> blob(sizeof(mbx_tables) + 1);
    const size_t n = fread(blob.data(), 1, blob.size(), f);
    fclose(f);
    const char* dev = getenv("MBX_DEVICE");
    g_device = dev ? atoi(dev) : 0;
    if (mbx_init(g_device, blob.data(), n) != 0) {
        die("mbx_init");
    }
}

#define HIP_OK(expr)                                   \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) {                        \
            fprintf(stderr, "libmbe_neo_amd: %s: %s\n", #expr, hipGetErrorString(e_)); \
            abort();                                   \
        }                                              \
    } while (0)

// Per-thread device scratch for one frame.
struct Slot {
    hipStream_t       stream = nullptr;
    uint8_t*          frame = nullptr;    // 18 bytes
    mbx_param_record* rec = nullptr;
    mbe_parms*        state = nullptr;    // 3
    mbx_stream_rng*   rng = nullptr;
    float*            pcmf = nullptr;     // 160
    int16_t*          pcm16 = nullptr;    // 160
    mbe_process_result* res = nullptr;
    uint32_t*         words = nullptr;    // 4: in, out, errs
    mbe_soft_bit*     soft = nullptr;     // one soft frame (184 cells)
    Slot() {
        std::call_once(g_once, init_once);
        HIP_OK(hipSetDevice(g_device));   // HIP's current device is per thread: every thread that decodes selects the library's
        HIP_OK(hipStreamCreate(&stream));
        HIP_OK(hipMalloc(&frame, 32));
        HIP_OK(hipMalloc(&rec, sizeof(mbx_param_record)));
        HIP_OK(hipMalloc(&state, 3 * sizeof(mbe_parms)));
        HIP_OK(hipMalloc(&rng, sizeof(mbx_stream_rng)));
        HIP_OK(hipMalloc(&pcmf, 160 * sizeof(float)));
        HIP_OK(hipMalloc(&pcm16, 160 * sizeof(int16_t)));
        HIP_OK(hipMalloc(&res, sizeof(mbe_process_result)));
        HIP_OK(hipMalloc(&words, 4 * sizeof(uint32_t)));
        HIP_OK(hipMalloc(&soft, MBX_IMBE_SOFT_BITS * sizeof(mbe_soft_bit)));   // the largest soft frame (184 cells)
    }
    void up(void* dst, const void* src, size_t n) { HIP_OK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, stream)); }
    void down(void* dst, const void* src, size_t n) { HIP_OK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, stream)); }
    void sync() { HIP_OK(hipStreamSynchronize(stream)); }
};

Slot& slot() {
    thread_local Slot* s = new Slot();   // lives as long as the thread's HIP context; never freed
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != g_device) {   // the host application switched devices on this thread
        HIP_OK(hipSetDevice(g_device));
    }
    return *s;
}

struct ThreadRng {
    mbx_stream_rng r;
    ThreadRng() { mbx_rng_default(&r); }
};
thread_local ThreadRng t_rng;

void must(int rc, const char* what) {
    if (rc < 0) {
        die(what);
    }
}

// ---- result bookkeeping: ref src/internal/mbe_result.h:18-121 ---------------------------------
constexpr unsigned kContext = MBE_PROCESS_FLAG_SOFT_INPUT | MBE_PROCESS_FLAG_C0_VALID | MBE_PROCESS_FLAG_C4_VALID;
constexpr unsigned kStatus = MBE_PROCESS_FLAG_TONE | MBE_PROCESS_FLAG_ERASURE | MBE_PROCESS_FLAG_REPEAT | MBE_PROCESS_FLAG_MUTE;

int validate_bits(const char* bits, size_t count) {
    if (!bits) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < count; ++i) {
        if (bits[i] != 0 && bits[i] != 1) {
            return MBE_STATUS_INVALID_BITS;
        }
    }
    return 0;
}

bool count_ok(int c) { return c >= 0 && c <= 184; }

int resolve_total_errors(const mbe_process_result* r, int* total_out) {
    if (!r) {
        *total_out = 0;
        return 0;
    }
    if ((r->flags & ~(kContext | kStatus)) != 0u) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (!count_ok(r->c0_errors) || !count_ok(r->protected_errors) || !count_ok(r->c4_errors) || !count_ok(r->total_errors)
        || r->c0_errors > 184 - r->protected_errors) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const int parts = r->c0_errors + r->protected_errors;
    if (!count_ok(parts)) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    const int total = (r->total_errors == 0 && parts != 0) ? parts : r->total_errors;
    const bool c0v = (r->flags & MBE_PROCESS_FLAG_C0_VALID) != 0u, c4v = (r->flags & MBE_PROCESS_FLAG_C4_VALID) != 0u;
    if (!((parts == 0 || total == parts) && (!c0v || total >= r->c0_errors) && (!c4v || total >= r->c4_errors))) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    *total_out = total;
    return 0;
}

bool valid_L(int L) { return L >= 1 && L <= 56; }

// parameter bits + error context -> the record the stream kernels consume
mbx_param_record make_record(const char* bits, int nbits, const mbe_process_result* r, int total) {
    mbx_param_record rec;
    memset(&rec, 0, sizeof(rec));
    for (int i = 0; i < nbits; ++i) {
        if (bits[i]) {
            rec.w[i >> 5] |= 0x80000000u >> (i & 31);
        }
    }
    const unsigned ctx = r->flags & kContext;
    const int c0 = (ctx & MBE_PROCESS_FLAG_C0_VALID) ? r->c0_errors : 0;
    const int c4 = (ctx & MBE_PROCESS_FLAG_C4_VALID) ? r->c4_errors : 0;
    rec.w[3] = (uint32_t)c0 | ((uint32_t)(total - c0) << 8) | ((uint32_t)c4 << 16) | (ctx << 24);
    return rec;
}

int decode_frame(int codec, const char* cells, int ncell, int nbits, char* bits_out, mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!bits_out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(cells, (size_t)ncell);
    if (rc < 0) {
        return rc;
    }
    uint8_t packed[MBX_IMBE_FRAME_BYTES];
    rc = (codec == MBX_CODEC_IMBE7200X4400)   ? mbx_pack_imbe7200x4400(cells, 1, packed)
         : (codec == MBX_CODEC_IMBE7100X4400) ? mbx_pack_imbe7100x4400(cells, 1, packed)
                                              : mbx_pack_ambe3600x2450(cells, 1, packed);
    if (rc < 0) {
        return rc;
    }
    Slot& s = slot();
    const size_t fb = (codec == MBX_CODEC_AMBE3600X2450) ? MBX_AMBE_FRAME_BYTES : MBX_IMBE_FRAME_BYTES;
    s.up(s.frame, packed, fb);
    must((codec == MBX_CODEC_IMBE7200X4400)   ? mbx_fec_imbe7200x4400(s.frame, 1, s.rec, s.stream)
         : (codec == MBX_CODEC_IMBE7100X4400) ? mbx_fec_imbe7100x4400(s.frame, 1, s.rec, s.stream)
                                              : mbx_fec_ambe3600x2450(s.frame, 1, s.rec, s.stream),
         "mbx_fec");
    mbx_param_record rec;
    s.down(&rec, s.rec, sizeof(rec));
    s.sync();
    mbe_process_result r;
    mbx_unpack_records(&rec, 1, nbits, bits_out, &r);
    if (result) {
        *result = r;
    }
    return r.total_errors;
}

// mbe_process*Dataf: ref src/imbe/imbe7200x4400.c:858-909, src/ambe/ambe3600x2450.c:851-898
int process_data(int codec, float* aout_f, short* aout_s, mbe_process_result* result, const char* bits, int nbits,
                 mbe_parms* cur, mbe_parms* prev, mbe_parms* enh) {
    mbe_process_result local;
    if (!result) {
        memset(&local, 0, sizeof(local));
        result = &local;
    }
    if ((!aout_f && !aout_s) || !cur || !prev || !enh) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int total = 0;
    int rc = resolve_total_errors(result, &total);
    if (rc < 0) {
        return rc;
    }
    rc = validate_bits(bits, (size_t)nbits);
    if (rc < 0) {
        return rc;
    }
    const mbx_param_record rec = make_record(bits, nbits, result, total);
    Slot& s = slot();
    s.up(s.rec, &rec, sizeof(rec));
    s.up(&s.state[0], cur, sizeof(mbe_parms));
    s.up(&s.state[1], prev, sizeof(mbe_parms));
    s.up(&s.state[2], enh, sizeof(mbe_parms));
    s.up(s.rng, &t_rng.r, sizeof(mbx_stream_rng));
    must(mbx_process_records(codec, 1, 1, s.rec, s.state, s.rng, aout_s ? s.pcm16 : nullptr, aout_f ? s.pcmf : nullptr, s.res,
                             s.stream),
         "mbx_process_records");
    if (aout_f) {
        s.down(aout_f, s.pcmf, 160 * sizeof(float));
    }
    if (aout_s) {
        s.down(aout_s, s.pcm16, 160 * sizeof(int16_t));
    }
    s.down(cur, &s.state[0], sizeof(mbe_parms));
    s.down(prev, &s.state[1], sizeof(mbe_parms));
    s.down(enh, &s.state[2], sizeof(mbe_parms));
    s.down(&t_rng.r, s.rng, sizeof(mbx_stream_rng));
    s.down(result, s.res, sizeof(mbe_process_result));
    s.sync();
    return result->total_errors;
}

void synth_speech(float* aout_f, short* aout_s, mbe_parms* cur, mbe_parms* prev) {
    Slot& s = slot();
    s.up(&s.state[0], cur, sizeof(mbe_parms));
    s.up(&s.state[1], prev, sizeof(mbe_parms));
    s.up(s.rng, &t_rng.r, sizeof(mbx_stream_rng));
    must(mbx_synthesize_speech(1, &s.state[0], &s.state[1], s.rng, aout_f ? s.pcmf : nullptr, aout_s ? s.pcm16 : nullptr,
                               s.stream),
         "mbx_synthesize_speech");
    if (aout_f) {
        s.down(aout_f, s.pcmf, 160 * sizeof(float));
    }
    if (aout_s) {
        s.down(aout_s, s.pcm16, 160 * sizeof(int16_t));
    }
    s.down(cur, &s.state[0], sizeof(mbe_parms));
    s.down(prev, &s.state[1], sizeof(mbe_parms));
    s.down(&t_rng.r, s.rng, sizeof(mbx_stream_rng));
    s.sync();
}

int ecc_word(int kind, uint32_t in, uint32_t* out) {
    Slot& s = slot();
    s.up(&s.words[0], &in, 4);
    must(mbx_ecc_words(kind, &s.words[0], 1, &s.words[1], reinterpret_cast<int32_t*>(&s.words[2]), s.stream), "mbx_ecc_words");
    uint32_t back[2];
    s.down(back, &s.words[1], 8);
    s.sync();
    *out = back[0];
    return (int)back[1];
}

// soft-decision code word (kind 0: 23 soft bits, kind 1: 15): one-wave launch
int ecc_soft_word(int kind, const mbe_soft_bit* in, uint32_t* out) {
    Slot& s = slot();
    const size_t width = kind == 0 ? 23 : 15;   // kind 2: Hamming with the 7100x4400 mapping
    s.up(s.soft, in, width * sizeof(mbe_soft_bit));
    must(mbx_ecc_soft_words(kind, s.soft, 1, &s.words[1], reinterpret_cast<int32_t*>(&s.words[2]), s.stream), "mbx_ecc_soft_words");
    uint32_t back[2];
    s.down(back, &s.words[1], 8);
    s.sync();
    *out = back[0];
    return (int)back[1];
}

// mbe_decode*SoftFrame: ref src/imbe/imbe7200x4400.c:746-778, src/ambe/ambe3600x2450.c:684-714
int decode_soft_frame(int codec, const mbe_soft_bit* cells, int ncell, int nbits, char* bits_out, mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
    if (!bits_out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = mbx_validate_soft_bits(cells, (size_t)ncell);
    if (rc < 0) {
        return rc;
    }
    Slot& s = slot();
    s.up(s.soft, cells, (size_t)ncell * sizeof(mbe_soft_bit));
    must(mbx_fec_soft(codec, s.soft, 1, s.rec, s.stream), "mbx_fec_soft");
    mbx_param_record rec;
    s.down(&rec, s.rec, sizeof(rec));
    s.sync();
    mbe_process_result r;
    mbx_unpack_records(&rec, 1, nbits, bits_out, &r);
    if (result) {
        *result = r;
    }
    return r.total_errors;
}

}  // namespace

extern "C" {

const char* mbe_versionString(void) { return "2.0.0+mi355x"; }

void mbe_initProcessResult(mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
}

void mbe_setThreadRngSeed(uint32_t seed) { mbx_rng_seed(&t_rng.r, seed); }

void mbe_moveMbeParms(const mbe_parms* source_mp, mbe_parms* destination_mp) {
    if (source_mp && destination_mp) {
        *destination_mp = *source_mp;
    }
}

void mbe_useLastMbeParms(mbe_parms* cur_mp, const mbe_parms* prev_mp) {
    if (cur_mp && prev_mp) {
        *cur_mp = *prev_mp;
    }
}

// defaults only, no signal processing: ref src/core/mbelib.c:367-410
void mbe_initMbeParms(mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!cur_mp || !prev_mp || !prev_mp_enhanced) {
        return;
    }
    mbe_parms p;
    memset(&p, 0, sizeof(p));
    p.w0 = (float)((4.0 * M_PI) / (134.0 + 39.5));
    p.L = (int)(0.9254 * (int)((M_PI / p.w0) + 0.25));
    p.K = 12;
    for (int l = 0; l <= 56; ++l) {
        p.Ml[l] = 1.0f;
    }
    p.localEnergy = 75000.0f;
    p.amplitudeThreshold = 20480;
    p.mutingThreshold = MBE_MUTING_THRESHOLD_IMBE;
    p.noiseSeed = -1.0f;
    *prev_mp = p;
    *cur_mp = p;
    *prev_mp_enhanced = p;
}

int mbe_requiresAdaptiveSmoothing(const mbe_parms* mp) { return mp ? ((mp->errorRate > 0.0125f) || (mp->errorCountTotal > 4)) : 0; }
int mbe_requiresMuting(const mbe_parms* mp) { return mp ? (mp->errorRate > mp->mutingThreshold) : 0; }
int mbe_isMaxFrameRepeat(const mbe_parms* mp) { return mp ? (mp->repeatCount >= MBE_MAX_FRAME_REPEATS) : 0; }

void mbe_synthesizeSilencef(float* aout_buf) {
    if (aout_buf) {
        memset(aout_buf, 0, 160 * sizeof(float));
    }
}

void mbe_synthesizeSilence(short* aout_buf) {
    if (aout_buf) {
        memset(aout_buf, 0, 160 * sizeof(short));
    }
}

// ---- ECC words ------------------------------------------------------------------------------
int mbe_checkGolayBlock(long int* block) {
    if (!block) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    uint32_t fixed;
    (void)ecc_word(0, (uint32_t)(*block) & 0x7fffffu, &fixed);
    *block = (long)(fixed >> 11);
    return 0;
}

int mbe_golay2312(const char* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(in, 23u);
    if (rc < 0) {
        return rc;
    }
    uint32_t cw = 0, fixed;
    for (int j = 22; j >= 0; --j) {
        cw = (cw << 1) | (uint32_t)(in[j] & 1);
    }
    const int errs = ecc_word(0, cw, &fixed);
    for (int j = 0; j < 23; ++j) {
        out[j] = (char)((fixed >> j) & 1u);
    }
    return errs;
}

int mbe_hamming1511(const char* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(in, 15u);
    if (rc < 0) {
        return rc;
    }
    uint32_t cw = 0, fixed;
    for (int j = 14; j >= 0; --j) {
        cw = (cw << 1) | (uint32_t)(in[j] & 1);
    }
    const int errs = ecc_word(1, cw, &fixed);
    for (int j = 0; j < 15; ++j) {
        out[j] = (char)((fixed >> j) & 1u);
    }
    return errs;
}

// ---- IMBE 7100x4400: ref src/ecc/ecc.c:422-464, src/imbe/imbe7100x4400.c:381-479, 527-592 -------
int mbe_7100x4400hamming1511(const char* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(in, 15u);
    if (rc < 0) {
        return rc;
    }
    uint32_t cw = 0, fixed;
    for (int j = 14; j >= 0; --j) {
        cw = (cw << 1) | (uint32_t)(in[j] & 1);
    }
    const int errs = ecc_word(2, cw, &fixed);
    for (int j = 0; j < 15; ++j) {
        out[j] = (char)((fixed >> j) & 1u);
    }
    return errs;
}

int mbe_decodeImbe7100x4400Frame(const char imbe_fr[7][24], char imbe_d[88], mbe_process_result* result) {
    return decode_frame(MBX_CODEC_IMBE7100X4400, reinterpret_cast<const char*>(imbe_fr), 168, 88, imbe_d, result);
}

int mbe_processImbe7100x4400Framef(float* aout_buf, mbe_process_result* result, const char imbe_fr[7][24], char imbe_d[88],
                                   mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7100x4400Frame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Dataf(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processImbe7100x4400Frame(short* aout_buf, mbe_process_result* result, const char imbe_fr[7][24], char imbe_d[88],
                                  mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7100x4400Frame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Data(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

// ---- tones and the status trace: ref src/core/mbelib.c:68-104 (format documented in mbelib.h:195-202), :745-856 ----
void mbe_formatProcessResult(char* str, size_t size, const mbe_process_result* result) {
    if (!str || size == 0u) {
        return;
    }
    size_t pos = 0u;
    const int total = (result && result->total_errors > 0) ? result->total_errors : 0;
    while (pos + 1u < size && (int)pos < total) {
        str[pos++] = '=';
    }
    if (result) {
        const unsigned order[4] = {MBE_PROCESS_FLAG_ERASURE, MBE_PROCESS_FLAG_TONE, MBE_PROCESS_FLAG_REPEAT, MBE_PROCESS_FLAG_MUTE};
        const char mark[4] = {'E', 'T', 'R', 'M'};
        for (int i = 0; i < 4 && pos + 1u < size; ++i) {
            if (result->flags & order[i]) {
                str[pos++] = mark[i];
            }
        }
    }
    str[pos] = '\0';
}

static void tone_call(float* aout_buf, const char* ambe_d, mbe_parms* cur_mp, const int* dstar_id) {
    if (!aout_buf) {
        return;
    }
    memset(aout_buf, 0, 160 * sizeof(float));
    if (!cur_mp) {
        return;
    }
    Slot& s = slot();
    int32_t id = 0;
    if (dstar_id) {
        id = *dstar_id;
        s.up(&s.words[0], &id, sizeof(id));
    } else {
        if (validate_bits(ambe_d, 49u) < 0) {
            return;
        }
        mbe_process_result none;
        memset(&none, 0, sizeof(none));
        const mbx_param_record rec = make_record(ambe_d, 49, &none, 0);
        s.up(s.rec, &rec, sizeof(rec));
    }
    s.up(&s.state[0], cur_mp, sizeof(mbe_parms));
    must(mbx_synthesize_tone(1, dstar_id ? nullptr : s.rec, dstar_id ? reinterpret_cast<const int32_t*>(&s.words[0]) : nullptr,
                             &s.state[0], s.pcmf, nullptr, s.stream),
         "mbx_synthesize_tone");
    s.down(aout_buf, s.pcmf, 160 * sizeof(float));
    s.down(cur_mp, &s.state[0], sizeof(mbe_parms));
    s.sync();
}

void mbe_synthesizeTonef(float* aout_buf, const char* ambe_d, mbe_parms* cur_mp) { tone_call(aout_buf, ambe_d, cur_mp, nullptr); }

void mbe_synthesizeTonefdstar(float* aout_buf, const char* ambe_d, mbe_parms* cur_mp, int ID1) {
    (void)ambe_d;
    tone_call(aout_buf, nullptr, cur_mp, &ID1);
}

// ---- soft-decision helpers: ref src/core/mbelib.c:107-158, src/ecc/ecc.c:303-357, 410-413 -----
mbe_soft_bit mbe_softBitFromHard(int bit, uint8_t reliability) {
    mbe_soft_bit s;
    s.bit = (uint8_t)(bit ? 1u : 0u);
    s.reliability = reliability;
    return s;
}

mbe_soft_bit mbe_softBitFromLlr(int16_t llr) {
    mbe_soft_bit s;
    (void)mbx_soft_bits_from_llr(&llr, &s, 1);
    return s;
}

int mbe_softBitsFromHard(const char* bits, mbe_soft_bit* soft, size_t count, uint8_t reliability) {
    if (!soft) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = validate_bits(bits, count);
    if (rc < 0) {
        return rc;
    }
    return mbx_soft_bits_from_hard(bits, soft, count, reliability);
}

int mbe_softBitsFromLlr(const int16_t* llr, mbe_soft_bit* soft, size_t count) { return mbx_soft_bits_from_llr(llr, soft, count); }

int mbe_golay2312Soft(const mbe_soft_bit* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = mbx_validate_soft_bits(in, 23u);
    if (rc < 0) {
        return rc;
    }
    uint32_t w;
    const int diffs = ecc_soft_word(0, in, &w);
    for (int j = 0; j < 23; ++j) {
        out[j] = (char)((w >> j) & 1u);
    }
    return diffs;
}

int mbe_hamming1511Soft(const mbe_soft_bit* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = mbx_validate_soft_bits(in, 15u);
    if (rc < 0) {
        return rc;
    }
    uint32_t w;
    const int diffs = ecc_soft_word(1, in, &w);
    for (int j = 0; j < 15; ++j) {
        out[j] = (char)((w >> j) & 1u);
    }
    return diffs;
}

int mbe_decodeImbe7200x4400SoftFrame(const mbe_soft_bit imbe_fr[8][23], char imbe_d[88], mbe_process_result* result) {
    return decode_soft_frame(MBX_CODEC_IMBE7200X4400, reinterpret_cast<const mbe_soft_bit*>(imbe_fr), 184, 88, imbe_d, result);
}

int mbe_decodeAmbe3600x2450SoftFrame(const mbe_soft_bit ambe_fr[4][24], char ambe_d[49], mbe_process_result* result) {
    return decode_soft_frame(MBX_CODEC_AMBE3600X2450, reinterpret_cast<const mbe_soft_bit*>(ambe_fr), 96, 49, ambe_d, result);
}

// ---- frame decode ---------------------------------------------------------------------------
int mbe_decodeImbe7200x4400Frame(const char imbe_fr[8][23], char imbe_d[88], mbe_process_result* result) {
    return decode_frame(MBX_CODEC_IMBE7200X4400, reinterpret_cast<const char*>(imbe_fr), 184, 88, imbe_d, result);
}

int mbe_decodeAmbe3600x2450Frame(const char ambe_fr[4][24], char ambe_d[49], mbe_process_result* result) {
    return decode_frame(MBX_CODEC_AMBE3600X2450, reinterpret_cast<const char*>(ambe_fr), 96, 49, ambe_d, result);
}

// ---- parameters -> PCM ------------------------------------------------------------------------
int mbe_processImbe4400Dataf(float* aout_buf, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    return process_data(MBX_CODEC_IMBE7200X4400, aout_buf, nullptr, result, imbe_d, 88, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processImbe4400Data(short* aout_buf, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur_mp,
                            mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    return process_data(MBX_CODEC_IMBE7200X4400, nullptr, aout_buf, result, imbe_d, 88, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe2450Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    return process_data(MBX_CODEC_AMBE3600X2450, aout_buf, nullptr, result, ambe_d, 49, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe2450Data(short* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                            mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    return process_data(MBX_CODEC_AMBE3600X2450, nullptr, aout_buf, result, ambe_d, 49, cur_mp, prev_mp, prev_mp_enhanced);
}

// ---- frames -> PCM: ref src/imbe/imbe7200x4400.c:935-1001, src/ambe/ambe3600x2450.c:924-990 ------
int mbe_processImbe7200x4400Framef(float* aout_buf, mbe_process_result* result, const char imbe_fr[8][23], char imbe_d[88],
                                   mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7200x4400Frame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Dataf(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processImbe7200x4400Frame(short* aout_buf, mbe_process_result* result, const char imbe_fr[8][23], char imbe_d[88],
                                  mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7200x4400Frame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Data(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2450Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24], char ambe_d[49],
                                   mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2450Frame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2450Dataf(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2450Frame(short* aout_buf, mbe_process_result* result, const char ambe_fr[4][24], char ambe_d[49],
                                  mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2450Frame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2450Data(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

// ---- synthesis pieces --------------------------------------------------------------------------
void mbe_synthesizeSpeechf(float* aout_buf, mbe_parms* cur_mp, mbe_parms* prev_mp) {
    if (!aout_buf) {
        return;
    }
    if (!cur_mp || !prev_mp || !valid_L(cur_mp->L) || !valid_L(prev_mp->L)) {
        mbe_synthesizeSilencef(aout_buf);
        return;
    }
    synth_speech(aout_buf, nullptr, cur_mp, prev_mp);
}

void mbe_synthesizeSpeech(short* aout_buf, mbe_parms* cur_mp, mbe_parms* prev_mp) {
    if (!aout_buf) {
        return;
    }
    if (!cur_mp || !prev_mp || !valid_L(cur_mp->L) || !valid_L(prev_mp->L)) {
        mbe_synthesizeSilence(aout_buf);
        return;
    }
    synth_speech(nullptr, aout_buf, cur_mp, prev_mp);
}

void mbe_spectralAmpEnhance(mbe_parms* cur_mp) {
    if (!cur_mp || !valid_L(cur_mp->L)) {
        return;
    }
    Slot& s = slot();
    s.up(&s.state[0], cur_mp, sizeof(mbe_parms));
    must(mbx_spectral_amp_enhance(1, &s.state[0], s.stream), "mbx_spectral_amp_enhance");
    s.down(cur_mp, &s.state[0], sizeof(mbe_parms));
    s.sync();
}

void mbe_applyAdaptiveSmoothing(mbe_parms* cur_mp, const mbe_parms* prev_mp) {
    if (!cur_mp || !prev_mp || !valid_L(cur_mp->L) || !valid_L(prev_mp->L)) {
        return;
    }
    Slot& s = slot();
    s.up(&s.state[0], cur_mp, sizeof(mbe_parms));
    s.up(&s.state[1], prev_mp, sizeof(mbe_parms));
    must(mbx_adaptive_smoothing(1, &s.state[0], &s.state[1], s.stream), "mbx_adaptive_smoothing");
    s.down(cur_mp, &s.state[0], sizeof(mbe_parms));
    s.sync();
}

void mbe_floattoshort(const float* float_buf, short* aout_buf) {
    if (!float_buf || !aout_buf) {
        return;
    }
    Slot& s = slot();
    s.up(s.pcmf, float_buf, 160 * sizeof(float));
    must(mbx_floattoshort(s.pcmf, s.pcm16, 1, s.stream), "mbx_floattoshort");
    s.down(aout_buf, s.pcm16, 160 * sizeof(int16_t));
    s.sync();
}

void mbe_synthesizeComfortNoisef(float* aout_buf) {
    if (!aout_buf) {
        return;
    }
    Slot& s = slot();
    s.up(s.rng, &t_rng.r, sizeof(mbx_stream_rng));
    must(mbx_comfort_noise(1, s.rng, s.pcmf, nullptr, s.stream), "mbx_comfort_noise");
    s.down(aout_buf, s.pcmf, 160 * sizeof(float));
    s.down(&t_rng.r, s.rng, sizeof(mbx_stream_rng));
    s.sync();
}

void mbe_synthesizeComfortNoise(short* aout_buf) {
    if (!aout_buf) {
        return;
    }
    Slot& s = slot();
    s.up(s.rng, &t_rng.r, sizeof(mbx_stream_rng));
    must(mbx_comfort_noise(1, s.rng, nullptr, s.pcm16, s.stream), "mbx_comfort_noise");
    s.down(aout_buf, s.pcm16, 160 * sizeof(int16_t));
    s.down(&t_rng.r, s.rng, sizeof(mbx_stream_rng));
    s.sync();
}

// ---- soft IMBE 7100x4400: ref src/ecc/ecc.c:466-469, src/imbe/imbe7100x4400.c:481-525, 542-576 -------------
int mbe_7100x4400hamming1511Soft(const mbe_soft_bit* in, char* out) {
    if (!out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    int rc = mbx_validate_soft_bits(in, 15u);
    if (rc < 0) {
        return rc;
    }
    uint32_t w;
    const int diffs = ecc_soft_word(2, in, &w);
    for (int j = 0; j < 15; ++j) {
        out[j] = (char)((w >> j) & 1u);
    }
    return diffs;
}

int mbe_decodeImbe7100x4400SoftFrame(const mbe_soft_bit imbe_fr[7][24], char imbe_d[88], mbe_process_result* result) {
    return decode_soft_frame(MBX_CODEC_IMBE7100X4400, reinterpret_cast<const mbe_soft_bit*>(imbe_fr), 168, 88, imbe_d, result);
}

int mbe_processImbe7100x4400SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[7][24],
                                       char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7100x4400SoftFrame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Dataf(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processImbe7100x4400SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[7][24],
                                      char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7100x4400SoftFrame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Data(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

// ---- AMBE 3600x2400 (D-STAR): ref src/ambe/ambe3600x2400.c:563-627 (frame decode = the AMBE+2 one), :733-852 --
int mbe_decodeAmbe3600x2400Frame(const char ambe_fr[4][24], char ambe_d[49], mbe_process_result* result) {
    return decode_frame(MBX_CODEC_AMBE3600X2450, reinterpret_cast<const char*>(ambe_fr), 96, 49, ambe_d, result);
}

int mbe_decodeAmbe3600x2400SoftFrame(const mbe_soft_bit ambe_fr[4][24], char ambe_d[49], mbe_process_result* result) {
    return decode_soft_frame(MBX_CODEC_AMBE3600X2450, reinterpret_cast<const mbe_soft_bit*>(ambe_fr), 96, 49, ambe_d, result);
}

int mbe_processAmbe2400Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    return process_data(MBX_CODEC_AMBE3600X2400, aout_buf, nullptr, result, ambe_d, 49, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe2400Data(short* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                            mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    return process_data(MBX_CODEC_AMBE3600X2400, nullptr, aout_buf, result, ambe_d, 49, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2400Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24], char ambe_d[49],
                                   mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2400Frame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2400Dataf(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2400Frame(short* aout_buf, mbe_process_result* result, const char ambe_fr[4][24], char ambe_d[49],
                                  mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2400Frame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2400Data(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2400SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                       char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2400SoftFrame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2400Dataf(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2400SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                      char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2400SoftFrame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2400Data(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

// ---- soft frames -> PCM: ref src/imbe/imbe7200x4400.c:950-980, src/ambe/ambe3600x2450.c:939-969 --------
int mbe_processImbe7200x4400SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[8][23],
                                       char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7200x4400SoftFrame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Dataf(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processImbe7200x4400SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit imbe_fr[8][23],
                                      char imbe_d[88], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeImbe7200x4400SoftFrame(imbe_fr, imbe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processImbe4400Data(aout_buf, result, imbe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2450SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                       char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2450SoftFrame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2450Dataf(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

int mbe_processAmbe3600x2450SoftFrame(short* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                      char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_process_result local;
    if (!result) {
        result = &local;
    }
    const int rc = mbe_decodeAmbe3600x2450SoftFrame(ambe_fr, ambe_d, result);
    if (rc < 0) {
        return rc;
    }
    return mbe_processAmbe2450Data(aout_buf, result, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

}  // extern "C"
