// valu_issue.hip -- what one instruction of each class costs a gfx950 SIMD, as a function of the waves resident on it.
//
// The stream kernels are priced against an issue model; this program measures that model's
// constants on the box instead of assuming them: for each instruction class a wave runs a long unrolled stream of
// INDEPENDENT instructions (eight accumulators), W waves per SIMD on every SIMD of the chip, and the elapsed shader cycles
// (s_memtime) of the slowest wave of a SIMD divided by the W x N instructions that SIMD retired is the cost per
// wave-instruction in SIMD cycles.  W = 1 shows what one wave alone sustains, W >= 2 the SIMD's throughput.
// Two rows are kept as a warning, not as a cost: v_cndmask_b32_e32 with its mask in VCC measures 12-19 cycles here while the
// e64 form of the same instruction measures 3.1 -- but rewriting all 868 v_cndmask_b32_e32 of mbx_stream.hip to e64 in the
// assembly changed no kernel time (2.762 vs 2.769 ms, 0.2261 vs 0.2268 ms), so that figure is a property of this loop.
// Under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` the same
// launches calibrate those counters on a known instruction count (tools/sq_profile.sh does that).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_issue tools/valu_issue.hip && tools/bin/valu_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

constexpr int kIters = 128;   // loop trips
constexpr int kBody = 64;     // instructions per trip (8 accumulators x 8)

#define R8(s) s s s s s s s s
// eight independent instructions, one per accumulator
#define OPS8(op, tail)                                                                                                 \
    op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" \
       op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"

enum Op { kFma, kPkFma, kPkMul, kCos, kFma64, kAddDpp, kMulLoU32, kMov, kReadlane, kDsReadB128, kBankLoop, kAddE32, kMulE32, kFmacE32, kCndmaskE32, kMulSgpr, kFmaSgpr, kPkFmaSgpr, kCndmaskE64, kCndmaskVccE64, kCndmaskVccFresh, kCmpThenCndmask, kMulInline, kMulLiteral, kAddU32Inline, kLshlInline, kAndVgpr, kCmpVcc, kCmpE64, kCvtI2F, kMulOtherDst, kMixPkMul, kSubrevInline, kMaxInline, kNumOps };
static const char* kNames[kNumOps] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_cos_f32", "v_fma_f64", "v_add_f32 dpp",
                                      "v_mul_lo_u32", "v_mov_b32", "v_readlane_b32", "ds_read_b128 (broadcast)",
                                      "bank loop body (8 pk + 2 ds_read_b128)", "v_add_f32_e32 (VOP2, 4 B)", "v_mul_f32_e32 (VOP2, 4 B)",
                                      "v_fmac_f32_e32 (VOP2, 4 B)", "v_cndmask_b32_e32 (VOP2, 4 B, mask in VCC)",
                                      "v_mul_f32_e32 with an SGPR source", "v_fma_f32 with an SGPR source", "v_pk_fma_f32 with an SGPR-pair source",
                                      "v_cndmask_b32_e64 (mask in an SGPR pair)", "v_cndmask_b32_e64 (mask in VCC)",
                                      "v_cndmask_b32_e32 (VCC written by s_mov once per 8)", "v_cmp_gt_f32_e32 + 7 x v_cndmask_b32_e32 (per 8)",
                                      "v_mul_f32_e32 with an inline constant (2.0)", "v_mul_f32_e32 with a 32-bit literal (8 B)",
                                      "v_add_u32_e32 with an inline constant", "v_lshlrev_b32_e32 by an inline constant", "v_and_b32_e32 (two VGPRs)",
                                      "v_cmp_gt_f32_e32 (to VCC)", "v_cmp_gt_f32_e64 (to an SGPR pair)", "v_cvt_f32_i32_e32 (one source)",
                                      "v_mul_f32_e32, destination distinct from both sources", "4 x v_pk_fma_f32 + 4 x v_mul_f32_e32 interleaved (per instruction)",
                                      "v_sub_f32_e32 with an inline constant (1.0 - x)", "v_max_f32_e32 with an inline constant (0)"};
// instructions of the measured class per loop trip
static const int kPerTrip[kNumOps] = {kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, 8 * 8, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody, kBody};

template <int kOp>
__global__ void __launch_bounds__(1024) issue_kernel(unsigned long long* cycles, float* sink, float seed) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    lds[tid] = seed + (float)tid;
    __syncthreads();
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    int i0 = tid, i1 = tid + 1, i2 = tid + 2, i3 = tid + 3, i4 = tid + 4, i5 = tid + 5, i6 = tid + 6, i7 = tid + 7;
    const float k = 0.999f;
    const v2f pk = {0.999f, 1.001f};
    const double dk = 0.999;
    int sacc = 0;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < kIters; ++it) {
        if constexpr (kOp == kFma) {
            asm volatile(R8(OPS8("v_fma_f32", ", %8, %8"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k));
        } else if constexpr (kOp == kPkFma) {
            asm volatile(R8(OPS8("v_pk_fma_f32", ", %8, %8"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         : "v"(pk));
        } else if constexpr (kOp == kPkMul) {
            asm volatile(R8(OPS8("v_pk_mul_f32", ", %8"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         : "v"(pk));
        } else if constexpr (kOp == kCos) {
            asm volatile(R8(OPS8("v_cos_f32", ""))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (kOp == kFma64) {
            asm volatile(R8(OPS8("v_fma_f64", ", %8, %8"))
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                         : "v"(dk));
        } else if constexpr (kOp == kAddDpp) {
            asm volatile(R8(OPS8("v_add_f32_dpp", ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k));
        } else if constexpr (kOp == kMulLoU32) {
            asm volatile(R8(OPS8("v_mul_lo_u32", ", %8"))
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "v"(i0 | 3));
        } else if constexpr (kOp == kMov) {
            asm volatile(R8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                            "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k));
        } else if constexpr (kOp == kAddE32) {
            asm volatile(R8(OPS8("v_add_f32_e32", ", %8"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k));
        } else if constexpr (kOp == kMulE32) {
            asm volatile(R8(OPS8("v_mul_f32_e32", ", %8"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k));
        } else if constexpr (kOp == kFmacE32) {
            asm volatile(R8("v_fmac_f32_e32 %0, %8, %8\n v_fmac_f32_e32 %1, %8, %8\n v_fmac_f32_e32 %2, %8, %8\n v_fmac_f32_e32 %3, %8, %8\n"
                            "v_fmac_f32_e32 %4, %8, %8\n v_fmac_f32_e32 %5, %8, %8\n v_fmac_f32_e32 %6, %8, %8\n v_fmac_f32_e32 %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k));
        } else if constexpr (kOp == kCndmaskE32) {
            asm volatile(R8(OPS8("v_cndmask_b32_e32", ", %8, vcc"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k)
                         : "vcc");
        } else if constexpr (kOp == kMulSgpr) {
            asm volatile(R8("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                            "v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "s"(k));
        } else if constexpr (kOp == kFmaSgpr) {
            asm volatile(R8(OPS8("v_fma_f32", ", %8, %9"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "s"(k), "v"(seed));
        } else if constexpr (kOp == kPkFmaSgpr) {
            asm volatile(R8(OPS8("v_pk_fma_f32", ", %8, %9"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                         : "s"(pk), "v"(pk));
        } else if constexpr (kOp == kCndmaskE64) {
            const unsigned long long m = 0x5555aaaa3333ccccULL ^ (unsigned long long)it;
            asm volatile(R8(OPS8("v_cndmask_b32_e64", ", %8, %9"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k), "s"(m));
        } else if constexpr (kOp == kCndmaskVccE64) {
            asm volatile(R8(OPS8("v_cndmask_b32_e64", ", %8, vcc"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k)
                         : "vcc");
        } else if constexpr (kOp == kCndmaskVccFresh) {
            asm volatile(R8("s_mov_b64 vcc, %9\n s_nop 4\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n"
                            "v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n"
                            "v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k), "s"(0x5555aaaa3333ccccULL)
                         : "vcc");
        } else if constexpr (kOp == kCmpThenCndmask) {
            asm volatile(R8("v_cmp_gt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n"
                            "v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n"
                            "v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n v_cndmask_b32_e32 %0, %0, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k)
                         : "vcc");
        } else if constexpr (kOp == kMulInline) {
            asm volatile(R8("v_mul_f32_e32 %0, 2.0, %0\n v_mul_f32_e32 %1, 2.0, %1\n v_mul_f32_e32 %2, 0.5, %2\n v_mul_f32_e32 %3, 0.5, %3\n"
                            "v_mul_f32_e32 %4, 2.0, %4\n v_mul_f32_e32 %5, 2.0, %5\n v_mul_f32_e32 %6, 0.5, %6\n v_mul_f32_e32 %7, 0.5, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (kOp == kMulLiteral) {
            asm volatile(R8("v_mul_f32_e32 %0, 0x3f7fbe77, %0\n v_mul_f32_e32 %1, 0x3f7fbe77, %1\n v_mul_f32_e32 %2, 0x3f7fbe77, %2\n v_mul_f32_e32 %3, 0x3f7fbe77, %3\n"
                            "v_mul_f32_e32 %4, 0x3f7fbe77, %4\n v_mul_f32_e32 %5, 0x3f7fbe77, %5\n v_mul_f32_e32 %6, 0x3f7fbe77, %6\n v_mul_f32_e32 %7, 0x3f7fbe77, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (kOp == kAddU32Inline) {
            asm volatile(R8("v_add_u32_e32 %0, 3, %0\n v_add_u32_e32 %1, 3, %1\n v_add_u32_e32 %2, 3, %2\n v_add_u32_e32 %3, 3, %3\n"
                            "v_add_u32_e32 %4, 3, %4\n v_add_u32_e32 %5, 3, %5\n v_add_u32_e32 %6, 3, %6\n v_add_u32_e32 %7, 3, %7\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if constexpr (kOp == kLshlInline) {
            asm volatile(R8("v_lshlrev_b32_e32 %0, 1, %0\n v_lshlrev_b32_e32 %1, 1, %1\n v_lshlrev_b32_e32 %2, 1, %2\n v_lshlrev_b32_e32 %3, 1, %3\n"
                            "v_lshlrev_b32_e32 %4, 1, %4\n v_lshlrev_b32_e32 %5, 1, %5\n v_lshlrev_b32_e32 %6, 1, %6\n v_lshlrev_b32_e32 %7, 1, %7\n")
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
        } else if constexpr (kOp == kAndVgpr) {
            asm volatile(R8(OPS8("v_and_b32_e32", ", %8"))
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "v"(~tid));
        } else if constexpr (kOp == kCmpVcc) {
            asm volatile(R8("v_cmp_gt_f32_e32 vcc, %0, %8\n v_cmp_gt_f32_e32 vcc, %1, %8\n v_cmp_gt_f32_e32 vcc, %2, %8\n v_cmp_gt_f32_e32 vcc, %3, %8\n"
                            "v_cmp_gt_f32_e32 vcc, %4, %8\n v_cmp_gt_f32_e32 vcc, %5, %8\n v_cmp_gt_f32_e32 vcc, %6, %8\n v_cmp_gt_f32_e32 vcc, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k)
                         : "vcc");
        } else if constexpr (kOp == kCmpE64) {
            unsigned long long m0, m1, m2, m3;
            asm volatile(R8("v_cmp_gt_f32_e64 %8, %0, %12\n v_cmp_gt_f32_e64 %9, %1, %12\n v_cmp_gt_f32_e64 %10, %2, %12\n v_cmp_gt_f32_e64 %11, %3, %12\n"
                            "v_cmp_gt_f32_e64 %8, %4, %12\n v_cmp_gt_f32_e64 %9, %5, %12\n v_cmp_gt_f32_e64 %10, %6, %12\n v_cmp_gt_f32_e64 %11, %7, %12\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                         : "v"(k));
            sacc += (int)(m0 ^ m1 ^ m2 ^ m3);
        } else if constexpr (kOp == kCvtI2F) {
            asm volatile(R8(OPS8("v_cvt_f32_i32_e32", ""))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (kOp == kMulOtherDst) {
            asm volatile(R8("v_mul_f32_e32 %0, %8, %9\n v_mul_f32_e32 %1, %8, %9\n v_mul_f32_e32 %2, %8, %9\n v_mul_f32_e32 %3, %8, %9\n"
                            "v_mul_f32_e32 %4, %8, %9\n v_mul_f32_e32 %5, %8, %9\n v_mul_f32_e32 %6, %8, %9\n v_mul_f32_e32 %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(k), "v"(seed));
        } else if constexpr (kOp == kMixPkMul) {
            asm volatile(R8("v_pk_fma_f32 %0, %0, %8, %8\n v_mul_f32_e32 %4, %4, %9\n v_pk_fma_f32 %1, %1, %8, %8\n v_mul_f32_e32 %5, %5, %9\n"
                            "v_pk_fma_f32 %2, %2, %8, %8\n v_mul_f32_e32 %6, %6, %9\n v_pk_fma_f32 %3, %3, %8, %8\n v_mul_f32_e32 %7, %7, %9\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(pk), "v"(k));
        } else if constexpr (kOp == kSubrevInline) {
            asm volatile(R8("v_sub_f32_e32 %0, 1.0, %0\n v_sub_f32_e32 %1, 1.0, %1\n v_sub_f32_e32 %2, 1.0, %2\n v_sub_f32_e32 %3, 1.0, %3\n"
                            "v_sub_f32_e32 %4, 1.0, %4\n v_sub_f32_e32 %5, 1.0, %5\n v_sub_f32_e32 %6, 1.0, %6\n v_sub_f32_e32 %7, 1.0, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (kOp == kMaxInline) {
            asm volatile(R8("v_max_f32_e32 %0, 0, %0\n v_max_f32_e32 %1, 0, %1\n v_max_f32_e32 %2, 0, %2\n v_max_f32_e32 %3, 0, %3\n"
                            "v_max_f32_e32 %4, 0, %4\n v_max_f32_e32 %5, 0, %5\n v_max_f32_e32 %6, 0, %6\n v_max_f32_e32 %7, 0, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (kOp == kReadlane) {
            int s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile(R8("v_readlane_b32 %0, %8, 1\n v_readlane_b32 %1, %9, 2\n v_readlane_b32 %2, %10, 3\n v_readlane_b32 %3, %11, 4\n"
                            "v_readlane_b32 %4, %12, 5\n v_readlane_b32 %5, %13, 6\n v_readlane_b32 %6, %14, 7\n v_readlane_b32 %7, %15, 8\n")
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
                         : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));
            sacc += s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
        } else if constexpr (kOp == kDsReadB128) {
            // wave-uniform address (an LDS broadcast, the voiced bank's coefficient reads); 8 outstanding, then a wait
            float4 q0, q1, q2, q3, q4, q5, q6, q7;
            const int base = (it & 7) * 16;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                             "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
                             : "v"(base)
                             : "memory");
                a0 += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
            }
        } else if constexpr (kOp == kBankLoop) {
            // the voiced bank's harmonic step as the compiler emits it: 2 ds_read_b128 (wave-uniform address) feeding 4
            // accumulating v_pk_fma_f32, then the rotation (v_pk_mul, v_pk_fma, v_pk_mul, v_pk_fma) -- a dependent chain
            const int base = (it & 7) * 32;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float4 ca, cd;
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(ca), "=&v"(cd) : "v"(base + r * 32) : "memory");
                const v2f ax = {ca.x, ca.y}, az = {ca.z, ca.w}, bx = {cd.x, cd.y}, bz = {cd.z, cd.w};
                v2f u1, u2;
                asm volatile("v_pk_fma_f32 %0, %4, %8, %0\n v_pk_fma_f32 %1, %5, %9, %1\n v_pk_fma_f32 %2, %4, %10, %2\n v_pk_fma_f32 %3, %5, %11, %3\n"
                             "v_pk_mul_f32 %6, %5, %12\n v_pk_mul_f32 %7, %4, %12\n"
                             "v_pk_fma_f32 %4, %4, %12, %6 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %5, %5, %12, %7\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "=&v"(u1), "=&v"(u2)
                             : "v"(ax), "v"(az), "v"(bx), "v"(bz), "v"(pk));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const int wave = (blockIdx.x * blockDim.x + tid) >> 6;
    if ((tid & 63) == 0) {
        cycles[wave] = t1 - t0;
    }
    sink[blockIdx.x * blockDim.x + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y
                                          + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7) + (float)sacc;
}

template <int kOp>
static void run(int cus, unsigned long long* d_cycles, float* d_sink, int only_w) {
    const int Ws[] = {1, 2, 4};
    for (int W : Ws) {
        if (only_w && W != only_w) {
            continue;
        }
        // one workgroup of 4 W waves per CU (W per SIMD); 100 KB of LDS keeps a second workgroup off the CU
        const int threads = 256 * W;
        const size_t lds = 100 * 1024;
        CHECK(hipFuncSetAttribute((const void*)issue_kernel<kOp>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(issue_kernel<kOp>, dim3(cus), dim3(threads), lds, 0, d_cycles, d_sink, 1.0f);
        }
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(cus * 4 * W);
        CHECK(hipMemcpy(h.data(), d_cycles, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2], mx = (double)h.back();
        const double n = (double)kIters * kPerTrip[kOp];
        printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"insts_per_wave\": %.0f, \"median_wave_cycles\": %.0f, \"max_wave_cycles\": %.0f, "
               "\"simd_cycles_per_wave_inst\": %.3f}\n",
               kNames[kOp], W, n, med, mx, med / (n * W));
    }
}

int main(int argc, char** argv) {
    const int only_w = argc > 1 ? atoi(argv[1]) : 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_cycles;
    float* d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * cus * 16));
    CHECK(hipMalloc(&d_sink, sizeof(float) * cus * 1024));
    printf("{\"device\": \"%s\", \"cus\": %d, \"insts_note\": \"independent instructions, 8 accumulators; W waves on every SIMD of the chip\"}\n",
           prop.gcnArchName, cus);
    run<kFma>(cus, d_cycles, d_sink, only_w);
    run<kPkFma>(cus, d_cycles, d_sink, only_w);
    run<kPkMul>(cus, d_cycles, d_sink, only_w);
    run<kCos>(cus, d_cycles, d_sink, only_w);
    run<kFma64>(cus, d_cycles, d_sink, only_w);
    run<kAddDpp>(cus, d_cycles, d_sink, only_w);
    run<kMulLoU32>(cus, d_cycles, d_sink, only_w);
    run<kMov>(cus, d_cycles, d_sink, only_w);
    run<kReadlane>(cus, d_cycles, d_sink, only_w);
    run<kDsReadB128>(cus, d_cycles, d_sink, only_w);
    run<kBankLoop>(cus, d_cycles, d_sink, only_w);
    run<kAddE32>(cus, d_cycles, d_sink, only_w);
    run<kMulE32>(cus, d_cycles, d_sink, only_w);
    run<kFmacE32>(cus, d_cycles, d_sink, only_w);
    run<kCndmaskE32>(cus, d_cycles, d_sink, only_w);
    run<kMulSgpr>(cus, d_cycles, d_sink, only_w);
    run<kFmaSgpr>(cus, d_cycles, d_sink, only_w);
    run<kPkFmaSgpr>(cus, d_cycles, d_sink, only_w);
    run<kCndmaskE64>(cus, d_cycles, d_sink, only_w);
    run<kCndmaskVccE64>(cus, d_cycles, d_sink, only_w);
    run<kCndmaskVccFresh>(cus, d_cycles, d_sink, only_w);
    run<kCmpThenCndmask>(cus, d_cycles, d_sink, only_w);
    run<kMulInline>(cus, d_cycles, d_sink, only_w);
    run<kMulLiteral>(cus, d_cycles, d_sink, only_w);
    run<kAddU32Inline>(cus, d_cycles, d_sink, only_w);
    run<kLshlInline>(cus, d_cycles, d_sink, only_w);
    run<kAndVgpr>(cus, d_cycles, d_sink, only_w);
    run<kCmpVcc>(cus, d_cycles, d_sink, only_w);
    run<kCmpE64>(cus, d_cycles, d_sink, only_w);
    run<kCvtI2F>(cus, d_cycles, d_sink, only_w);
    run<kMulOtherDst>(cus, d_cycles, d_sink, only_w);
    run<kMixPkMul>(cus, d_cycles, d_sink, only_w);
    run<kSubrevInline>(cus, d_cycles, d_sink, only_w);
    run<kMaxInline>(cus, d_cycles, d_sink, only_w);
    return 0;
}
