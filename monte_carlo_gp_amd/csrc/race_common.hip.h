// race_common.hip.h -- pieces shared by the race kernels: the parameter block, the Philox4x32-10
// block function, the inverse-normal transform and the tyre-compound rule.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcgp {

constexpr int kMaxCars = 32;
constexpr int kNormalRows = 448;
constexpr int kNormal53Rows = 784;             // binary64 inverse-normal table of the reference-width build (normal53_table.h)
constexpr int kNormal53RowBytes = 64;          // 8 coefficients
constexpr int kCompStride = 8;

// Philox counter word 3 = purpose << 16 | index   (the oracle uses the same map)
constexpr uint32_t kPurposeGrid = 0u << 16;
constexpr uint32_t kPurposeEvent = 1u << 16;
constexpr uint32_t kPurposeCar = 2u << 16;
constexpr uint32_t kPurposeOvt = 3u << 16;
constexpr uint32_t kPurposeRetire = 4u << 16;

// ---- retirements of laps >= 2 (reference :190-197), drawn once per race ----
// The reference retires a running car on a lap if a fresh uniform is below the driver's per-lap probability p; the lap of
// the first success is geometric.  Here it comes from ONE word w per driver and race (counter {sim, 0, RETIRE | d >> 2},
// word d & 3): with t = ceil(p 2^32) (as for every Bernoulli draw of the path) and q = 2^32 - t, the car survives lap k
// (k = 2, 3, ..) iff  w < S_k,  S_2 = q,  S_{k+1} = floor(S_k q / 2^32)  -- integer arithmetic, the same on every
// device and in the oracle.  S is non-increasing, so the retirement lap is 2 + the number of laps survived; t = 0 never
// retires, t = 2^32 (p >= 1) retires on lap 2.  P(retire on lap k | running) = 1 - S_k / S_{k-1} = p to within 2^-32 / S,
// the resolution every other draw of the path has.
__host__ __device__ inline uint32_t retire_next_threshold(uint32_t S, uint32_t q)
{
    return (uint32_t)(((uint64_t)S * (uint64_t)q) >> 32);
}
// the lap (2 .. L), or 0 for "not in this race"
__host__ __device__ inline uint32_t draw_retirement_lap(uint32_t w, uint64_t t, int L)
{
    if (t == 0ull) return 0u;
    const uint32_t q = (uint32_t)(4294967296ull - t);
    uint32_t S = q;
    for (int k = 2; k <= L; ++k) {
        if (!(w < S)) return (uint32_t)k;
        S = retire_next_threshold(S, q);
    }
    return 0u;
}

// Read-only problem description, resident in device memory, uniform across lanes.
struct KParams {
    int32_t n, total_laps, track, pop_sh, pop_mh, pad0;
    double pit_loss, overtake_delta, drs_delta, dirty_thr, dirty_pen;
    double overtake_delta_31;       // overtake_delta x 2^31 (exact): the register kernel's pace deltas carry that scale
    // u < p  <=>  w < ceil(p * 2^32) for the 32-bit uniform u = w / 2^32
    uint64_t t_red, t_sc, t_vsc, t_vsc_tire;
    double comp_deg[kCompStride], comp_delta[kCompStride];
    double base_pace[kMaxCars];
    double factor[kMaxCars];        // deg / 0.05 if deg > 0 else 1.0      reference :321
    double tire_deg[kMaxCars];      // overtake pace                        reference :514
    double variance[kMaxCars];
    uint64_t t_dnf1[kMaxCars];      // lap 1: team rate * 4.0               reference :286-287
    uint64_t t_dnf[kMaxCars];       // laps >= 2                            reference :190-194
    uint16_t opt_laps[kMaxCars * kCompStride];   // pit threshold per driver and compound  :454-462
    double grid_probs[kMaxCars * kMaxCars];      // [driver][slot], row stride n
    uint32_t normal_bits[kNormalRows * 4];
    // ---- reference-width deviates (mcgp_config.deviates = MCGP_DEVIATES_53; race_kernel_reg<N, true>) ----
    int32_t wide, pad1;
    // u < p for the 53-bit uniform u = q / 2^53  <=>  q < ceil(p 2^53)   (p 2^53 is exact in binary64)
    uint64_t t53_red, t53_sc, t53_vsc, t53_vsc_tire;
    uint64_t t53_dnf1[kMaxCars];
    // retirement chain of laps >= 2 in 64 bits: q = 2^64 - ceil(p 2^64) (0 for p >= 1; unused for p <= 0: t_dnf = 0)
    uint64_t q64_dnf[kMaxCars];
};

// LDS bytes: block-shared tables, then per-thread rows.
constexpr size_t kSharedTableBytes =
    kNormalRows * 16            // inverse-normal cubic rows (float4)
    + 4 * kMaxCars * 8          // base_pace, factor, tire_deg, variance
    + 2 * kMaxCars * 8          // t_dnf, t_dnf1
    + kMaxCars * kCompStride * 2   // opt_laps
    + 2 * kCompStride * 8       // comp_deg, comp_delta
    + kMaxCars * kMaxCars * 4;  // histogram
__host__ __device__ constexpr size_t per_thread_lds_bytes(int n) { return (size_t)n * (8 + 8 + 4 + 1 + 2); }

// a ^ b ^ key in one instruction (v_bitop3_b32, truth table 0x96), `key` wave-uniform (a round key: it goes in as the
// instruction's one scalar operand).  Two VOP2 exclusive-ors cost more issue time than one VOP3 instruction at three
// waves per SIMD (2 x 2.85 cycles against 4.3, profiles/r3_valu_peak.json): 240 pairs per 20-car lap, 2.4 % of the kernel.
__device__ __forceinline__ uint32_t xor3_key(uint32_t a, uint32_t b, uint32_t key)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(key));
    return r;
#else
    return a ^ b ^ key;
#endif
}

// Philox4x32-10.  k0, k1: the key, the same for every lane of the wave (the run's seed).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1,
                                              uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3_key((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3_key((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// One 32-bit word -> N(0,1): piecewise-cubic inverse CDF (tools/gen_normal_table.py).  The row of a deviate is named by
// its byte offset `off` from a base kNormalRowBias rows BEFORE the table (the offset is a bit field of a float, whose
// biased exponent starts at 131: rows 48 .. 495 of that base); `row_of(off)` returns the row as a float4.
// (in two steps, so that a caller with several deviates to make can fetch all their rows before evaluating any)
constexpr uint32_t kNormalRowBias = 48;
__device__ __forceinline__ void normal_prepare(uint32_t w, uint32_t &off, float &t)
{
    // float(m + 16): exponent 4 .. 31 = the octave, its four leading mantissa bits = the cell, the other 19 = the offset
    // inside the cell -- the conversion does the count-leading-zeros and the shifts (tools/gen_normal_table.py)
    const uint32_t b = __float_as_uint((float)((w & 0x7fffffffu) + 16u));
    off = (b >> 15) & 0x1ff0u;
    t = (float)(b & 0x7ffffu);                                               // cell coordinate x 2^19
}
__device__ __forceinline__ float normal_evaluate(uint32_t w, const float4 cf, float t)
{
    float z = __builtin_fmaf(cf.w, t, cf.z);
    z = __builtin_fmaf(z, t, cf.y);
    z = __builtin_fmaf(z, t, cf.x);
    // z0 = Phi^-1(tail probability) is negative (or -0): the sign bit of w flips it
    return __uint_as_float(__float_as_uint(z) ^ (w & 0x80000000u));
}
template <typename RowFn>
__device__ __forceinline__ float normal_from_u32_rows(uint32_t w, RowFn row_of)
{
    uint32_t off;
    float t;
    normal_prepare(w, off, t);
    return normal_evaluate(w, row_of(off), t);
}

// Reference-width normal: z0 = Phi^-1((q + 0.5) / 2^53) < 0 for the 52-bit tail index q, degree-7 polynomial on
// log-spaced cells, explicit binary64 fma (tools/gen_normal53_table.py; the oracle's normal53_tail is the same text).
// `tab` = the table's rows of 8 doubles in device memory.
__device__ __forceinline__ double normal53_tail(uint64_t q, const double *__restrict__ tab)
{
    const bool small = q < 16ull;
    const uint64_t qq = small ? 16ull : q;
    const int sh = 59 - __clzll((long long)qq);                  // floor(log2 qq) - 4
    const uint32_t k = (uint32_t)(qq >> sh) & 15u;
    const uint64_t r = qq & ((1ull << sh) - 1ull);
    const double t = small ? 0.0 : ((double)r + 0.5) * __hiloint2double((1023 - sh) << 20, 0);   // x 2^-sh, exact
    const uint32_t row = small ? (uint32_t)q : 16u + 16u * (uint32_t)sh + k;
    const double *c = tab + 8u * row;
    double z = c[7];
#pragma unroll
    for (int d = 6; d >= 0; --d) z = __builtin_fma(z, t, c[d]);
    return z;
}
// the draw's 52-bit tail index and the deviate: sign and the 31 magnitude bits from the word w the 32-bit mode reads,
// 21 more bits from the same word position of the companion block
__device__ __forceinline__ double normal53(uint32_t w, uint32_t companion, const double *__restrict__ tab)
{
    const uint64_t q = ((uint64_t)(w & 0x7fffffffu) << 21) | (uint64_t)(companion >> 11);
    const double z0 = normal53_tail(q, tab);
    return (w >> 31) ? -z0 : z0;
}
// The same transform in three steps, for the hot loop (several deviates: all rows named, then fetched, then evaluated).
// The row and the cell coordinate come out of the bits of double(q) -- exact, q < 2^52 -- instead of a count-leading-zeros
// and 64-bit shifts: with e = floor(log2 q) >= 4 the biased exponent and the four leading mantissa bits ARE the row,
//     row = 16 + 16 (e - 4) + k = (hi word >> 16) - kNormal53HiBias,
// the other 48 mantissa bits are r left-aligned, so X = 1 + r 2^-sh is the same mantissa moved up by four bits under the
// exponent of 1.0, and  t = (r + 0.5) 2^-sh = (X - 1) + 2^(3 - e)  (both additions exact: the result is the oracle's
// ((double)r + 0.5) * 2^-sh, bit for bit).  Tail indices below 16 (rows 0 .. 15, no cell coordinate) are not served here:
// `hi` below (kNormal53HiBias + 16) << 16 says so, and the caller goes to normal53() above.
constexpr uint32_t kNormal53HiBias = 16u * 1023u + 48u;        // (hi >> 16) of double(16.0) is 16 x 1027 = bias + 16
__device__ __forceinline__ void normal53_prepare(uint32_t w, uint32_t companion, uint32_t &hi, double &t)
{
    const double D = __builtin_fma((double)(w & 0x7fffffffu), 0x1p21, (double)(companion >> 11));       // = q
    hi = (uint32_t)__double2hiint(D);
    const uint32_t lo = (uint32_t)__double2loint(D);
    const double X = __hiloint2double((int)((((hi << 4) | (lo >> 28)) & 0xFFFFFu) | 0x3FF00000u), (int)(lo << 4));
    const double H = __hiloint2double((int)(0x80100000u - (hi & 0x7FF00000u)), 0);                        // 2^(3 - e)
    t = (X - 1.0) + H;
}
// (c = the row's eight coefficients)
__device__ __forceinline__ double normal53_evaluate(uint32_t w, const double (&c)[8], double t)
{
    double z = c[7];
#pragma unroll
    for (int d = 6; d >= 0; --d) z = __builtin_fma(z, t, c[d]);
    // z0 = Phi^-1(tail probability) < 0: the sign bit of w flips it
    return __hiloint2double((int)((uint32_t)__double2hiint(z) ^ (w & 0x80000000u)), __double2loint(z));
}
// the 53-bit uniform's numerator q (u = q / 2^53) of a draw
__device__ __forceinline__ uint64_t uniform53(uint32_t w, uint32_t companion)
{
    return ((uint64_t)w << 21) | (uint64_t)(companion >> 11);
}
constexpr uint32_t kCompanion = 0x8000u;         // counter word 3 of a draw's companion block: | kCompanion

__device__ __forceinline__ float normal_from_u32(uint32_t w, const float4 *__restrict__ tab)
{
    return normal_from_u32_rows(w, [tab](uint32_t off) -> float4 { return tab[(off >> 4) - kNormalRowBias]; });
}

__device__ __forceinline__ double u32_to_unit(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }

__device__ __forceinline__ uint32_t stint_compound(int track, int remaining_laps)
{
    // reference :420-429 and :469-478
    if (track == 2) return 4u;           // wet -> WET
    if (track == 1) return 3u;           // damp -> INTERMEDIATE
    if (remaining_laps > 30) return 2u;  // HARD
    if (remaining_laps > 15) return 1u;  // MEDIUM
    return 0u;                           // SOFT
}

}  // namespace mcgp
