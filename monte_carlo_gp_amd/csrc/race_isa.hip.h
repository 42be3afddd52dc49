// race_isa.hip.h -- the few gfx950 instructions the race kernels issue directly (inline assembly).
#ifndef MCGP_RACE_ISA_H
#define MCGP_RACE_ISA_H
#include <hip/hip_runtime.h>

namespace mcgp {

// v_min_f64 / v_max_f64 issued directly: through fmin()/fmax() hipcc adds a canonicalising
// v_max_f64 x, x, x per operand (IEEE mode quiets signalling NaNs), tripling the cost.  Times are
// finite and non-negative here, for which the bare instructions are exact.  Plain VALU, interlocked
// by hardware: no wait states needed inside the statement.
__device__ __forceinline__ void minmax_f64(double a, double b, double &lo, double &hi)
{
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
}

__device__ __forceinline__ double max_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max(|a|, b): the magnitude of a through the instruction's source modifier (no extra instruction)
__device__ __forceinline__ double max_abs_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Compare-exchange on the time alone, the payload word following its time: after it (ca, pa) holds the smaller
// time.  Equal times are left as they are (stable).  One statement so that the two f64 selects sit between the
// compare and the payload selects: the v_cmp -> v_cndmask (VCC) dependency needs two wait states, and the compiler
// left to itself pads them with an s_nop per comparator.
__device__ __forceinline__ void cmpx_time(double &ca, uint32_t &pa, double &cb, uint32_t &pb)
{
    double lo, hi;
    uint32_t p0, p1;
    asm("v_cmp_gt_f64 vcc, %[a], %[b]\n\t"
        "v_min_f64 %[lo], %[a], %[b]\n\t"
        "v_max_f64 %[hi], %[a], %[b]\n\t"
        "v_cndmask_b32 %[p0], %[pa], %[pb], vcc\n\t"
        "v_cndmask_b32 %[p1], %[pb], %[pa], vcc"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [p0] "=&v"(p0), [p1] "=&v"(p1)
        : [a] "v"(ca), [b] "v"(cb), [pa] "v"(pa), [pb] "v"(pb)
        : "vcc");
    ca = lo; cb = hi; pa = p0; pb = p1;
}

// ---- merged comparators ----
// hipcc pads every inline-asm statement that touches VCC with an s_nop; several comparators per statement
// amortise it (and keep the v_cmp -> v_cndmask spacing without padding).  All are compare-exchanges on the time
// alone with the payload word following its time, exactly cmpx_time.

// two independent comparators (a1, b1) and (a2, b2)
__device__ __forceinline__ void cmpx_time2(double &a1, uint32_t &pa1, double &b1, uint32_t &pb1,
                                           double &a2, uint32_t &pa2, double &b2, uint32_t &pb2)
{
    double lo1, hi1, lo2, hi2;
    uint32_t x1, y1, x2, y2;
    asm(
        "v_cmp_gt_f64 vcc, %[a1], %[b1]\n\t"
        "v_min_f64 %[lo1], %[a1], %[b1]\n\t"
        "v_max_f64 %[hi1], %[a1], %[b1]\n\t"
        "v_cndmask_b32 %[x1], %[pa1], %[pb1], vcc\n\t"
        "v_cndmask_b32 %[y1], %[pb1], %[pa1], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[a2], %[b2]\n\t"
        "v_min_f64 %[lo2], %[a2], %[b2]\n\t"
        "v_max_f64 %[hi2], %[a2], %[b2]\n\t"
        "v_cndmask_b32 %[x2], %[pa2], %[pb2], vcc\n\t"
        "v_cndmask_b32 %[y2], %[pb2], %[pa2], vcc"
        : [lo1] "=&v"(lo1), [hi1] "=&v"(hi1), [x1] "=&v"(x1), [y1] "=&v"(y1), [lo2] "=&v"(lo2), [hi2] "=&v"(hi2),
          [x2] "=&v"(x2), [y2] "=&v"(y2)
        : [a1] "v"(a1), [b1] "v"(b1), [pa1] "v"(pa1), [pb1] "v"(pb1), [a2] "v"(a2), [b2] "v"(b2), [pa2] "v"(pa2),
          [pb2] "v"(pb2)
        : "vcc");
    a1 = lo1; b1 = hi1; pa1 = x1; pb1 = y1;
    a2 = lo2; b2 = hi2; pa2 = x2; pb2 = y2;
}

// forward bubble over 3 consecutive slots: comparators (0,1), (1,2), .. in sequence, the maximum ends in the last
__device__ __forceinline__ void bubble_fwd2(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2)
{
    double lo1, h1, lo2, h2;
    uint32_t q1, r1, q2, r2;
    asm(
        "v_cmp_gt_f64 vcc, %[c0], %[c1]\n\t"
        "v_min_f64 %[lo1], %[c0], %[c1]\n\t"
        "v_max_f64 %[h1], %[c0], %[c1]\n\t"
        "v_cndmask_b32 %[q1], %[p0], %[p1], vcc\n\t"
        "v_cndmask_b32 %[r1], %[p1], %[p0], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[h1], %[c2]\n\t"
        "v_min_f64 %[lo2], %[h1], %[c2]\n\t"
        "v_max_f64 %[h2], %[h1], %[c2]\n\t"
        "v_cndmask_b32 %[q2], %[r1], %[p2], vcc\n\t"
        "v_cndmask_b32 %[r2], %[p2], %[r1], vcc"
        : [lo1] "=&v"(lo1), [h1] "=&v"(h1), [q1] "=&v"(q1), [r1] "=&v"(r1), [lo2] "=&v"(lo2), [h2] "=&v"(h2), [q2] "=&v"(q2), [r2] "=&v"(r2)
        : [c0] "v"(c0), [p0] "v"(p0), [c1] "v"(c1), [p1] "v"(p1), [c2] "v"(c2), [p2] "v"(p2)
        : "vcc");
    c0 = lo1; p0 = q1;
    c1 = lo2; p1 = q2;
    c2 = h2; p2 = r2;
}

// backward bubble over 3 consecutive slots: comparators (1,2), (0,1), .. in sequence, the minimum ends in slot 0
__device__ __forceinline__ void bubble_bwd2(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2)
{
    double l1, hi1, l2, hi2;
    uint32_t q1, r1, q2, r2;
    asm(
        "v_cmp_gt_f64 vcc, %[c1], %[c2]\n\t"
        "v_min_f64 %[l1], %[c1], %[c2]\n\t"
        "v_max_f64 %[hi1], %[c1], %[c2]\n\t"
        "v_cndmask_b32 %[q1], %[p1], %[p2], vcc\n\t"
        "v_cndmask_b32 %[r1], %[p2], %[p1], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[c0], %[l1]\n\t"
        "v_min_f64 %[l2], %[c0], %[l1]\n\t"
        "v_max_f64 %[hi2], %[c0], %[l1]\n\t"
        "v_cndmask_b32 %[q2], %[p0], %[q1], vcc\n\t"
        "v_cndmask_b32 %[r2], %[q1], %[p0], vcc"
        : [l1] "=&v"(l1), [hi1] "=&v"(hi1), [q1] "=&v"(q1), [r1] "=&v"(r1), [l2] "=&v"(l2), [hi2] "=&v"(hi2), [q2] "=&v"(q2), [r2] "=&v"(r2)
        : [c0] "v"(c0), [p0] "v"(p0), [c1] "v"(c1), [p1] "v"(p1), [c2] "v"(c2), [p2] "v"(p2)
        : "vcc");
    c2 = hi1; p2 = r1;
    c1 = hi2; p1 = r2;
    c0 = l2; p0 = q2;
}

// forward bubble over 4 consecutive slots: comparators (0,1), (1,2), .. in sequence, the maximum ends in the last
__device__ __forceinline__ void bubble_fwd3(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3)
{
    double lo1, h1, lo2, h2, lo3, h3;
    uint32_t q1, r1, q2, r2, q3, r3;
    asm(
        "v_cmp_gt_f64 vcc, %[c0], %[c1]\n\t"
        "v_min_f64 %[lo1], %[c0], %[c1]\n\t"
        "v_max_f64 %[h1], %[c0], %[c1]\n\t"
        "v_cndmask_b32 %[q1], %[p0], %[p1], vcc\n\t"
        "v_cndmask_b32 %[r1], %[p1], %[p0], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[h1], %[c2]\n\t"
        "v_min_f64 %[lo2], %[h1], %[c2]\n\t"
        "v_max_f64 %[h2], %[h1], %[c2]\n\t"
        "v_cndmask_b32 %[q2], %[r1], %[p2], vcc\n\t"
        "v_cndmask_b32 %[r2], %[p2], %[r1], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[h2], %[c3]\n\t"
        "v_min_f64 %[lo3], %[h2], %[c3]\n\t"
        "v_max_f64 %[h3], %[h2], %[c3]\n\t"
        "v_cndmask_b32 %[q3], %[r2], %[p3], vcc\n\t"
        "v_cndmask_b32 %[r3], %[p3], %[r2], vcc"
        : [lo1] "=&v"(lo1), [h1] "=&v"(h1), [q1] "=&v"(q1), [r1] "=&v"(r1), [lo2] "=&v"(lo2), [h2] "=&v"(h2), [q2] "=&v"(q2), [r2] "=&v"(r2), [lo3] "=&v"(lo3), [h3] "=&v"(h3), [q3] "=&v"(q3), [r3] "=&v"(r3)
        : [c0] "v"(c0), [p0] "v"(p0), [c1] "v"(c1), [p1] "v"(p1), [c2] "v"(c2), [p2] "v"(p2), [c3] "v"(c3), [p3] "v"(p3)
        : "vcc");
    c0 = lo1; p0 = q1;
    c1 = lo2; p1 = q2;
    c2 = lo3; p2 = q3;
    c3 = h3; p3 = r3;
}

// backward bubble over 4 consecutive slots: comparators (2,3), (1,2), .. in sequence, the minimum ends in slot 0
__device__ __forceinline__ void bubble_bwd3(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3)
{
    double l1, hi1, l2, hi2, l3, hi3;
    uint32_t q1, r1, q2, r2, q3, r3;
    asm(
        "v_cmp_gt_f64 vcc, %[c2], %[c3]\n\t"
        "v_min_f64 %[l1], %[c2], %[c3]\n\t"
        "v_max_f64 %[hi1], %[c2], %[c3]\n\t"
        "v_cndmask_b32 %[q1], %[p2], %[p3], vcc\n\t"
        "v_cndmask_b32 %[r1], %[p3], %[p2], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[c1], %[l1]\n\t"
        "v_min_f64 %[l2], %[c1], %[l1]\n\t"
        "v_max_f64 %[hi2], %[c1], %[l1]\n\t"
        "v_cndmask_b32 %[q2], %[p1], %[q1], vcc\n\t"
        "v_cndmask_b32 %[r2], %[q1], %[p1], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[c0], %[l2]\n\t"
        "v_min_f64 %[l3], %[c0], %[l2]\n\t"
        "v_max_f64 %[hi3], %[c0], %[l2]\n\t"
        "v_cndmask_b32 %[q3], %[p0], %[q2], vcc\n\t"
        "v_cndmask_b32 %[r3], %[q2], %[p0], vcc"
        : [l1] "=&v"(l1), [hi1] "=&v"(hi1), [q1] "=&v"(q1), [r1] "=&v"(r1), [l2] "=&v"(l2), [hi2] "=&v"(hi2), [q2] "=&v"(q2), [r2] "=&v"(r2), [l3] "=&v"(l3), [hi3] "=&v"(hi3), [q3] "=&v"(q3), [r3] "=&v"(r3)
        : [c0] "v"(c0), [p0] "v"(p0), [c1] "v"(c1), [p1] "v"(p1), [c2] "v"(c2), [p2] "v"(p2), [c3] "v"(c3), [p3] "v"(p3)
        : "vcc");
    c3 = hi1; p3 = r1;
    c2 = hi2; p2 = r2;
    c1 = hi3; p1 = r3;
    c0 = l3; p0 = q3;
}

// forward bubble over 5 consecutive slots: comparators (0,1), (1,2), .. in sequence, the maximum ends in the last
__device__ __forceinline__ void bubble_fwd4(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3, double &c4, uint32_t &p4)
{
    double lo1, h1, lo2, h2, lo3, h3, lo4, h4;
    uint32_t q1, r1, q2, r2, q3, r3, q4, r4;
    asm(
        "v_cmp_gt_f64 vcc, %[c0], %[c1]\n\t"
        "v_min_f64 %[lo1], %[c0], %[c1]\n\t"
        "v_max_f64 %[h1], %[c0], %[c1]\n\t"
        "v_cndmask_b32 %[q1], %[p0], %[p1], vcc\n\t"
        "v_cndmask_b32 %[r1], %[p1], %[p0], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[h1], %[c2]\n\t"
        "v_min_f64 %[lo2], %[h1], %[c2]\n\t"
        "v_max_f64 %[h2], %[h1], %[c2]\n\t"
        "v_cndmask_b32 %[q2], %[r1], %[p2], vcc\n\t"
        "v_cndmask_b32 %[r2], %[p2], %[r1], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[h2], %[c3]\n\t"
        "v_min_f64 %[lo3], %[h2], %[c3]\n\t"
        "v_max_f64 %[h3], %[h2], %[c3]\n\t"
        "v_cndmask_b32 %[q3], %[r2], %[p3], vcc\n\t"
        "v_cndmask_b32 %[r3], %[p3], %[r2], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[h3], %[c4]\n\t"
        "v_min_f64 %[lo4], %[h3], %[c4]\n\t"
        "v_max_f64 %[h4], %[h3], %[c4]\n\t"
        "v_cndmask_b32 %[q4], %[r3], %[p4], vcc\n\t"
        "v_cndmask_b32 %[r4], %[p4], %[r3], vcc"
        : [lo1] "=&v"(lo1), [h1] "=&v"(h1), [q1] "=&v"(q1), [r1] "=&v"(r1), [lo2] "=&v"(lo2), [h2] "=&v"(h2), [q2] "=&v"(q2), [r2] "=&v"(r2), [lo3] "=&v"(lo3), [h3] "=&v"(h3), [q3] "=&v"(q3), [r3] "=&v"(r3), [lo4] "=&v"(lo4), [h4] "=&v"(h4), [q4] "=&v"(q4), [r4] "=&v"(r4)
        : [c0] "v"(c0), [p0] "v"(p0), [c1] "v"(c1), [p1] "v"(p1), [c2] "v"(c2), [p2] "v"(p2), [c3] "v"(c3), [p3] "v"(p3), [c4] "v"(c4), [p4] "v"(p4)
        : "vcc");
    c0 = lo1; p0 = q1;
    c1 = lo2; p1 = q2;
    c2 = lo3; p2 = q3;
    c3 = lo4; p3 = q4;
    c4 = h4; p4 = r4;
}

// backward bubble over 5 consecutive slots: comparators (3,4), (2,3), .. in sequence, the minimum ends in slot 0
__device__ __forceinline__ void bubble_bwd4(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3, double &c4, uint32_t &p4)
{
    double l1, hi1, l2, hi2, l3, hi3, l4, hi4;
    uint32_t q1, r1, q2, r2, q3, r3, q4, r4;
    asm(
        "v_cmp_gt_f64 vcc, %[c3], %[c4]\n\t"
        "v_min_f64 %[l1], %[c3], %[c4]\n\t"
        "v_max_f64 %[hi1], %[c3], %[c4]\n\t"
        "v_cndmask_b32 %[q1], %[p3], %[p4], vcc\n\t"
        "v_cndmask_b32 %[r1], %[p4], %[p3], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[c2], %[l1]\n\t"
        "v_min_f64 %[l2], %[c2], %[l1]\n\t"
        "v_max_f64 %[hi2], %[c2], %[l1]\n\t"
        "v_cndmask_b32 %[q2], %[p2], %[q1], vcc\n\t"
        "v_cndmask_b32 %[r2], %[q1], %[p2], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[c1], %[l2]\n\t"
        "v_min_f64 %[l3], %[c1], %[l2]\n\t"
        "v_max_f64 %[hi3], %[c1], %[l2]\n\t"
        "v_cndmask_b32 %[q3], %[p1], %[q2], vcc\n\t"
        "v_cndmask_b32 %[r3], %[q2], %[p1], vcc\n\t"
        "v_cmp_gt_f64 vcc, %[c0], %[l3]\n\t"
        "v_min_f64 %[l4], %[c0], %[l3]\n\t"
        "v_max_f64 %[hi4], %[c0], %[l3]\n\t"
        "v_cndmask_b32 %[q4], %[p0], %[q3], vcc\n\t"
        "v_cndmask_b32 %[r4], %[q3], %[p0], vcc"
        : [l1] "=&v"(l1), [hi1] "=&v"(hi1), [q1] "=&v"(q1), [r1] "=&v"(r1), [l2] "=&v"(l2), [hi2] "=&v"(hi2), [q2] "=&v"(q2), [r2] "=&v"(r2), [l3] "=&v"(l3), [hi3] "=&v"(hi3), [q3] "=&v"(q3), [r3] "=&v"(r3), [l4] "=&v"(l4), [hi4] "=&v"(hi4), [q4] "=&v"(q4), [r4] "=&v"(r4)
        : [c0] "v"(c0), [p0] "v"(p0), [c1] "v"(c1), [p1] "v"(p1), [c2] "v"(c2), [p2] "v"(p2), [c3] "v"(c3), [p3] "v"(p3), [c4] "v"(c4), [p4] "v"(p4)
        : "vcc");
    c4 = hi1; p4 = r1;
    c3 = hi2; p3 = r2;
    c2 = hi3; p2 = r3;
    c1 = hi4; p1 = r4;
    c0 = l4; p0 = q4;
}

// ---- selects through VCC ----
// A select whose mask sits in an SGPR pair is VOP3-encoded (v_cndmask_b32_e64) and occupies the SIMD for ~4.2 cycles; the
// VOP2 form, which reads its mask from VCC, for ~2.3 (tools/valu_peak.hip).  Where one compare feeds several selects the
// block is written out: the compare into VCC, the selects right behind.  gfx950 needs two wait states between a VALU write
// of VCC and a VALU read of it and does not interlock them: at least two independent instructions sit in between.

// One adjacent pair of an overtake pass (reference :516-524, pace deltas x 2^31): the pair is a candidate iff dl > od
// (NaN -- a retired car -- is not); thr = the draw-word threshold min(ceil(dl), 2^31) of a candidate, 0 otherwise; `next` =
// `row` advanced by `stride` for a candidate, so that the W-plane address of a pair's draw word is a running sum over the
// pairs before it instead of a popcount.  (STRIDE is an assembly-time literal: a register holding it would be one more
// value alive across the whole pass.)
// CEIL = false (the reference-width build): thr = min(floor(dl), 2^31) -- the conversion truncates by itself.
template <uint32_t STRIDE, bool CEIL = true>
__device__ __forceinline__ void ovt_threshold(double dl, double od, uint32_t row, uint32_t &thr, uint32_t &next)
{
    if constexpr (CEIL) {
        double t;
        asm("v_cmp_lt_f64 vcc, %[od], %[dl]\n\t"
            "v_ceil_f64 %[t], %[dl]\n\t"
            "v_cvt_u32_f64 %[thr], %[t]\n\t"
            "v_min_u32 %[thr], 0x80000000, %[thr]\n\t"
            "v_add_u32 %[next], %[stride], %[row]\n\t"
            "v_cndmask_b32 %[thr], 0, %[thr], vcc\n\t"
            "v_cndmask_b32 %[next], %[row], %[next], vcc"
            : [t] "=&v"(t), [thr] "=&v"(thr), [next] "=&v"(next)
            : [dl] "v"(dl), [od] "s"(od), [row] "v"(row), [stride] "n"(STRIDE)
            : "vcc");
    } else {
        asm("v_cmp_lt_f64 vcc, %[od], %[dl]\n\t"
            "v_cvt_u32_f64 %[thr], %[dl]\n\t"
            "v_add_u32 %[next], %[stride], %[row]\n\t"
            "v_min_u32 %[thr], 0x80000000, %[thr]\n\t"
            "v_cndmask_b32 %[next], %[row], %[next], vcc\n\t"
            "v_cndmask_b32 %[thr], 0, %[thr], vcc"
            : [thr] "=&v"(thr), [next] "=&v"(next)
            : [dl] "v"(dl), [od] "s"(od), [row] "v"(row), [stride] "n"(STRIDE)
            : "vcc");
    }
}

// LDS access by ABSOLUTE byte address.  The register kernel declares no static __shared__ data, so its
// dynamic LDS block starts at address 0 (checked once at kernel entry); addressing it by number instead of
// through the `extern __shared__` symbol lets the compiler fold every table / row base into the 16-bit
// immediate offset of the ds_read / ds_write instead of adding the (symbolic) base in a VALU instruction.
#define MCGP_LDS __attribute__((address_space(3)))
template <typename T>
__device__ __forceinline__ T lds_ld(uint32_t addr)
{
    return *reinterpret_cast<const MCGP_LDS T *>(addr);
}
template <typename T>
__device__ __forceinline__ void lds_st(uint32_t addr, T v)
{
    *reinterpret_cast<MCGP_LDS T *>(addr) = v;
}
// OR into a word of LDS without reading it back: one ds_or_b32, nothing to wait for
__device__ __forceinline__ void lds_or_u32(uint32_t addr, uint32_t bits)
{
    (void)__hip_atomic_fetch_or(reinterpret_cast<MCGP_LDS uint32_t *>(addr), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ float4 lds_ld_float4(uint32_t addr)          // one ds_read_b128
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = *reinterpret_cast<const MCGP_LDS v4f *>(addr);
    return make_float4(v.x, v.y, v.z, v.w);
}
struct f64x2 {
    double x, y;
};
__device__ __forceinline__ f64x2 lds_ld_f64x2(uint32_t addr)             // one ds_read_b128 (16-byte aligned address)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = *reinterpret_cast<const MCGP_LDS v2d *>(addr);
    return f64x2{v.x, v.y};
}
// 16-byte stores / loads of two doubles or four words (16-byte aligned address)
__device__ __forceinline__ void lds_st_f64x2(uint32_t addr, double a, double b)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d v;
    v.x = a;
    v.y = b;
    *reinterpret_cast<MCGP_LDS v2d *>(addr) = v;
}
__device__ __forceinline__ void lds_st_u32x4(uint32_t addr, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u v;
    v.x = a; v.y = b; v.z = c; v.w = d;
    *reinterpret_cast<MCGP_LDS v4u *>(addr) = v;
}
__device__ __forceinline__ void lds_ld_u32x4(uint32_t addr, uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = *reinterpret_cast<const MCGP_LDS v4u *>(addr);
    a = v.x; b = v.y; c = v.z; d = v.w;
}
// {f64, u32, u32}: one ds_read_b128 (16-byte aligned address)
__device__ __forceinline__ void lds_ld_f64_u32x2(uint32_t addr, double &d, uint32_t &u0, uint32_t &u1)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = *reinterpret_cast<const MCGP_LDS v4u *>(addr);
    d = __hiloint2double((int)v.y, (int)v.x);
    u0 = v.z;
    u1 = v.w;
}

// binary64 -> uint32 the way the hardware converts: toward zero, saturating at 0 and 2^32 - 1, NaN -> 0.
// (A C++ cast is undefined outside the range; the draw-word thresholds of the overtake step rely on the clamp.)
__device__ __forceinline__ uint32_t cvt_u32_f64_sat(double x)
{
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ double ceil_f64(double x) { return __builtin_ceil(x); }
__device__ __forceinline__ uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
// scheduling fence: the compiler's instruction scheduler moves nothing across it
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// ---- cross-lane pieces of the cooperative event handler ----
// LDS written by some lanes of the wave, read by others: program order is enough for the hardware (DS operations of
// one wave execute in order); this keeps the compiler from moving LDS accesses across the hand-over.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_f64(double x, int lane)       // lane: wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bpermute_f64(double x, int src_lane)   // src_lane: per lane
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
// The wave's next work ticket: one lane takes it from the launch's counter in device memory, every lane gets the
// value (wave-uniform).  `turn` and `waves` are for the host emulation, whose threads run one after another.
__device__ __forceinline__ uint32_t next_ticket(uint32_t *counter, uint32_t /*tid*/, uint32_t /*turn*/, int /*waves*/)
{
    uint32_t t = 0u;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}
// a ticket counter's current value, without taking a ticket
__device__ __forceinline__ uint32_t peek_ticket(const uint32_t *counter)
{
    return __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the first lane of the wavefront in which the predicate holds, 64 if there is none (wave-uniform result)
__device__ __forceinline__ uint32_t first_lane_with(bool pred)
{
    const unsigned long long m = __ballot(pred);
    return m ? (uint32_t)__builtin_ctzll(m) : 64u;
}
// true if the predicate holds in ANY lane of the wavefront (wave-uniform result)
#define MCGP_ANY(pred) (__any((int)(pred)) != 0)
// Keeps a value computed where it is written: the compiler may not sink its computation into one arm of a later
// select and turn the select into a branch (no instruction is emitted).
__device__ __forceinline__ void pin(uint32_t &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
// the same for a wave-uniform value (an SGPR)
__device__ __forceinline__ void pin_scalar(uint32_t &x) { asm volatile("" : "+s"(x)); }
// a wave-uniform pointer the optimiser cannot see through: loads through it stay where they are written
template <typename T>
__device__ __forceinline__ void pin_ptr(const T *&p) { asm volatile("" : "+s"(p)); }

// address of the dynamic LDS block as the hardware sees it
__device__ __forceinline__ uint32_t lds_base_of(const void *p)
{
    return (uint32_t)(uintptr_t)(const MCGP_LDS void *)p;
}

}  // namespace mcgp
#endif
