// Host stand-in for csrc/race_isa.hip.h (tools/emu only): same include guard, portable bodies.
#ifndef MCGP_RACE_ISA_H
#define MCGP_RACE_ISA_H
#include <hip/hip_runtime.h>
namespace mcgp {
inline void minmax_f64(double a, double b, double &lo, double &hi)
{
    lo = a < b ? a : b;       // times are finite, non-negative: identical to v_min_f64 / v_max_f64
    hi = a < b ? b : a;
}
inline double max_f64(double a, double b) { return a < b ? b : a; }
inline double max_abs_f64(double a, double b) { return std::fabs(a) < b ? b : std::fabs(a); }
inline void cmpx_time(double &ca, uint32_t &pa, double &cb, uint32_t &pb)
{
    if (ca > cb) {
        const double t = ca; ca = cb; cb = t;
        const uint32_t q = pa; pa = pb; pb = q;
    }
}
inline void cmpx_time2(double &a1, uint32_t &pa1, double &b1, uint32_t &pb1, double &a2, uint32_t &pa2, double &b2, uint32_t &pb2)
{
    cmpx_time(a1, pa1, b1, pb1);
    cmpx_time(a2, pa2, b2, pb2);
}
inline void bubble_fwd2(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2)
{
    cmpx_time(c0, p0, c1, p1);
    cmpx_time(c1, p1, c2, p2);
}
inline void bubble_bwd2(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2)
{
    cmpx_time(c1, p1, c2, p2);
    cmpx_time(c0, p0, c1, p1);
}
inline void bubble_fwd3(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3)
{
    cmpx_time(c0, p0, c1, p1);
    cmpx_time(c1, p1, c2, p2);
    cmpx_time(c2, p2, c3, p3);
}
inline void bubble_bwd3(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3)
{
    cmpx_time(c2, p2, c3, p3);
    cmpx_time(c1, p1, c2, p2);
    cmpx_time(c0, p0, c1, p1);
}
inline void bubble_fwd4(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3, double &c4, uint32_t &p4)
{
    cmpx_time(c0, p0, c1, p1);
    cmpx_time(c1, p1, c2, p2);
    cmpx_time(c2, p2, c3, p3);
    cmpx_time(c3, p3, c4, p4);
}
inline void bubble_bwd4(double &c0, uint32_t &p0, double &c1, uint32_t &p1, double &c2, uint32_t &p2, double &c3, uint32_t &p3, double &c4, uint32_t &p4)
{
    cmpx_time(c3, p3, c4, p4);
    cmpx_time(c2, p2, c3, p3);
    cmpx_time(c1, p1, c2, p2);
    cmpx_time(c0, p0, c1, p1);
}
// LDS by absolute byte address: the emulated LDS block is the array mcgp::smem
extern unsigned char smem[];
template <typename T>
inline T lds_ld(uint32_t addr)
{
    T v;
    std::memcpy(&v, smem + addr, sizeof(T));
    return v;
}
template <typename T>
inline void lds_st(uint32_t addr, T v)
{
    std::memcpy(smem + addr, &v, sizeof(T));
}
inline void lds_or_u32(uint32_t addr, uint32_t bits) { lds_st<uint32_t>(addr, lds_ld<uint32_t>(addr) | bits); }
struct f64x2 {
    double x, y;
};
inline f64x2 lds_ld_f64x2(uint32_t addr) { return lds_ld<f64x2>(addr); }
inline void lds_ld_f64_u32x2(uint32_t addr, double &d, uint32_t &u0, uint32_t &u1)
{
    d = lds_ld<double>(addr);
    u0 = lds_ld<uint32_t>(addr + 8);
    u1 = lds_ld<uint32_t>(addr + 12);
}
// v_cvt_u32_f64: toward zero, saturating, NaN -> 0
inline uint32_t cvt_u32_f64_sat(double x)
{
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
// the emulated threads run one after another: a wave's turn-th chunk is a fixed one
inline uint32_t next_ticket(uint32_t *, uint32_t tid, uint32_t turn, int waves) { return turn * (uint32_t)waves + tid / 64u; }
inline uint32_t peek_ticket(const uint32_t *counter) { return *counter; }
inline uint32_t first_lane_with(bool pred) { return pred ? 0u : 64u; }
inline void sched_fence() {}
inline double ceil_f64(double x) { return std::ceil(x); }
inline uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
template <uint32_t STRIDE, bool CEIL = true>
inline void ovt_threshold(double dl, double od, uint32_t row, uint32_t &thr, uint32_t &next)
{
    const uint32_t stride = STRIDE;
    const bool c = dl > od;
    thr = c ? min_u32(cvt_u32_f64_sat(CEIL ? std::ceil(dl) : dl), 0x80000000u) : 0u;
    next = row + (c ? stride : 0u);
}
// threads of the emulated block run one after another: "any lane" is this thread alone (both paths behind an
// MCGP_ANY test compute the same results, so which one a thread takes does not matter); EMU_FORCE_ANY=1 sends every
// thread down the "some lane needs it" path, which a single-thread view would otherwise reach only rarely
#ifdef EMU_FORCE_ANY
#define MCGP_ANY(pred) ((void)(pred), true)
#else
#define MCGP_ANY(pred) (pred)
#endif
inline void pin(uint32_t &) {}
inline void pin(double &) {}
inline void pin_scalar(uint32_t &) {}
template <typename T>
inline void pin_ptr(const T *&) {}
inline float4 lds_ld_float4(uint32_t addr) { return lds_ld<float4>(addr); }
inline uint32_t lds_base_of(const void *p) { return (uint32_t)((const unsigned char *)p - smem); }
}  // namespace mcgp
#endif
