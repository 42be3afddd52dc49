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

}  // namespace mcgp
#endif
