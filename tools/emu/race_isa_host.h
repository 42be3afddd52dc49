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
}  // namespace mcgp
#endif
