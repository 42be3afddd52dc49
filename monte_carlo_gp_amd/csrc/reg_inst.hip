// reg_inst.hip -- one explicit instantiation of the register-resident race kernel per object file.
// Built as reg_inst_<n>.o with -DMCGP_INST_N=<n> (see Makefile); the sizes must match
// MCGP_REG_SIZES in mcgp_hip.hip.
#include "race_kernel_reg.hip.h"

#ifndef MCGP_INST_N
#error "compile with -DMCGP_INST_N=<field size>"
#endif

namespace mcgp {
template __global__ void race_kernel_reg<MCGP_INST_N>(const KParams *, uint64_t, uint64_t, uint32_t, uint32_t,
                                                      unsigned long long *, uint8_t *, const uint8_t *, uint32_t, uint32_t *, uint32_t *);
// the small block shape (kSmallBlockWaves): what a launch falls back to when the default block does not fit the LDS on offer
template __global__ void race_kernel_reg<MCGP_INST_N, kSmallBlockWaves>(const KParams *, uint64_t, uint64_t, uint32_t, uint32_t,
                                                                        unsigned long long *, uint8_t *, const uint8_t *, uint32_t,
                                                                        uint32_t *, uint32_t *);
template __global__ void race_kernel_reg_batch<MCGP_INST_N>(const KParams *, const BatchItem *, uint32_t, uint64_t,
                                                            unsigned long long *, uint32_t, uint32_t *, uint32_t *);
// the reference-width build (mcgp_config.deviates = MCGP_DEVIATES_53)
template __global__ void race_kernel_reg_wide<MCGP_INST_N>(const KParams *, uint64_t, uint64_t, uint32_t, uint32_t,
                                                           unsigned long long *, uint8_t *, const uint8_t *, uint32_t, uint32_t *,
                                                           uint32_t *, const double *);
}  // namespace mcgp
