// hip/hip_runtime.h -- HOST stand-in used ONLY by tools/emu (a debugging build of the kernel sources for the CPU).
// It lets csrc/race_kernel_reg.hip.h compile with g++ so that a kernel edit can be checked against the oracle in
// seconds before it is sent to a GPU.  Execution model: the threads of ONE block run one after another, phase by
// phase (tools/emu/emu_kernel.cpp); __syncthreads() is a no-op and LDS is a plain array.  Nothing under monte_carlo_gp_amd/
// includes or links this: the product has no CPU path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__
#define __align__(x)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct emu_dim3 {
    unsigned x, y, z;
};
extern emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

struct float4 {
    float x, y, z, w;
};

inline void __syncthreads() {}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
inline double __hiloint2double(int hi, int lo)
{
    const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    double d;
    std::memcpy(&d, &u, 8);
    return d;
}
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b)
{
    return (unsigned long long)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
}
inline int __double2loint(double x)
{
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return (int)(uint32_t)u;
}
inline int __double2hiint(double x)
{
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return (int)(uint32_t)(u >> 32);
}
inline uint32_t __float_as_uint(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float __uint_as_float(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
template <typename T>
inline T atomicAdd(T *p, T v)
{
    const T old = *p;
    *p = old + v;
    return old;
}
