/* psd_platform.h -- the (small) set of HIP device features the kernels use.
 *
 * Product build: hipcc --offload-arch=gfx950, real intrinsics.
 * Test build (-DPSD_EMU, tests/emu/ only): the same kernel source compiled by g++ against
 * tests/emu/hip_emu.h, a fiber-based SIMT emulator used to debug kernel logic without a
 * GPU.  The emulator is never built into or loaded by the product library.
 *
 * Rules the kernels follow so both builds mean the same thing:
 *   - wavefront = 64 lanes, hard-coded;
 *   - every cross-lane operation (ballot/shfl/wave_sync) and __syncthreads() is reached in
 *     wave-uniform (resp. block-uniform) control flow by ALL lanes;
 *   - lanes of a wave exchange data through LDS only across a wave_sync().
 */
#ifndef PSD_PLATFORM_H
#define PSD_PLATFORM_H

#ifdef PSD_EMU
#include "hip_emu.h"
#define PSD_D static inline
#define PSD_M inline
#define PSD_NOINLINE static __attribute__((noinline))
#define PSD_LDS static thread_local /* (a block per host thread at a time) */
#define PSD_COLD_DEV static __attribute__((noinline, cold))
#else
#include <hip/hip_runtime.h>
#define PSD_D __device__ __forceinline__
#define PSD_M __device__ __forceinline__
/* out-of-line device functions: what keeps the forward kernel's hot loop inside the 64 KB
 * instruction cache (with everything inlined it was ~200 KB) */
#define PSD_NOINLINE __device__ __attribute__((noinline))
#define PSD_LDS __shared__
#define PSD_COLD_DEV __device__ __attribute__((noinline, cold))
#endif

/* Data in HBM reached through pointers that a device function loads from memory: without
 * the address space in the type the compiler must emit FLAT instructions, which are issued
 * to the LDS and the vector-memory pipelines both -- slower, and slowed further by any wave
 * that polls an LDS flag meanwhile. */
#ifdef PSD_EMU
typedef double gdouble;
typedef int gint;
typedef unsigned long long gull;
typedef double ldouble; /* a double in LDS reached through a computed pointer */
#else
typedef double __attribute__((address_space(3))) ldouble;
typedef double __attribute__((address_space(1))) gdouble;
typedef int __attribute__((address_space(1))) gint;
typedef unsigned long long __attribute__((address_space(1))) gull;
#endif

#if defined(__clang__)
#define PSD_ASSUME(x) __builtin_assume(x)
#else
#define PSD_ASSUME(x)           \
  do {                          \
    if (!(x)) __builtin_unreachable(); \
  } while (0)
#endif

#include "peakseg_detmath.h"

namespace psd {

constexpr int WAVE = 64;

PSD_D int lane_id() { return (int)(threadIdx.x & 63u); }
PSD_D int wave_id() { return (int)(threadIdx.x >> 6); }

#ifdef PSD_EMU
PSD_D unsigned long long ballot(bool p) { return emu::ballot(p); }
PSD_D double shfl_d(double v, int src) { return emu::shfl_f64(v, src); }
PSD_D int shfl_i(int v, int src) { return emu::shfl_i32(v, src); }
PSD_D void wave_sync() { emu::wave_sync(); }
PSD_D int uniform_i(int v) { return v; }
PSD_D double uniform_d(double v) { return v; }
/* value of lane `src`; src must be wave-uniform */
PSD_D int rdlane_i(int v, int src) { return emu::shfl_i32(v, src); }
PSD_D double rdlane_d(double v, int src) { return emu::shfl_f64(v, src); }
#else
PSD_D unsigned long long ballot(bool p) { return __ballot(p); }
PSD_D double shfl_d(double v, int src) { return __shfl(v, src, 64); }
PSD_D int shfl_i(int v, int src) { return __shfl(v, src, 64); }
/* Lanes of one wave execute LDS instructions in order; the fences keep the compiler from
 * moving this lane's LDS accesses across the point where it relies on another lane's. */
PSD_D void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
/* tell the compiler a value is wave-uniform (keeps it in SGPRs) */
PSD_D int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
PSD_D double uniform_d(double v) {
  uint64_t u = psd_d2u(v);
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return psd_u2d(((uint64_t)hi << 32) | lo);
}
/* value of lane `src`; src must be wave-uniform: v_readlane_b32 takes a few cycles where a
 * general shuffle (ds_bpermute) or an LDS broadcast read costs 60-100 in a dependent chain */
PSD_D int rdlane_i(int v, int src) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
PSD_D double rdlane_d(double v, int src) {
  uint64_t u = psd_d2u(v);
  int s = __builtin_amdgcn_readfirstlane(src);
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, s);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), s);
  return psd_u2d(((uint64_t)hi << 32) | lo);
}
#endif

#ifdef PSD_EMU
PSD_D long long cycle_now() { return 0; }
#else
PSD_D long long cycle_now() { return (long long)__builtin_readcyclecounter(); }
#endif

/* flag hand-off between two waves of a workgroup through LDS */
#ifdef PSD_EMU
PSD_D int flag_load(const int *p) { return *(const volatile int *)p; }
PSD_D void flag_store(int *p, int v) { *(volatile int *)p = v; }
PSD_D void spin_pause() { emu::yield_fiber(); }
PSD_D void device_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#else
PSD_D int flag_load(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
PSD_D void flag_store(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* between two polls of a flag.  Measured on the latency build: polling without a pause takes
 * LDS cycles from the waves at work (-1.2 %), and longer sleeps ended by s_wakeup from the
 * posting wave are slower still (-2 % to -4 %): profiles/r02/ab_step_barrier.log */
PSD_D void spin_pause() { __builtin_amdgcn_s_sleep(1); }
/* make this thread's global stores visible to the other waves of the device */
PSD_D void device_fence() { __threadfence(); }
#endif

/* words shared with the host while a kernel runs (pinned host memory): system scope.  Only
 * loads and fetch-add: PCIe atomics know no max. */
#ifdef PSD_EMU
PSD_D unsigned long long sys_load_u64(const unsigned long long *p) {
  return __atomic_load_n(p, __ATOMIC_ACQUIRE);
}
PSD_D void sys_add_u64(unsigned long long *p, unsigned long long v) {
  (void)__atomic_fetch_add(p, v, __ATOMIC_RELEASE);
}
/* an entry of a device table that other waves of the launch fill in */
PSD_D char *agent_load_ptr(char *const *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
PSD_D void agent_store_ptr(char **p, char *v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
#else
PSD_D unsigned long long sys_load_u64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
PSD_D void sys_add_u64(unsigned long long *p, unsigned long long v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
PSD_D char *agent_load_ptr(char *const *p) {
  return (char *)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
}
PSD_D void agent_store_ptr(char **p, char *v) {
  __hip_atomic_store((unsigned long long *)p, (unsigned long long)v, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
#endif

template <class T>
PSD_D T *uniform_p(T *p) {
#ifdef PSD_EMU
  return p;
#else
  uint64_t u = (uint64_t)p;
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return (T *)(((uint64_t)hi << 32) | lo);
#endif
}

PSD_D int popc64(unsigned long long m) { return __builtin_popcountll(m); }
/* index of the lowest set bit; m != 0 */
PSD_D int ctz64(unsigned long long m) { return __builtin_ctzll(m); }
/* index of the highest set bit; m != 0 */
PSD_D int msb64(unsigned long long m) { return 63 - __builtin_clzll(m); }
PSD_D unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

}  // namespace psd
#endif
