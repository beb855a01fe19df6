/* hip_emu.h -- TEST INFRASTRUCTURE: a minimal single-process SIMT emulator.
 *
 * Lets the product's kernel source (peaksegdisk_amd/csrc/) be compiled by g++ and
 * run on the CPU so kernel *logic* can be debugged and parity-tested without a GPU.  Each
 * GPU thread is a fiber (hand-rolled x86-64 context switch); the blocks of a launch are dealt
 * to a few host threads (PSD_EMU_THREADS, default: the cores, at most 8), each of which runs its
 * blocks one after another with its own copy of every __shared__ variable (thread_local);
 * ballot/shfl/wave_sync and __syncthreads() are rendezvous points that every lane of the
 * wave (every thread of the block) must reach in the same order -- the discipline the
 * kernels follow on the real hardware too (psd_platform.h).
 *
 * Never linked into or loaded by peaksegdisk_amd: only tests/ builds libpeaksegdisk_emu.so.
 * It checks logic, not LDS hazards, memory ordering or performance; the `-m gpu` tests on
 * an MI355X remain the parity tests proper.
 */
#ifndef PSD_HIP_EMU_H
#define PSD_HIP_EMU_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct ThreadCtx {
  dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
};
extern thread_local ThreadCtx *g_cur;
unsigned long long ballot(bool p);
double shfl_f64(double v, int src);
int shfl_i32(int v, int src);
void wave_sync();
void syncthreads();
void yield_fiber(); /* let the other fibers run (spin-wait loops) */
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
}  // namespace emu

#define threadIdx (emu::g_cur->threadIdx_)
#define blockIdx (emu::g_cur->blockIdx_)
#define blockDim (emu::g_cur->blockDim_)
#define gridDim (emu::g_cur->gridDim_)

static inline void __syncthreads() { emu::syncthreads(); }

/* (the fibers of a block never run concurrently, but blocks on other host threads do) */
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
static inline unsigned int atomicAdd(unsigned int *p, unsigned int v) {
  return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicAdd_system(int *p, int v) { return atomicAdd(p, v); }
static inline int atomicMax(int *p, int v) {
  int o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return o;
}
static inline int atomicCAS(int *p, int cmp, int v) {
  (void)__atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}

/* ---- host runtime subset ------------------------------------------------------------ */
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorOutOfMemory 2
#define hipErrorNoDevice 100
typedef void *hipStream_t;
struct emu_event {
  double t_ms;
};
typedef emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeClockRate };
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int device);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
#define hipHostMallocDefault 0u
#define hipHostMallocMapped 2u
#define hipHostMallocCoherent 0x40000000u
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void *p, int v, size_t n);
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, struct emu_event *e, unsigned flags);
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);

template <class T>
static inline hipError_t hipMalloc(T **p, size_t n) {
  return hipMalloc((void **)p, n);
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

#endif
