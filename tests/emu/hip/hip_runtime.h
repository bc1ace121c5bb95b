// TEST INFRASTRUCTURE ONLY -- a minimal stand-in for <hip/hip_runtime.h> that lets tests/emu build the chaining-stage
// sources (cactus_amd/csrc/mp_kernels.hip + mp_chain.cpp) for the HOST, so that the `-m "not gpu"` suite can exercise
// the host orchestration and the kernels' logic where there is no GPU.  A workgroup is emulated by one pthread per
// work-item: __syncthreads is a barrier over the group, __shfl_xor exchanges through a per-wave (64 consecutive
// work-items) buffer between two wave barriers.  Workgroups run one after another.  Nothing of this is shipped, linked
// into libmiblast.so, or measured; the product has no CPU path.
#pragma once
#define MB_EMU_HIP_STANDIN 1                  // (mb_guard.h: no virtual-memory calls here -- the electric fence is a device-only tool)

#include <pthread.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100, hipErrorInvalidDevice = 101;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

inline const char *hipGetErrorString(hipError_t) { return "emulated HIP"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

namespace emu {
struct Group {
    pthread_barrier_t all;
    std::vector<pthread_barrier_t> wave;
    std::vector<unsigned long long> slot;                    // one exchange slot per work-item
};
extern thread_local Group *g_group;
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void launch(const std::function<void()> &body, dim3 grid, dim3 block);
}  // namespace emu

#define threadIdx emu::t_threadIdx
#define blockIdx emu::t_blockIdx
#define blockDim emu::t_blockDim
#define gridDim emu::t_gridDim

inline void __syncthreads() { pthread_barrier_wait(&emu::g_group->all); }

template <typename T>
inline T __shfl_xor(T v, int mask) {
    static_assert(sizeof(T) <= 8, "exchange slot is 8 bytes");
    emu::Group *g = emu::g_group;
    const unsigned tid = emu::t_threadIdx.x, w = tid >> 6;
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    g->slot[tid] = raw;
    pthread_barrier_wait(&g->wave[w]);
    raw = g->slot[(tid & ~63u) | ((tid ^ (unsigned)mask) & 63u)];
    pthread_barrier_wait(&g->wave[w]);
    T out;
    memcpy(&out, &raw, sizeof(T));
    return out;
}

template <typename T> inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <typename T> inline T atomicMax(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch([&] { kernel(__VA_ARGS__); }, (grid), (block))
