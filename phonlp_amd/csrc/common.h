// Shared host/device helpers for libphonlp_heads.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/phonlp_heads.h"

namespace phonlp {

// Records a printf-style message for phonlp_last_error() (thread-local) and returns `code`.
int fail(int code, const char *fmt, ...);

// Checks the launch that was just enqueued.
int check_launch(const char *what);

// Optional per-kernel timing (phonlp_profile_begin/_end): records a hipEvent pair on the launch stream
// around the kernel launched inside the scope.  A no-op unless profiling is enabled.
struct ProfScope {
    ProfScope(const char *tag, hipStream_t stream);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

constexpr int kWave = 64;

// Launch-once work that is per DEVICE: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device
// only, so a process that drives several GPUs must repeat it on each.  Two threads racing here both do the (idempotent)
// work; nobody skips it.
struct PerDeviceOnce {
    std::atomic<unsigned char> done[64];
    bool pending(int *dev) {
        int d = 0;
        (void)hipGetDevice(&d);
        *dev = d;
        return d < 0 || d >= 64 || !done[d].load(std::memory_order_acquire);
    }
    void mark(int d) {
        if (d >= 0 && d < 64) done[d].store(1, std::memory_order_release);
    }
};

#ifdef __HIPCC__
// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): unlike __syncthreads()
// it does not wait for global loads/stores in flight, so prefetches and epilogue stores stay asynchronous.
// Use only where no global data is exchanged between the waves of the block across the barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// KP: hidden features + ones column, zero-padded to the MFMA k-group (8 floats = two 16-byte LDS reads).
__host__ __device__ inline int kp_of(int H) { return (H + 1 + 7) / 8 * 8; }

}  // namespace phonlp
