// common.h -- internal helpers of libnerfpp_hip (gfx950 only).
#pragma once

#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "nerfpp_hip.h"
#include "nrf_math.h"

namespace nrf {

void set_error(const char *fmt, ...);
// the library's own stream-ordered scratch (scratch.hip): a buffer for work enqueued on `st`; given back, it serves the next taker ON THAT STREAM.  Drop-in replacements of
// hipMallocAsync / hipFreeAsync (same return type), which proved unsafe inside the LibTorch host
hipError_t scratch_take(void **out, size_t bytes, hipStream_t st);
hipError_t scratch_give(void *p, hipStream_t st);

#define NRF_CHECK_ARG(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            ::nrf::set_error(__VA_ARGS__);                         \
            return NRF_ERR_INVALID_ARG;                            \
        }                                                          \
    } while (0)

#define NRF_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            ::nrf::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return NRF_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

#define NRF_LAUNCH_CHECK()                                                                        \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) {                                                                   \
            ::nrf::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return NRF_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

#define NRF_TRY(expr)                  \
    do {                               \
        int s_ = (expr);               \
        if (s_ != NRF_OK) return s_;   \
    } while (0)

// One-time kernel attribute setup (the dynamic-LDS window of a kernel) is per DEVICE, not per process: a process that drives several GPUs (the nrf_comm_* C ABI
// allows one communicator per device) must set it on each.  needed() is true until done() has been called for the calling thread's current device; the setup
// itself is idempotent, so two first callers racing on the same device both run it.
struct PerDeviceOnce {
    std::atomic<uint64_t> mask{0};
    static uint64_t bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool needed() const { return (mask.load(std::memory_order_acquire) & bit()) == 0; }
    void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- profiling (nrf_profile_*): HIP events on the caller's stream around the dominant kernels ----
struct ProfScope {
    int slot;
    hipStream_t stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool active;
    ProfScope(int slot, hipStream_t stream);
    ~ProfScope();
};

// ---- small device helpers ----

// Orders LDS traffic between the lanes of ONE wavefront (hardware executes a wave's DS ops in order; the fences
// stop the compiler from moving accesses across, the barrier pins the schedule).  No instruction is emitted beyond
// the waits the compiler derives.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Bbox {
    float mn[3];
    float mx[3];
};

struct F3 {
    float x, y, z;
};

// A sample point is either read from an explicit [p,3] array or formed as o + d*z from a packed ray
// batch and a depth table (NeRFRenderer.h:419) -- the fused pipeline never materialises pts.
struct PointSource {
    const float *pts;     // explicit points, or nullptr
    const float *rays;    // [n, ray_stride] packed rays
    const float *z;       // [n, s]
    int ray_stride;
    int s;
};

__device__ __forceinline__ F3 load_point(const PointSource &ps, int64_t i)
{
    F3 r;
    if (ps.pts) {
        r.x = ps.pts[i * 3 + 0];
        r.y = ps.pts[i * 3 + 1];
        r.z = ps.pts[i * 3 + 2];
    } else {
        // 32-bit division when the index fits (always, for a chunk): the 64-bit one is a ~60-instruction sequence
        const int64_t ray = (i >> 31) == 0 ? (int64_t)((uint32_t)i / (uint32_t)ps.s) : i / ps.s;
        const float *rp = ps.rays + ray * ps.ray_stride;
        const float zz = ps.z[i];
        r.x = rp[0] + rp[3] * zz;
        r.y = rp[1] + rp[4] * zz;
        r.z = rp[2] + rp[5] * zz;
    }
    return r;
}

}  // namespace nrf
