// Shared device/host helpers for the REFIL gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace refil {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// thread-local error string returned by refil_last_error()
void set_error(const char* fmt, ...);

#define REFIL_CHECK(cond, ...)                    \
    do {                                          \
        if (!(cond)) {                            \
            refil::set_error(__VA_ARGS__);        \
            return 1;                             \
        }                                         \
    } while (0)

#define REFIL_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            refil::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

#define REFIL_LAUNCH_CHECK()                                                              \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess) {                                                           \
            refil::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline long cdivl(long a, long b) { return (a + b - 1) / b; }

// physical memory row of logical row r:  (r / grp) * gstride + (r % grp) + off.
// Branch-free on purpose (it sits in the GEMM load path): the identity is encoded as grp = 2^30.
// The quotient comes from a multiplication by magic = ceil(2^40 / grp) (exact for r * grp < 2^40, checked where the
// map is built): an integer division costs ~20 VALU instructions per lane, and the row-list kernels evaluate a map
// per lane per step right beside their MFMAs.
struct RowMap {
    int grp, gstride, off;
    unsigned long long magic;
    __host__ __device__ inline long operator()(int r) const {
        const int q = (int)(((unsigned long long)(unsigned)r * magic) >> 40);
        return (long)q * gstride + (r - q * grp) + off;
    }
};
// identity (grp == 0) is encoded as grp = 2^30. rows = an upper bound of the logical row numbers the map will see.
inline RowMap make_rowmap(int grp, int gstride, int off) {
    if (!grp) { grp = 1 << 30; gstride = 0; off = 0; }
    return RowMap{grp, gstride, off, ((1ull << 40) + (unsigned long long)grp - 1) / (unsigned long long)grp};
}
inline bool rowmap_exact(int grp, long rows) { return !grp || ((long)grp <= (1L << 20) && rows * (long)grp < (1L << 40)); }

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt = [15:14][3:0], expcnt [6:4], lgkmcnt [11:8])
constexpr int vmcnt_only(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }

// Workgroup barrier that orders LDS traffic only. __syncthreads() is a workgroup-scope release/acquire FENCE around the
// barrier: on gfx9 the compiler implements it with s_waitcnt vmcnt(0), i.e. every global load AND store in flight is
// drained at every barrier -- in a recurrence step loop that puts the write-acknowledge latency of the step's own stores
// (and the prefetch distance of its loads) on the per-step critical path. Here only the LDS operations are completed
// (lgkmcnt) before the barrier; the "memory" clobber keeps the compiler from moving memory operations across it. Use it
// where the data exchanged between the waves lives in LDS and global memory is private to a lane.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// hardware-rate versions for the per-step GRU gate math (v_exp_f32 + v_rcp_f32, ~1 ulp each; both saturate
// correctly: exp2 -> inf gives 0 resp. +-1). expf / tanhf / IEEE division cost ~10x the instructions, and the
// persistent GRU runs one wave per SIMD, so gate-math instructions are directly on the per-step critical path.
__device__ inline float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ inline float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }

}  // namespace refil
