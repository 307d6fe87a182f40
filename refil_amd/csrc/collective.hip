// One-shot all-reduce over peer memory (SURVEY.md section 8e topology note; REFIL_ALLREDUCE=oneshot in refil_amd/dp.py).
//
// The step's message ([gradients | stat sums], 1.7 MB at the north-star shape) is latency-bound: a ring all-reduce pays
// 2 (N-1) hops. xGMI is a full point-to-point mesh, so every rank can instead READ its N-1 peers' buffers directly and
// add them up itself: one hop, one kernel, no intermediate copies -- and all ranks add in the same (rank) order, so the
// replicas stay bit-identical.
//
//   per rank:  in[2]  two staging buffers of n floats, exported with hipIpcGetMemHandle (double buffered: a peer may
//                     still read the buffer of step k while this rank already stages step k+1)
//              flag   one word, IPC-exported as well: the number of the last step whose staging buffer is complete
//   step e:    copy inout -> in[e & 1]            (stream order; the kernel-end release makes it visible system-wide)
//              flag <- e                          (one-thread kernel, system-scope release store)
//              reduce kernel: wait until every peer's flag >= e (system-scope acquire loads, bounded by a wall-clock
//              timeout so that a dead peer cannot hang the GPU), then inout[i] = sum_r in_r[e & 1][i]
//   A rank finishes step e+1's reduction only after every peer has signalled e+1, i.e. after every peer has left the
//   reduction of step e: when buffer e & 1 is staged again at step e+2, nobody reads it any more.
//
// Validation status: exercised by two processes sharing ONE GPU (tests/test_gpu_dp.py: IPC handles, flag protocol,
// double buffering, equality with torch.distributed.all_reduce). No multi-GPU box was available: cross-device
// visibility of the staged data relies on the kernel-end release of the copy and on cache-bypassing peer loads.
#include <string.h>

#include "common.h"
#include "../../include/refil_hip.h"

namespace refil {

struct OneShot {
    int world, rank;
    long n;
    float* in[2];                  // own staging buffers
    unsigned* flag;                // own flag word (+ status word behind it)
    const float* peer_in[16][2];   // every rank's buffers as mapped here (own entries = own pointers)
    unsigned* peer_flag[16];
    unsigned epoch;
    bool connected;
};

struct ReduceArgs {
    const float* src[16];
    const unsigned* flag[16];
    unsigned* status;              // [1]: set to 1 on timeout
    float* out;
    long n;
    int world;
    unsigned epoch;
};

__global__ void oneshot_signal_kernel(unsigned* flag, unsigned epoch) {
    __threadfence_system();
    __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void oneshot_reduce_kernel(ReduceArgs a) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned long long t0 = wall_clock64();                 // 100 MHz
        for (int r = 0; r < a.world && ok; ++r) {
            while ((int)(__hip_atomic_load(a.flag[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.epoch) < 0) {
                if (wall_clock64() - t0 > 1000000000ull) { ok = 0; break; }        // 10 s: a peer died
                __builtin_amdgcn_s_sleep(32);
            }
        }
        if (!ok) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok_s = ok;
    }
    __syncthreads();
    if (!ok_s) return;
    __threadfence_system();
    const long n4 = a.n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < a.world; ++r) {                           // rank order: identical sums on every rank
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a.src[r]) + i);     // (bypasses the caches)
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(a.out)[i] = s;
    }
    for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < a.world; ++r) s += __builtin_nontemporal_load(a.src[r] + i);
        a.out[i] = s;
    }
}

}  // namespace refil

using namespace refil;

extern "C" int refil_oneshot_create(int32_t world, int32_t rank, int64_t n_floats, uint8_t* handles_out, void** ctx_out) {
    REFIL_CHECK(world >= 1 && world <= 16 && rank >= 0 && rank < world && n_floats > 0 && handles_out && ctx_out,
                "refil_oneshot_create: bad arguments (world <= 16)");
    static_assert(sizeof(hipIpcMemHandle_t) == REFIL_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    OneShot* o = new OneShot();
    memset(o, 0, sizeof(*o));
    o->world = world; o->rank = rank; o->n = n_floats;
    const size_t bytes = (((size_t)n_floats * sizeof(float)) + 255) & ~(size_t)255;
    for (int k = 0; k < 2; ++k) {
        REFIL_HIP(hipMalloc((void**)&o->in[k], bytes));
        REFIL_HIP(hipMemset(o->in[k], 0, bytes));
    }
    REFIL_HIP(hipMalloc((void**)&o->flag, 256));
    REFIL_HIP(hipMemset(o->flag, 0, 256));
    hipIpcMemHandle_t h;
    for (int k = 0; k < 3; ++k) {
        REFIL_HIP(hipIpcGetMemHandle(&h, k < 2 ? (void*)o->in[k] : (void*)o->flag));
        memcpy(handles_out + (size_t)k * REFIL_IPC_HANDLE_BYTES, &h, REFIL_IPC_HANDLE_BYTES);
    }
    REFIL_HIP(hipDeviceSynchronize());
    *ctx_out = o;
    return 0;
}

extern "C" int refil_oneshot_connect(void* ctx, const uint8_t* all_handles) {
    OneShot* o = static_cast<OneShot*>(ctx);
    REFIL_CHECK(o && all_handles && !o->connected, "refil_oneshot_connect: bad arguments");
    for (int r = 0; r < o->world; ++r) {
        if (r == o->rank) {
            o->peer_in[r][0] = o->in[0]; o->peer_in[r][1] = o->in[1]; o->peer_flag[r] = o->flag;
            continue;
        }
        for (int k = 0; k < 3; ++k) {
            hipIpcMemHandle_t h;
            memcpy(&h, all_handles + ((size_t)r * 3 + k) * REFIL_IPC_HANDLE_BYTES, REFIL_IPC_HANDLE_BYTES);
            void* p = nullptr;
            REFIL_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            if (k < 2) o->peer_in[r][k] = static_cast<const float*>(p);
            else o->peer_flag[r] = static_cast<unsigned*>(p);
        }
    }
    o->connected = true;
    return 0;
}

extern "C" int refil_oneshot_allreduce(void* ctx, float* inout, void* stream) {
    OneShot* o = static_cast<OneShot*>(ctx);
    REFIL_CHECK(o && o->connected && inout, "refil_oneshot_allreduce: not connected");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned e = ++o->epoch;
    float* stage = o->in[e & 1];
    REFIL_HIP(hipMemcpyAsync(stage, inout, (size_t)o->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(oneshot_signal_kernel, dim3(1), dim3(1), 0, st, o->flag, e);
    ReduceArgs a;
    memset(&a, 0, sizeof(a));
    for (int r = 0; r < o->world; ++r) { a.src[r] = o->peer_in[r][e & 1]; a.flag[r] = o->peer_flag[r]; }
    a.status = o->flag + 1; a.out = inout; a.n = o->n; a.world = o->world; a.epoch = e;
    const int blocks = (int)min((long)64, cdivl(o->n, 4 * 256));
    hipLaunchKernelGGL(oneshot_reduce_kernel, dim3(max(blocks, 1)), dim3(256), 0, st, a);
    REFIL_LAUNCH_CHECK();
    return 0;
}

extern "C" int refil_oneshot_status(void* ctx, int32_t* timed_out) {
    OneShot* o = static_cast<OneShot*>(ctx);
    REFIL_CHECK(o && timed_out, "refil_oneshot_status: bad arguments");
    unsigned s = 0;
    REFIL_HIP(hipMemcpy(&s, o->flag + 1, sizeof(s), hipMemcpyDeviceToHost));     // (synchronises: diagnostics only)
    *timed_out = (int32_t)s;
    return 0;
}

extern "C" int refil_oneshot_destroy(void* ctx) {
    OneShot* o = static_cast<OneShot*>(ctx);
    if (!o) return 0;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < o->world; ++r) {
        if (r == o->rank || !o->connected) continue;
        for (int k = 0; k < 2; ++k) if (o->peer_in[r][k]) (void)hipIpcCloseMemHandle(const_cast<float*>(o->peer_in[r][k]));
        if (o->peer_flag[r]) (void)hipIpcCloseMemHandle(o->peer_flag[r]);
    }
    for (int k = 0; k < 2; ++k) if (o->in[k]) (void)hipFree(o->in[k]);
    if (o->flag) (void)hipFree(o->flag);
    (void)hipGetLastError();
    delete o;
    return 0;
}
