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
// visibility of the staged data rests on fine-grained (system-coherent) staging memory (hipExtMallocWithFlags), the
// system-scope release / acquire pair on the flag and cache-bypassing peer loads -- and on the start-up equality check.
#include <dlfcn.h>
#include <stdlib.h>
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
    bool finegrained;              // staging + flag live in fine-grained (system-coherent) device memory
    unsigned* status_host;         // pinned, device-mapped word: the reduce kernel sets it when a peer did not arrive
    unsigned* status_dev;          // its device address
    unsigned long long timeout_ticks;   // 100 MHz wall-clock ticks a rank waits for its peers
};

struct ReduceArgs {
    const float* src[16];
    const unsigned* flag[16];
    unsigned* status;              // [1] host-mapped: set to the epoch that timed out
    float* out;
    long n;
    int world;
    unsigned epoch;
    unsigned long long timeout_ticks;
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
                if (wall_clock64() - t0 > a.timeout_ticks) { ok = 0; break; }      // a peer died (REFIL_ONESHOT_TIMEOUT_S, default 120 s)
                __builtin_amdgcn_s_sleep(32);
            }
        }
        if (!ok) __hip_atomic_store(a.status, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok_s = ok;
    }
    __syncthreads();
    if (!ok_s) {
        // nothing was reduced: poison the buffer so that an optimiser step behind this launch cannot silently use rank-local
        // gradients (the host sees the status word at its next call and raises)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) a.out[i] = __builtin_nanf("");
        return;
    }
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
    // Staging buffers and the flag word are read by PEER devices while this device keeps running: fine-grained (system-coherent)
    // device memory, so that a peer's loads are not served from a stale copy in this device's L2 and the staged data is visible
    // at the release store of the flag, not only at the end of the kernel. (Plain hipMalloc memory is coarse-grained: coherent
    // across devices at kernel boundaries only.) REFIL_ONESHOT_COARSE=1 goes back to hipMalloc; the start-up equality check of
    // dp.OneShotAllReduce guards either choice.
    static const bool coarse = [] { const char* e = getenv("REFIL_ONESHOT_COARSE"); return e && e[0] == '1'; }();
    auto alloc = [&](void** p, size_t n) -> hipError_t {
        if (!coarse) {
            if (hipExtMallocWithFlags(p, n, hipDeviceMallocFinegrained) == hipSuccess) { o->finegrained = true; return hipSuccess; }
            (void)hipGetLastError();
        }
        return hipMalloc(p, n);
    };
    for (int k = 0; k < 2; ++k) {
        REFIL_HIP(alloc((void**)&o->in[k], bytes));
        REFIL_HIP(hipMemset(o->in[k], 0, bytes));
    }
    REFIL_HIP(alloc((void**)&o->flag, 256));
    REFIL_HIP(hipMemset(o->flag, 0, 256));
    REFIL_HIP(hipHostMalloc((void**)&o->status_host, 64, hipHostMallocMapped));
    *o->status_host = 0;
    REFIL_HIP(hipHostGetDevicePointer((void**)&o->status_dev, o->status_host, 0));
    const char* te = getenv("REFIL_ONESHOT_TIMEOUT_S");
    const double tsec = te && atof(te) > 0 ? atof(te) : 120.0;
    o->timeout_ticks = (unsigned long long)(tsec * 1.0e8);
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
    // a peer that did not arrive in an EARLIER call leaves NaNs in that call's buffer and its epoch here: fatal, the replicas
    // can no longer be trusted to hold the same parameters (and the double-buffer argument above no longer holds)
    REFIL_CHECK(*static_cast<volatile unsigned*>(o->status_host) == 0,
                "refil_oneshot_allreduce: a peer did not arrive within the timeout in call %u (REFIL_ONESHOT_TIMEOUT_S); the buffer of that call was "
                "poisoned with NaN -- restart the job or use the backend's all-reduce", *static_cast<volatile unsigned*>(o->status_host));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned e = ++o->epoch;
    float* stage = o->in[e & 1];
    REFIL_HIP(hipMemcpyAsync(stage, inout, (size_t)o->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(oneshot_signal_kernel, dim3(1), dim3(1), 0, st, o->flag, e);
    ReduceArgs a;
    memset(&a, 0, sizeof(a));
    for (int r = 0; r < o->world; ++r) { a.src[r] = o->peer_in[r][e & 1]; a.flag[r] = o->peer_flag[r]; }
    a.status = o->status_dev; a.out = inout; a.n = o->n; a.world = o->world; a.epoch = e; a.timeout_ticks = o->timeout_ticks;
    const int blocks = (int)min((long)64, cdivl(o->n, 4 * 256));
    hipLaunchKernelGGL(oneshot_reduce_kernel, dim3(max(blocks, 1)), dim3(256), 0, st, a);
    REFIL_LAUNCH_CHECK();
    return 0;
}

extern "C" int refil_oneshot_status(void* ctx, int32_t* timed_out) {
    OneShot* o = static_cast<OneShot*>(ctx);
    REFIL_CHECK(o && timed_out, "refil_oneshot_status: bad arguments");
    *timed_out = (int32_t)*static_cast<volatile unsigned*>(o->status_host);      // (host-mapped word: no synchronisation; 0 or the call that timed out)
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
    if (o->status_host) (void)hipHostFree(o->status_host);
    (void)hipGetLastError();
    delete o;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// refil_allreduce_flat: the step's ONE collective for a host that is not Python: all-reduce(SUM)
// of the flat fp32 buffer [gradients | stat sums] on the CALLER's RCCL communicator, enqueued on the caller's stream --
// what refil_amd/dp.py gets from torch.distributed (backend "nccl" = RCCL). librccl is resolved at the first call
// (dlopen: the library itself does not link against RCCL, a single-GPU host never loads it).
// ------------------------------------------------------------------------------------------------
namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
struct Rccl { void* h; nccl_allreduce_fn allreduce; nccl_errstr_fn errstr; };
Rccl* rccl() {
    static Rccl r = [] {
        Rccl x{nullptr, nullptr, nullptr};
        const char* names[] = {getenv("REFIL_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !n[0]) continue;
            x.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (x.h) break;
        }
        if (x.h) {
            x.allreduce = reinterpret_cast<nccl_allreduce_fn>(dlsym(x.h, "ncclAllReduce"));
            x.errstr = reinterpret_cast<nccl_errstr_fn>(dlsym(x.h, "ncclGetErrorString"));
        }
        return x;
    }();
    return &r;
}
}  // namespace

extern "C" int refil_allreduce_flat(float* buf, int64_t n_floats, void* comm, void* stream) {
    REFIL_CHECK(buf && n_floats > 0 && comm, "refil_allreduce_flat: null buffer / communicator");
    Rccl* r = rccl();
    REFIL_CHECK(r->h && r->allreduce, "refil_allreduce_flat: librccl.so not found (set REFIL_RCCL_LIB)");
    const int rc = r->allreduce(buf, buf, (size_t)n_floats, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, static_cast<hipStream_t>(stream));
    REFIL_CHECK(rc == 0, "refil_allreduce_flat: ncclAllReduce failed: %s", r->errstr ? r->errstr(rc) : "?");
    return 0;
}
