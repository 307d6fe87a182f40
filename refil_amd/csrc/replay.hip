// Replay sampling on the device: gathers the sampled episodes of every scheme field from the ring storage into one
// staging minibatch (reference: ReplayBuffer.sample -> EpisodeBatch.__getitem__ fancy indexing,
// src/components/episode_buffer.py:123-159,233-240, followed by run.py:266-273). The reference issues one
// index_select per field plus a max_t_filled() host synchronisation to trim the time axis; here ONE launch copies
// all fields, and no trimming is needed because the learner skips the steps after an episode's end on the device
// (kernels.h: ListArgs). Pure byte movement: 16-byte vector copies when source, destination and sizes allow.
#include "kernels.h"
#include "profile.h"

namespace refil {

constexpr int MAX_GATHER_FIELDS = 24;
struct GatherK {
    refil_gather_field f[MAX_GATHER_FIELDS];
    const int64_t* ids; int B; long capacity;
};

__global__ __launch_bounds__(256) void replay_gather_kernel(GatherK k) {
    const refil_gather_field& f = k.f[blockIdx.z];
    const int b = blockIdx.y;
    const long ep = k.ids[b];
    if (ep < 0 || ep >= k.capacity) return;                    // (checked on the host side for host-visible ids)
    const char* src = static_cast<const char*>(f.src) + ep * f.src_episode_bytes;
    char* dst = static_cast<char*>(f.dst) + (long)b * f.dst_episode_bytes;
    const long n = f.copy_bytes;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const long stride = (long)gridDim.x * blockDim.x;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const long n16 = n >> 4;
        for (long i = tid; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (long i = (n16 << 4) + tid; i < n; i += stride) dst[i] = src[i];
    } else {
        for (long i = tid; i < n; i += stride) dst[i] = src[i];
    }
}

}  // namespace refil

using namespace refil;

extern "C" int refil_replay_gather(const refil_gather_field* fields, int32_t n_fields, const int64_t* episode_ids,
                                   int32_t B, int64_t capacity, void* stream) {
    REFIL_CHECK(fields && episode_ids, "refil_replay_gather: null pointer");
    REFIL_CHECK(n_fields > 0 && n_fields <= MAX_GATHER_FIELDS, "refil_replay_gather: 1..%d fields (got %d)", MAX_GATHER_FIELDS, n_fields);
    REFIL_CHECK(B > 0 && capacity > 0, "refil_replay_gather: bad batch size / capacity");
    GatherK k;
    long maxb = 0;
    for (int i = 0; i < n_fields; ++i) {
        const refil_gather_field& f = fields[i];
        REFIL_CHECK(f.src && f.dst && f.copy_bytes >= 0 && f.copy_bytes <= f.src_episode_bytes && f.copy_bytes <= f.dst_episode_bytes,
                    "refil_replay_gather: field %d: bad pointers / sizes", i);
        k.f[i] = f;
        maxb = max(maxb, (long)f.copy_bytes);
    }
    k.ids = episode_ids; k.B = B; k.capacity = capacity;
    const int bx = (int)max(1L, min(64L, cdivl(maxb, 256L * 16 * 4)));      // ~64 bytes per thread of the largest field
    ProfScope prof("replay_gather_kernel", 0.0, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(replay_gather_kernel, dim3(bx, B, n_fields), dim3(256), 0, (hipStream_t)stream, k);
    REFIL_LAUNCH_CHECK();
    return 0;
}
