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
    if (f.unpack_width > 0) {                                  // bit-packed mask rows -> bytes (4 output bytes per thread step)
        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(src);
        const int W = f.unpack_width;
        const long stride1 = (long)gridDim.x * blockDim.x;
        if ((W & 3) == 0 && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {      // 4 mask bytes per store
            const long n4 = n >> 2;
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride1) {
                const long row = (4 * i) / W;
                const unsigned bits = (unsigned)(w[row] >> (int)(4 * i - row * W)) & 15u;
                reinterpret_cast<unsigned*>(dst)[i] = (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
            }
            return;
        }
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride1) {
            const long row = i / W;
            dst[i] = (char)((w[row] >> (int)(i - row * W)) & 1ull);
        }
        return;
    }
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

namespace refil {
__global__ __launch_bounds__(256) void pack_mask_bits_kernel(const uint8_t* src, unsigned long long* dst, long rows, int width) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
        const unsigned long long w = __ballot(lane < width && src[r * width + (lane < width ? lane : 0)] != 0);
        if (lane == 0) dst[r] = w;
    }
}
}  // namespace refil

extern "C" int refil_pack_mask_bits(const uint8_t* src, uint64_t* dst, int64_t rows, int32_t width, void* stream) {
    REFIL_CHECK(src && dst && rows >= 0 && width >= 1 && width <= 64, "refil_pack_mask_bits: bad arguments (width 1..64)");
    if (rows == 0) return 0;
    ProfScope prof("pack_mask_bits_kernel", 0.0, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(refil::pack_mask_bits_kernel, dim3((int)min(65535L, cdivl(rows, 4))), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<unsigned long long*>(dst), (long)rows, width);
    REFIL_LAUNCH_CHECK();
    return 0;
}

extern "C" int refil_replay_gather(const refil_gather_field* fields, int32_t n_fields, const int64_t* episode_ids,
                                   int32_t B, int64_t capacity, void* stream) {
    REFIL_CHECK(fields && episode_ids, "refil_replay_gather: null pointer");
    REFIL_CHECK(n_fields > 0 && n_fields <= MAX_GATHER_FIELDS, "refil_replay_gather: 1..%d fields (got %d)", MAX_GATHER_FIELDS, n_fields);
    REFIL_CHECK(B > 0 && capacity > 0, "refil_replay_gather: bad batch size / capacity");
    GatherK k;
    long maxb = 0;
    for (int i = 0; i < n_fields; ++i) {
        const refil_gather_field& f = fields[i];
        REFIL_CHECK(f.unpack_width >= 0 && f.unpack_width <= 64, "refil_replay_gather: field %d: unpack_width must be 0..64", i);
        REFIL_CHECK(f.src && f.dst && f.copy_bytes >= 0 && f.copy_bytes <= f.dst_episode_bytes &&
                    (f.unpack_width ? cdivl(f.copy_bytes, f.unpack_width) * 8 : f.copy_bytes) <= f.src_episode_bytes,
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
