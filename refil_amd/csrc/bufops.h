// Buffer-instruction helpers shared by the kernels that predicate their global accesses with out-of-range offsets: a wave-
// uniform resource (base, size) + a per-lane byte offset; an offset >= the size is dropped by the hardware's range check
// (loads return zeros, stores vanish, no data moves, no branch) -- see attention_mfma.hip.
#pragma once
#include "common.h"

namespace refil {

typedef unsigned int u32x4 __attribute__((vector_size(16)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr int BUF_OOB = 0x7ffffff0;          // >= every num_records used here: the access is dropped by the range check
constexpr int BUF_MAX = 0x7fffffe0;

__device__ inline rsrc_t mk_rsrc(const void* base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > BUF_MAX ? BUF_MAX : bytes), 0x00020000);
}
__device__ inline float4 buf_ld4(rsrc_t rs, int off) {
    // (the whole vector is bit-cast at once: __builtin_bit_cast of ONE element of a vector reads element 0 in this clang)
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ inline unsigned long long buf_ld_u64(rsrc_t rs, int off) {
    return __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}
__device__ inline void buf_st4(rsrc_t rs, int off, f32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 0); }


}  // namespace refil
