// TEST INFRASTRUCTURE -- a host-side stand-in for <hip/hip_runtime.h> used ONLY by tests/emu/build_emu.py.
//
// It lets the kernel sources of refil_amd/csrc be compiled as plain x86 C++ and executed on a CPU wavefront emulator
// (emu_rt.cpp: one fiber per lane, 64 lanes per wave, wave-collective builtins, workgroup barriers, LDS), so that the CPU test
// tier can run the product's kernel SOURCE against the oracle where no GPU exists. Nothing under refil_amd/ includes, loads or
// links anything in tests/emu; the product library is built by hipcc from the same sources for gfx950 and has no CPU path.
//
// Semantics emulated: the lane layouts of the five MFMA instructions the kernels issue (cdna_hip_programming.md section 3 and
// tools/probes/mfma*_probe.hip, which were checked on gfx950), DPP controls, ds_permute / ds_bpermute, readlane / readfirstlane,
// shuffles, ballots, raw buffer loads / stores with the hardware range check, atomics, workgroup barriers.
// NOT emulated: timing, waitcnt hazards, register pressure, LDS bank conflicts, the memory model beyond x86's.
#pragma once
#define REFIL_EMU_BUILD 1

#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <tuple>
#include <type_traits>
#include <utility>

// ---- function / storage attributes ----
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// static LDS arrays: one workgroup runs on one OS thread at a time (emu_rt.cpp), so thread-local storage is workgroup-local
#define __shared__ static thread_local

// ---- vector types ----
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {

constexpr int XS = 64;      // bytes a lane can publish in one wave collective

struct Idx { unsigned x, y, z; };
struct LaneIds { Idx tid; int lane; };
struct BlockIds { Idx bid, bdim, gdim; };
// (a fiber never migrates between OS threads, and while it runs these point at its own ids)
extern thread_local const LaneIds* cur_lane_ids;
extern thread_local const BlockIds* cur_block_ids;
extern thread_local void* cur_dyn_smem;
static inline const LaneIds& lane_ids() { return *cur_lane_ids; }
static inline const BlockIds& block_ids() { return *cur_block_ids; }

struct Xchg { const unsigned char* tab; unsigned long long mask; };   // tab + XS * l = what lane l published; mask = participating lanes
Xchg xchg(const void* mine, int nbytes);      // (never inlined: its return address identifies the collective's call site)
void block_sync();
void lane_yield();           // a spin-wait iteration: lets the other lanes of the workgroup (and the OS) run
static inline void* dyn_smem() { return cur_dyn_smem; }      // the workgroup's dynamic LDS allocation
unsigned long long wall_ticks();

struct rsrc { char* base; long bytes; };

// executed-work counters (per kernel instantiation; read with emu_counters_dump): wave-level matrix instructions by type, bytes moved by
// the raw buffer instructions (in-range dwords only; plain pointer accesses are not seen)
enum Counter { C_MFMA_32x32x2_F32, C_MFMA_16x16x4_F32, C_MFMA_4x4x1_F32, C_MFMA_16x16x32_BF16, C_MFMA_32x32x16_BF16, C_BUF_LOAD_BYTES, C_BUF_STORE_BYTES, C_N };
extern thread_local unsigned long long counters[C_N];

}  // namespace emu

#define threadIdx (emu::lane_ids().tid)
#define blockIdx (emu::block_ids().bid)
#define blockDim (emu::block_ids().bdim)
#define gridDim (emu::block_ids().gdim)
#define warpSize 64

// ---- scalar helpers ----
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
#define __expf(x) expf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline unsigned long long wall_clock64() { return emu::wall_ticks(); }
template <class T> static inline T min(T a, T b) { return b < a ? b : a; }
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline long min(long a, int b) { return a < b ? a : b; }
static inline long min(int a, long b) { return a < b ? a : b; }
static inline long max(long a, int b) { return a > b ? a : b; }
static inline long max(int a, long b) { return a > b ? a : b; }

// ---- barriers / scheduling hints ----
static inline void __syncthreads() { emu::block_sync(); }
// A wave executes in lockstep on the hardware; here its lanes are fibers that run one after the other between two collectives. Every
// point where the kernels tell the COMPILER that lanes exchange data (wave_barrier around same-wave LDS traffic, explicit waitcnts,
// scheduling barriers) is therefore a real wave-level rendezvous in the emulator.
#define __builtin_amdgcn_wave_barrier() ((void)emu::xchg(nullptr, 0))
#define __builtin_amdgcn_s_waitcnt(x) ((void)emu::xchg(nullptr, 0))
#define __builtin_amdgcn_sched_barrier(x) ((void)emu::xchg(nullptr, 0))
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) emu::lane_yield()
#define __builtin_amdgcn_s_barrier() emu::block_sync()
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))

// ---- wave collectives ----
namespace emu {

template <class T>
static inline T lane_get(const Xchg& e, int l) { T v; memcpy(&v, e.tab + XS * l, sizeof(T)); return v; }
static inline bool lane_on(const Xchg& e, int l) { return (e.mask >> l) & 1ull; }

template <class T>
__attribute__((always_inline)) static inline T shfl_from(T v, int src) {
    static_assert(sizeof(T) <= XS, "payload");
    const Xchg e = xchg(&v, sizeof(T));
    if (!lane_on(e, src)) { T z; memset(&z, 0, sizeof(T)); return z; }      // (HIP's shuffles are ds_bpermute: a disabled source lane reads as 0)
    return lane_get<T>(e, src);
}
__attribute__((always_inline)) static inline int readfirstlane(int v) {
    const Xchg e = xchg(&v, 4);
    return lane_get<int>(e, __builtin_ctzll(e.mask));
}
__attribute__((always_inline)) static inline int readlane(int v, int l) {
    const Xchg e = xchg(&v, 4);
    return lane_get<int>(e, l & 63);
}
__attribute__((always_inline)) static inline unsigned long long ballot(bool p) {
    const unsigned char b = p ? 1 : 0;
    const Xchg e = xchg(&b, 1);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (lane_on(e, l) && e.tab[XS * l]) m |= 1ull << l;
    return m;
}
// v_mov_b32 dpp: the source lane of lane l under a DPP control (gfx9 encodings); -1 = no source (bound_ctrl -> 0, else `old`)
static inline int dpp_src(int l, int ctrl) {
    const int row = l & ~15, i = l & 15;
    if (ctrl <= 0xFF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                  // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; return i + n <= 15 ? row + i + n : -1; }   // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; return i - n >= 0 ? row + i - n : -1; }    // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row + ((i - n) & 15); }             // row_ror
    if (ctrl == 0x130) return l + 1 <= 63 ? l + 1 : -1;        // wave_shl:1
    if (ctrl == 0x134) return (l + 1) & 63;                    // wave_rol:1
    if (ctrl == 0x138) return l - 1 >= 0 ? l - 1 : -1;         // wave_shr:1
    if (ctrl == 0x13C) return (l - 1) & 63;                    // wave_ror:1
    if (ctrl == 0x140) return row + 15 - i;                    // row_mirror
    if (ctrl == 0x141) return row + (i & 8) + 7 - (i & 7);     // row_half_mirror
    fprintf(stderr, "emu: DPP control 0x%x not emulated\n", ctrl);
    abort();
}
__attribute__((always_inline)) static inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int l = lane_ids().lane;
    const Xchg e = xchg(&src, 4);
    if (!((row_mask >> (l >> 4)) & 1) || !((bank_mask >> ((l >> 2) & 3)) & 1)) return old;
    const int s = dpp_src(l, ctrl);
    if (s < 0 || !lane_on(e, s)) return bound_ctrl ? 0 : old;
    return lane_get<int>(e, s);
}
// ds_bpermute_b32: lane l reads the data of lane (addr / 4) % 64;  ds_permute_b32: lane l SENDS its data to lane (addr / 4) % 64
__attribute__((always_inline)) static inline int ds_bpermute(int addr, int data) {
    const Xchg e = xchg(&data, 4);
    const int s = (addr >> 2) & 63;
    return lane_on(e, s) ? lane_get<int>(e, s) : 0;
}
__attribute__((always_inline)) static inline int ds_permute(int addr, int data) {
    const int pr[2] = {(addr >> 2) & 63, data};
    const Xchg e = xchg(pr, 8);
    const int me = lane_ids().lane;
    int out = 0;
    for (int l = 0; l < 64; ++l)      // (several senders to one lane: the highest lane wins, as the hardware's in-order write does)
        if (lane_on(e, l)) { int q[2]; memcpy(q, e.tab + XS * l, 8); if (q[0] == me) out = q[1]; }
    return out;
}

// ---- matrix instructions (fp32 accumulate in reduction-index order) ----
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// v_mfma_f32_32x32x2_f32: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]; D reg r of lane l = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]
__attribute__((always_inline)) static inline v16f mfma_32x32x2f32(float a, float b, v16f c, int, int, int) {
    const float ab[2] = {a, b};
    const Xchg e = xchg(ab, 8);
    const int l = lane_ids().lane, j = l & 31;
    if (l == __builtin_ctzll(e.mask)) counters[C_MFMA_32x32x2_F32]++;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float A[2], B[2];
            memcpy(A, e.tab + XS * (32 * k + i), 8); memcpy(B, e.tab + XS * (32 * k + j), 8);
            acc = fmaf(A[0], B[1], acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; D reg r of lane l = D[4 (l >> 4) + r][l & 15]
__attribute__((always_inline)) static inline v4f mfma_16x16x4f32(float a, float b, v4f c, int, int, int) {
    const float ab[2] = {a, b};
    const Xchg e = xchg(ab, 8);
    const int l = lane_ids().lane, j = l & 15;
    if (l == __builtin_ctzll(e.mask)) counters[C_MFMA_16x16x4_F32]++;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float A[2], B[2];
            memcpy(A, e.tab + XS * (16 * k + i), 8); memcpy(B, e.tab + XS * (16 * k + j), 8);
            acc = fmaf(A[0], B[1], acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products; D reg i of lane l = A[lane 4 (l / 4) + i] * B[lane l] (tools/probes/mfma4_probe.hip)
__attribute__((always_inline)) static inline v4f mfma_4x4x1f32(float a, float b, v4f c, int, int, int) {
    const float ab[2] = {a, b};
    const Xchg e = xchg(ab, 8);
    const int l = lane_ids().lane;
    if (l == __builtin_ctzll(e.mask)) counters[C_MFMA_4x4x1_F32]++;
    for (int i = 0; i < 4; ++i) {
        float A[2];
        memcpy(A, e.tab + XS * (4 * (l / 4) + i), 8);
        c[i] = fmaf(A[0], b, c[i]);
    }
    return c;
}
static inline float bf(const unsigned char* p, int e) {
    unsigned short h; memcpy(&h, p + 2 * e, 2);
    return __uint_as_float((unsigned)h << 16);
}
// v_mfma_f32_16x16x32_bf16: A[i = l & 15][k = 8 (l >> 4) + e], B[k = 8 (l >> 4) + e][j = l & 15]; D as 16x16x4
__attribute__((always_inline)) static inline v4f mfma_16x16x32_bf16(v8bf a, v8bf b, v4f c, int, int, int) {
    unsigned char ab[32];
    memcpy(ab, &a, 16); memcpy(ab + 16, &b, 16);
    const Xchg e = xchg(ab, 32);
    const int l = lane_ids().lane, j = l & 15;
    if (l == __builtin_ctzll(e.mask)) counters[C_MFMA_16x16x32_BF16]++;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        // (bf16 products are exact; the instruction's 32-term dot product is modelled as exact with ONE rounding into the accumulator --
        // rounding after every term, as an fp32 fma chain would, doubles the error of the bf16 x 6 forms against what the GPU tests measure)
        double acc = 0.0;
        for (int g = 0; g < 4; ++g)
            for (int x = 0; x < 8; ++x) acc += (double)bf(e.tab + XS * (16 * g + i), x) * (double)bf(e.tab + XS * (16 * g + j) + 16, x);
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}
// v_mfma_f32_32x32x16_bf16: A[i = l & 31][k = 8 (l >> 5) + e], B likewise; D as 32x32x2
__attribute__((always_inline)) static inline v16f mfma_32x32x16_bf16(v8bf a, v8bf b, v16f c, int, int, int) {
    unsigned char ab[32];
    memcpy(ab, &a, 16); memcpy(ab + 16, &b, 16);
    const Xchg e = xchg(ab, 32);
    const int l = lane_ids().lane, j = l & 31;
    if (l == __builtin_ctzll(e.mask)) counters[C_MFMA_32x32x16_BF16]++;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = 0.0;
        for (int g = 0; g < 2; ++g)
            for (int x = 0; x < 8; ++x) acc += (double)bf(e.tab + XS * (32 * g + i), x) * (double)bf(e.tab + XS * (32 * g + j) + 16, x);
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}

// ---- raw buffer accesses: a dword whose byte offset reaches past num_records is dropped (loads return 0) ----
static inline rsrc make_rsrc(void* base, int /*stride*/, int num_records, int /*flags*/) { return rsrc{(char*)base, (long)(unsigned)num_records}; }
template <int NDW>
static inline void buf_load(rsrc rs, int voff, int soff, unsigned* out) {
    const long o = (long)(unsigned)voff + (long)(unsigned)soff;
    for (int d = 0; d < NDW; ++d) {
        out[d] = 0;
        if (o + 4 * d + 4 <= rs.bytes) { memcpy(&out[d], rs.base + o + 4 * d, 4); counters[C_BUF_LOAD_BYTES] += 4; }
    }
}
template <int NDW>
static inline void buf_store(rsrc rs, int voff, int soff, const unsigned* in) {
    const long o = (long)(unsigned)voff + (long)(unsigned)soff;
    for (int d = 0; d < NDW; ++d)
        if (o + 4 * d + 4 <= rs.bytes) { memcpy(rs.base + o + 4 * d, &in[d], 4); counters[C_BUF_STORE_BYTES] += 4; }
}
typedef unsigned u32x4_gcc __attribute__((vector_size(16)));
typedef unsigned u32x2_gcc __attribute__((vector_size(8)));
static inline unsigned raw_buffer_load_b32(rsrc rs, int voff, int soff, int) { unsigned v; buf_load<1>(rs, voff, soff, &v); return v; }
static inline u32x2_gcc raw_buffer_load_b64(rsrc rs, int voff, int soff, int) { unsigned v[2]; buf_load<2>(rs, voff, soff, v); u32x2_gcc r; memcpy(&r, v, 8); return r; }
static inline u32x4_gcc raw_buffer_load_b128(rsrc rs, int voff, int soff, int) { unsigned v[4]; buf_load<4>(rs, voff, soff, v); u32x4_gcc r; memcpy(&r, v, 16); return r; }
static inline void raw_buffer_store_b32(unsigned v, rsrc rs, int voff, int soff, int) { buf_store<1>(rs, voff, soff, &v); }
template <class V>
static inline void raw_buffer_store_b128(V v, rsrc rs, int voff, int soff, int) { static_assert(sizeof(V) == 16, ""); unsigned u[4]; memcpy(u, &v, 16); buf_store<4>(rs, voff, soff, u); }

}  // namespace emu

#define __amdgpu_buffer_rsrc_t emu::rsrc
#define __builtin_amdgcn_make_buffer_rsrc emu::make_rsrc
#define __builtin_amdgcn_raw_buffer_load_b32 emu::raw_buffer_load_b32
#define __builtin_amdgcn_raw_buffer_load_b64 emu::raw_buffer_load_b64
#define __builtin_amdgcn_raw_buffer_load_b128 emu::raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_store_b32 emu::raw_buffer_store_b32
#define __builtin_amdgcn_raw_buffer_store_b128 emu::raw_buffer_store_b128
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu::mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu::mfma_16x16x4f32
#define __builtin_amdgcn_mfma_f32_4x4x1f32 emu::mfma_4x4x1f32
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu::mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu::mfma_32x32x16_bf16
#define __builtin_amdgcn_readfirstlane emu::readfirstlane
#define __builtin_amdgcn_readlane emu::readlane
#define __builtin_amdgcn_update_dpp emu::update_dpp
#define __builtin_amdgcn_ds_bpermute emu::ds_bpermute
#define __builtin_amdgcn_ds_permute emu::ds_permute

template <class T> __attribute__((always_inline)) static inline T __shfl_xor(T v, int m, int w = 64) { (void)w; return emu::shfl_from(v, emu::lane_ids().lane ^ m); }
template <class T> __attribute__((always_inline)) static inline T __shfl(T v, int src, int w = 64) { const int l = emu::lane_ids().lane; return emu::shfl_from(v, (l & ~(w - 1)) + (src & (w - 1))); }
template <class T> __attribute__((always_inline)) static inline T __shfl_up(T v, int d, int w = 64) { const int l = emu::lane_ids().lane; return emu::shfl_from(v, (l & (w - 1)) >= d ? l - d : l); }
template <class T> __attribute__((always_inline)) static inline T __shfl_down(T v, int d, int w = 64) { const int l = emu::lane_ids().lane; return emu::shfl_from(v, (l & (w - 1)) + d < w ? l + d : l); }
__attribute__((always_inline)) static inline unsigned long long __ballot(int p) { return emu::ballot(p != 0); }
__attribute__((always_inline)) static inline int __any(int p) { return emu::ballot(p != 0) != 0; }
__attribute__((always_inline)) static inline int __all(int p) { const emu::Xchg e = emu::xchg(&p, 4); for (int l = 0; l < 64; ++l) if (emu::lane_on(e, l) && !emu::lane_get<int>(e, l)) return 0; return 1; }

// ---- atomics / fences ----
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned o = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        const unsigned n = __float_as_uint(__uint_as_float(o) + v);
        if (__atomic_compare_exchange_n(u, &o, n, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return __uint_as_float(o);
    }
}
static inline int atomicMax(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline int atomicCAS(int* p, int c, int v) { __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return c; }

// ---- host runtime API (emu_rt.cpp): streams are ordered queues that execute at once, in host program order ----
typedef int hipError_t;
enum : int {
    hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801, hipErrorNotReady = 600,
};
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
typedef int hipDeviceAttribute_t;
typedef int hipFuncAttribute;
typedef int hipMemcpyKind;
typedef int hipStreamCaptureStatus;
typedef int hipStreamCaptureMode;
typedef struct emuGraph* hipGraph_t;
struct hipIpcMemHandle_t { char reserved[64]; };
enum : int {
    hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4,
    hipDeviceAttributeMultiprocessorCount = 63, hipFuncAttributeMaxDynamicSharedMemorySize = 8,
    hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1,
    hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocMapped = 2, hipHostMallocDefault = 0,
    hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1,
};
#define HIP_SYMBOL(x) (&(x))
#define HIP_KERNEL_NAME(...) __VA_ARGS__

const char* hipGetErrorString(hipError_t);
hipError_t hipGetLastError();
hipError_t hipGetDevice(int*);
hipError_t hipSetDevice(int);
hipError_t hipDeviceGetAttribute(int*, hipDeviceAttribute_t, int);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetStreamPriorityRange(int*, int*);
hipError_t hipMalloc(void**, size_t);
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
hipError_t hipExtMallocWithFlags(void**, size_t, unsigned);
hipError_t hipFree(void*);
hipError_t hipHostMalloc(void**, size_t, unsigned = 0);
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
hipError_t hipHostFree(void*);
hipError_t hipHostGetDevicePointer(void**, void*, unsigned);
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t = nullptr);
hipError_t hipMemcpyFromSymbol(void*, const void*, size_t, size_t = 0, hipMemcpyKind = hipMemcpyDeviceToHost);
hipError_t hipMemcpyToSymbol(const void*, const void*, size_t, size_t = 0, hipMemcpyKind = hipMemcpyHostToDevice);
hipError_t hipMemcpy2DAsync(void*, size_t, const void*, size_t, size_t, size_t, hipMemcpyKind, hipStream_t = nullptr);
hipError_t hipMemset(void*, int, size_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t = nullptr);
hipError_t hipStreamCreateWithPriority(hipStream_t*, unsigned, int);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamCreate(hipStream_t*);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0);
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus*);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventQuery(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int);
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*);
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned);
hipError_t hipIpcCloseMemHandle(void*);
template <class F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 2; return hipSuccess; }

namespace emu {
struct KernelCall { void (*run)(void*); void* ctx; };
void launch(dim3 grid, dim3 block, size_t shmem, const char* name, const void* fn, KernelCall call);

template <class F, class... A>
struct Bound {
    F f; std::tuple<std::decay_t<A>...> args;
    static void run(void* self) { Bound* b = static_cast<Bound*>(self); std::apply(b->f, b->args); }
};
template <class F, class... A>
static inline void launch_bound(dim3 grid, dim3 block, size_t shmem, const char* name, F f, A&&... a) {
    Bound<F, A...> b{f, std::tuple<std::decay_t<A>...>(std::forward<A>(a)...)};
    launch(grid, block, shmem, name, reinterpret_cast<const void*>(f), KernelCall{&Bound<F, A...>::run, &b});
}
}  // namespace emu

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch_bound(dim3(grid), dim3(block), (size_t)(shmem), #kern, kern, ##__VA_ARGS__)
