// Persistent GRU (nn.GRUCell unrolled over the episode) forward + BPTT for gfx950.
//
// Restates the hot loop of EntityAttentionRNNAgent.forward (reference:
// src/modules/agents/entity_rnn_agent.py:49-55): T1 dependent GRUCell steps on [G*B*na, H] rows.
// The reference issues 81 cell launches forward (+81 backward); here ONE kernel walks all steps:
//
//  * rows are independent, so a workgroup owns a tile of FOUR rows for the whole episode -- no grid sync. The input
//    projection x_t W_ih^T + b_ih for ALL steps is one big GEMM done beforehand; only the recurrent h W_hh^T stays in the loop;
//  * the recurrent product runs on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction): the chain
//    of T1 dependent steps is bound by the time of ONE step on ONE CU, and a 4-row tile needs a quarter of the matrix-core
//    cycles per step of the 16-row tile v_mfma_f32_16x16x4_f32 would force (details at the kernels below);
//  * W_hh fragments (3 x H/4 VGPRs per lane) are loaded ONCE and stay in registers for all steps; backward keeps the
//    fragments of W_hh^T the same way;
//  * the new hidden tile is exchanged between the waves through a double-buffered LDS tile, one LDS-only barrier per step
//    (common.h: lds_barrier -- __syncthreads() would drain the step's global stores at every step);
//  * the per-step inputs (forward: input gates; backward: six saved tensors) are fetched REFIL_GRU_PD steps ahead into a
//    ring of registers.
#include <stdlib.h>

#include "bufops.h"
#include "common.h"
#include "kernels.h"
#include "profile.h"
#include "../../include/refil_hip.h"

namespace refil {

// Hidden size GH (rnn_hidden_dim) is a template parameter: 32, 64 (every shipped config) or 128; a workgroup has GH/16 waves.
// LDS pitches: h tile GH + 4, dgh tile 3 GH + 4 (pitch/4 odd: conflict-free 16-byte fragment reads)

struct GruK {
    const float* gi; float* hsx; const float* w_hh; const float* b_hh;
    float* save_r; float* save_z; float* save_n; float* save_ghn;
    const float* dhs; float* dgi; float* dgh;
    int NR, T1, na;
    const int* t_last; int B;     // optional: episode b = gb % B only needs steps t <= t_last[b]
    int zero_h0;                  // forward: h_0 = 0, slot 0 of hsx is written here (no separate fill launch)
    const uint8_t* ever;          // optional [B, na]: rows of never-active agents move no data (loads alias row 0, no stores)
};
// two independent recurrences (the live and the target agent's) in ONE launch: the first nblk0 workgroups run `a`,
// the others `b` -- the two 100 us latency chains overlap instead of queueing behind each other
struct GruK2 { GruK a, b; int nblk0; };

// Global accesses of the step loops: scalar (SGPR) base + 32-bit byte offset per lane -- one instruction per access
// instead of a 64-bit multiply-add chain in front of each (the launchers check that the tensors stay below 4 GiB).
__device__ inline float ldg32(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ inline void stg32(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// Stores of the 4-row step loops: buffer instructions, predicated by the OFFSET (lanes that own no row get an offset past
// num_records: the store is dropped by the range check) instead of a branch. A branch around a step's stores gives the step
// two paths with different numbers of vector-memory operations; the compiler then sizes every s_waitcnt of the input ring
// for the path WITHOUT stores, and on the path with stores that count also drains the stores of the step before -- their
// write latency lands on every step of the recurrence.
constexpr unsigned GRU_BUF_ALL = 0xffff0000u, GRU_BUF_DROP = 0xffff0000u;    // (tensors < 4.0e9 bytes: checked by the launchers; DROP + small immediate offsets do not wrap)
__device__ inline rsrc_t gru_rsrc(const void* base, bool live) {
    // (the descriptor is chosen per workgroup -- GruK2 -- so the compiler cannot see that the base is wave-uniform, and a
    // divergent resource turns every store into a waterfall loop)
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    void* ub = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    const int n = __builtin_amdgcn_readfirstlane(live ? (int)GRU_BUF_ALL : 0);
    return __builtin_amdgcn_make_buffer_rsrc(ub, 0, n, 0x00020000);
}
__device__ inline void stb32(rsrc_t rs, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)byte_off, 0, 0);
}

// ================================================================================================
// 4-row tiles on v_mfma_f32_4x4x1_16b_f32.
//
// A recurrence is a chain of T1 dependent steps per row tile; what bounds it is the time of ONE step on ONE CU, not
// throughput (128 sixteen-row tiles leave half of the 256 CUs idle). With v_mfma_f32_16x16x4_f32 a tile cannot have fewer
// than 16 rows, i.e. 48 MFMAs (1536 cycles) per step and SIMD. v_mfma_f32_4x4x1_16b_f32 computes 16 independent 4x4 outer
// products per instruction at the same FLOP rate (8 cycles of the pipe with two waves feeding a SIMD -- tools/probes/mfma4_probe.hip: layout
// and rate; ONE wave issues them every ~20 cycles whatever the accumulator chains: profiles/r05_gru_timing.txt), so a tile of FOUR
// rows works: block (cg, ks) = lane / 4 of wave w multiplies the 4 rows (A: lane % 4 = row) by the 4 hidden columns
// 16 w + 4 cg + (lane % 4) over the k slice ks (a quarter of the reduction: the k order is free), 3 gates x H/4 instructions
// = 384 cycles per step instead of 1536, on four times as many workgroups (every CU gets two). The four k-slice partials of
// an element sit in the four quads of a DPP row: two row_ror adds. Afterwards lane (cg, ks, j) OWNS element (row ks, column
// 16 w + 4 cg + j): a quarter of the gate arithmetic, of the per-step loads and of the stores of the 16-row kernels per lane.
// ================================================================================================
constexpr int GR4 = 4;
#ifndef REFIL_GRU_PD
#define REFIL_GRU_PD 4      // prefetch distance (steps) of the 4-row recurrence kernels = unroll factor of their step loops
#endif
// a tile whose four rows all belong to agents that are never active in their episode has nothing to do: its rows are neither
// read nor written by anyone (ListArgs::ever); with 16-row tiles (all agents of an episode copy) this never happened
__device__ inline bool tile_never_active(const GruK& p, int r0) {
    if (!p.ever) return false;
    bool any = false;
    for (int k = 0; k < GR4; ++k) {
        const int rr = r0 + k;
        if (rr < p.NR) any |= p.ever[((rr / p.na) % p.B) * p.na + rr % p.na] != 0;
    }
    return !any;
}

__device__ inline int tile_steps4(const GruK& p, int r0) {
    if (!p.t_last) return p.T1;
    int tend = 0;
    const int rl = min(r0 + GR4, p.NR) - 1;
    for (int gb = r0 / p.na; gb <= rl / p.na; ++gb) tend = max(tend, p.t_last[gb % p.B] + 1);
    return min(tend, p.T1);
}
template <int CTRL>
__device__ inline float dpp_rot(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the four quads of each 16-lane row (the four k slices of a block column), in every lane
__device__ inline float quad4_sum(float v) { v += dpp_rot<0x124>(v); v += dpp_rot<0x128>(v); return v; }   // row_ror:4, row_ror:8
__device__ inline float pick4(const f32x4& v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3])); }
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#ifndef GRU_NACC
#define GRU_NACC 2          // accumulators per gate in the 4-row forward recurrence (2 or 4)
#endif
#ifndef GRU_NACC_BWD
#define GRU_NACC_BWD 3      // accumulators of the 4-row backward recurrence (3, 6 or 12)
#endif

// VALU: the recurrent product on the vector ALUs (packed FMAs) instead of the matrix cores -- see gru_valu_enabled()
#ifdef REFIL_GRU_TIMING
// debug build only (REFIL_EXTRA_FLAGS=-DREFIL_GRU_TIMING; tools/probes/gru_timing.py): cycle sums (shader clock) of a step's phases in
// workgroup 0 / wave 0 of the LAST gru_fwd4 launch: [0] wait for the step's inputs, [1] LDS reads + recurrent product, [2] k-slice sums,
// [3] gates, [4] stores + barrier, [5] steps
__device__ unsigned long long g_gru_dbg[8];
#define GRU_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; } while (0)
#else
#define GRU_STAMP(i) do { } while (0)
#endif
template <bool SAVE, int GH, bool VALU = false, int PD_ = REFIL_GRU_PD>
__global__ __launch_bounds__(4 * GH) void gru_fwd4_kernel(GruK2 p2) {
    constexpr int HP = GH + 4, KS = VALU ? GH : GH / 4;    // KS: reduction indices per lane (MFMA: per k slice)
    __shared__ __attribute__((aligned(16))) float hbuf[2][GR4 * HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 3, ks = (lane >> 2) & 3, cg = lane >> 4;
    const int c = wave * 16 + 4 * cg + j;                  // hidden column of this lane (B operand / result column)
    const bool second = (int)blockIdx.x >= p2.nblk0;
    const GruK& p = second ? p2.b : p2.a;
    const int r0 = (second ? blockIdx.x - p2.nblk0 : blockIdx.x) * GR4;
    if (tile_never_active(p, r0)) return;                 // (uniform)
    const int tend = tile_steps4(p, r0);
    const bool save = SAVE && p.save_r != nullptr;

    // W_hh fragments of this lane's column over its k slice: bw[g][s] = W_hh[g GH + c][KS ks + s]
    float bw[3][KS];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4) {
            const float4 v = *reinterpret_cast<const float4*>(p.w_hh + (long)(g * GH + c) * GH + (VALU ? 0 : KS * ks) + 4 * s4);
            bw[g][4 * s4] = v.x; bw[g][4 * s4 + 1] = v.y; bw[g][4 * s4 + 2] = v.z; bw[g][4 * s4 + 3] = v.w;
        }
    const float bhr = p.b_hh[c], bhz = p.b_hh[GH + c], bhn = p.b_hh[2 * GH + c];

    // the element this lane owns: row r0 + ks, column c
    const int rr = r0 + ks;
    bool valid = rr < p.NR;
    if (valid && p.ever) valid = p.ever[((rr / p.na) % p.B) * p.na + rr % p.na] != 0;      // (never-active agent: no traffic)
    const int gb = valid ? rr / p.na : 0, ia = valid ? rr % p.na : 0;
    const long gi_base = (long)gb * p.T1 * p.na + ia, hs_base = (long)gb * (p.T1 + 1) * p.na + ia;
    float hold = (valid && !p.zero_h0) ? p.hsx[hs_base * GH + c] : 0.f;
    if (valid && p.zero_h0) p.hsx[hs_base * GH + c] = 0.f;       // (the backward and the weight gradients read h_{-1} from slot 0)
    hbuf[0][ks * HP + c] = hold;
    // byte offsets at step 0 (rows past the end / never-active rows alias row 0: loaded, never stored)
    const unsigned go = (unsigned)((gi_base * (3 * GH) + c) * sizeof(float));
    const unsigned gi_step = (unsigned)(p.na * 3 * GH * sizeof(float)), row_step = (unsigned)(p.na * GH * sizeof(float));
    // store offsets: lanes without a row keep the dropped offset for the whole loop (their step is 0)
    unsigned so = valid ? (unsigned)((gi_base * GH + c) * sizeof(float)) : GRU_BUF_DROP;
    unsigned ho = valid ? (unsigned)(((hs_base + p.na) * GH + c) * sizeof(float)) : GRU_BUF_DROP;
    const unsigned st_step = valid ? row_step : 0u;
    const rsrc_t rs_h = gru_rsrc(p.hsx, true);
    const rsrc_t rs_r = gru_rsrc(p.save_r, save), rs_z = gru_rsrc(p.save_z, save), rs_n = gru_rsrc(p.save_n, save),
                 rs_g = gru_rsrc(p.save_ghn, save), rs_none = gru_rsrc(p.hsx, false);
    // The per-step inputs are fetched PD steps ahead into a ring of registers (the loop is unrolled by PD: static ring
    // indices, no moves of in-flight registers). gfx9 retires loads and stores through ONE in-order counter: awaiting the
    // load of step t + PD also awaits every store issued before it, so the distance is what gives the step's own stores
    // (and the HBM latency under load, > 1 us) PD steps to complete instead of one.
    constexpr int PD = PD_;
    float gq[PD][3];
    const int tlast = max(tend - 1, 0);
#pragma unroll
    for (int u = 0; u < PD; ++u) {
#pragma unroll
        for (int g = 0; g < 3; ++g) gq[u][g] = ldg32(p.gi, go + (unsigned)min(u, tlast) * gi_step + g * GH * (unsigned)sizeof(float));
        // as many (dropped) stores as a step issues behind its loads: the way into the loop then looks like the back edge, and the
        // compiler -- which sizes a wait for the more conservative of the paths into the loop head -- counts a full ring on both
#pragma unroll
        for (int k = 0; k < (SAVE ? 5 : 1); ++k) stb32(rs_none, 64u * (u * 5 + k), 0.f);    // (apart: adjacent ones would be merged into one wide store)
        __builtin_amdgcn_sched_barrier(0);     // (issue order = ring order: slot u has the later slots' operations behind it on the way into the loop too)
    }
    lds_barrier();

    // The trip count is rounded up to whole ring periods: a step past the end issues the SAME vector-memory operations (its
    // loads re-read the last step, its stores carry the dropped offset) and skips the arithmetic behind a uniform branch that
    // contains none -- one back edge, no exit from the middle of the unrolled body, every path with the same operation count:
    // the s_waitcnt in front of a slot's use then counts exactly the PD steps of loads and stores issued behind it.
#ifdef REFIL_GRU_TIMING
    unsigned long long dbg_acc[6] = {0, 0, 0, 0, 0, 0}, dbg_t = __builtin_readcyclecounter();
#endif
    for (int t0 = 0; t0 < tend; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = t0 + u;
            // (uniform. Opaque to the compiler: the first copy of the unrolled body is always live, and with that knowledge its
            // arithmetic is contracted into different fused multiply-adds than the other copies' -- a step's result must not
            // depend on its position in the ring: tests/test_gpu_ops.py::test_gru_time_bounds compares bit for bit)
            int t_op = t;
            asm volatile("" : "+s"(t_op));
            const bool live = t_op < tend;
            float gcur[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) { asm volatile("" : "+v"(gq[u][g])); gcur[g] = gq[u][g]; }
            GRU_STAMP(0);
            {   // refill the slot with step t + PD (past the end: the last step again -- no branch around the loads)
                const unsigned o = go + (unsigned)min(t + PD, tlast) * gi_step;
#pragma unroll
                for (int g = 0; g < 3; ++g) gq[u][g] = ldg32(p.gi, o + g * GH * (unsigned)sizeof(float));
            }
            __builtin_amdgcn_sched_barrier(0);             // the loads are ISSUED here
            float rg = 0.f, zg = 0.f, gh = 0.f, ng = 0.f;
            if (live) {
                const float* hb = hbuf[t & 1];
                float* hn = hbuf[(t + 1) & 1];
                float pre[3];
                if (VALU) {
                    // the lane's own element over the whole reduction: row ks of h (an LDS broadcast per 16 bytes) against its
                    // column of W_hh in registers, packed FMAs, two accumulator pairs per gate
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 acc[3][2];
#pragma unroll
                    for (int g = 0; g < 3; ++g) { acc[g][0] = f32x2{0.f, 0.f}; acc[g][1] = acc[g][0]; }
#pragma unroll
                    for (int s4 = 0; s4 < GH / 4; ++s4) {
                        const float4 v = *reinterpret_cast<const float4*>(hb + ks * HP + 4 * s4);
                        const f32x2 h0 = {v.x, v.y}, h1 = {v.z, v.w};
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            acc[g][0] = __builtin_elementwise_fma(h0, f32x2{bw[g][4 * s4], bw[g][4 * s4 + 1]}, acc[g][0]);
                            acc[g][1] = __builtin_elementwise_fma(h1, f32x2{bw[g][4 * s4 + 2], bw[g][4 * s4 + 3]}, acc[g][1]);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 3; ++g) { const f32x2 t2 = acc[g][0] + acc[g][1]; pre[g] = t2[0] + t2[1]; }
                } else {
                float a[KS];                               // A operand: row j of the h tile over this lane's k slice
#pragma unroll
                for (int s4 = 0; s4 < KS / 4; ++s4) {
                    const float4 v = *reinterpret_cast<const float4*>(hb + j * HP + KS * ks + 4 * s4);
                    a[4 * s4] = v.x; a[4 * s4 + 1] = v.y; a[4 * s4 + 2] = v.z; a[4 * s4 + 3] = v.w;
                }
                f32x4 acc[3][GRU_NACC];                    // GRU_NACC accumulators per gate: dependent chains that much shorter
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < GRU_NACC; ++i) acc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][s % GRU_NACC] = MFMA4(a[s], bw[g][s], acc[g][s % GRU_NACC]);
#ifdef REFIL_GRU_TIMING
                asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][GRU_NACC - 1][0]), "v"(acc[2][0][0]), "v"(acc[2][GRU_NACC - 1][0]));
                GRU_STAMP(1);
#endif
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    f32x4 v = acc[g][0] + acc[g][1];
                    if (GRU_NACC == 4) v = v + (acc[g][2] + acc[g][3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = quad4_sum(v[i]);
                    pre[g] = pick4(v, ks);                 // (row ks, column c): the element this lane owns
                }
#ifdef REFIL_GRU_TIMING
                asm volatile("s_nop 0" :: "v"(pre[0]), "v"(pre[1]), "v"(pre[2]));
                GRU_STAMP(2);
#endif
                }
                rg = fast_sigmoid(gcur[0] + pre[0] + bhr);
                zg = fast_sigmoid(gcur[1] + pre[1] + bhz);
                gh = pre[2] + bhn;
                ng = fast_tanh(gcur[2] + rg * gh);
                hold = (1.0f - zg) * ng + zg * hold;
                hn[ks * HP + c] = hold;
#ifdef REFIL_GRU_TIMING
                asm volatile("s_nop 0" :: "v"(hold));
                GRU_STAMP(3);
#endif
            }
            const unsigned hod = live ? ho : GRU_BUF_DROP, sod = live ? so : GRU_BUF_DROP;
            stb32(rs_h, hod, hold);
            if (SAVE) { stb32(rs_r, sod, rg); stb32(rs_z, sod, zg); stb32(rs_n, sod, ng); stb32(rs_g, sod, gh); }   // (no save buffers: zero-sized resources)
            ho += st_step; so += st_step;
            if (live) lds_barrier();                       // (not __syncthreads(): its fence would drain the step's stores -- common.h)
#ifdef REFIL_GRU_TIMING
            GRU_STAMP(4);
            dbg_acc[5] += live ? 1 : 0;
#endif
        }
    }
#ifdef REFIL_GRU_TIMING
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 6; ++i) g_gru_dbg[i] = dbg_acc[i];
#endif
}

// BPTT on 4-row tiles: lane (cg, ks, j) owns element (row ks, column c) of the tile. For t = T1-1 .. 0:
//   dh = carry + dhs[t];  dn = dh (1-z);  dz = dh (h_{t-1} - n);  carry' = dh z
//   dn_pre = dn (1-n^2);  dr = dn_pre ghn;  dgh_n = dn_pre r
//   dr_pre = dr r (1-r);  dz_pre = dz z (1-z)
//   dgi[t] = (dr_pre, dz_pre, dn_pre);  d(gh)[t] = (dr_pre, dz_pre, dgh_n) -- only dgh_n is STORED (p.dgh: [rows, GH]; the r / z
//   blocks are dgi's: 41 MB of the 122 MB the recurrence wrote per launch at cfg-T);  carry' += d(gh)[t] W_hh
template <int GH, bool VALU = false, int PD_ = REFIL_GRU_PD>
__global__ __launch_bounds__(4 * GH) void gru_bwd4_kernel(GruK p) {
    constexpr int GP = 3 * GH + 4, KS = VALU ? 3 * GH : 3 * GH / 4, NW = GH / 16;
    __shared__ __attribute__((aligned(16))) float gbuf[2][GR4 * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 3, ks = (lane >> 2) & 3, cg = lane >> 4;
    const int c = wave * 16 + 4 * cg + j;
    const int r0 = blockIdx.x * GR4;
    if (tile_never_active(p, r0)) return;                 // (uniform)
    const int tend = tile_steps4(p, r0);

    // carry[row][c] += sum_k dgh[row][k] W_hh[k][c]: bw[s] = W_hh[KS ks + s][c]
    float bw[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) bw[s] = p.w_hh[(long)((VALU ? 0 : KS * ks) + s) * GH + c];

    const int rr = r0 + ks;
    bool valid = rr < p.NR;
    if (valid && p.ever) valid = p.ever[((rr / p.na) % p.B) * p.na + rr % p.na] != 0;
    const int gb = valid ? rr / p.na : 0, ia = valid ? rr % p.na : 0;
    const long gi_base = (long)gb * p.T1 * p.na + ia, hs_base = (long)gb * (p.T1 + 1) * p.na + ia;
    const unsigned go = (unsigned)((gi_base * GH + c) * sizeof(float));
    const unsigned ho = (unsigned)((hs_base * GH + c) * sizeof(float));
    const unsigned so = (unsigned)((gi_base * (3 * GH) + c) * sizeof(float));
    const unsigned row_step = (unsigned)(p.na * GH * sizeof(float));
    // per-step inputs of the own element: 0 dhs, 1 r, 2 z, 3 n, 4 ghn, 5 h_{t-1}
    auto fetch = [&](float (&dst)[6], int t) __attribute__((always_inline)) {
        const unsigned adv = (unsigned)t * row_step;
        dst[0] = ldg32(p.dhs, go + adv); dst[1] = ldg32(p.save_r, go + adv); dst[2] = ldg32(p.save_z, go + adv);
        dst[3] = ldg32(p.save_n, go + adv); dst[4] = ldg32(p.save_ghn, go + adv); dst[5] = ldg32(p.hsx, ho + adv);
    };
    // steps the episode's loss cannot reach (t >= tend): exact zeros, what the full recurrence would have produced
    for (int t = tend + wave; t < p.T1; t += NW) {
        for (int idx = lane; idx < GR4 * (3 * GH / 4); idx += 64) {
            const int row = idx / (3 * GH / 4), c4 = idx % (3 * GH / 4);
            const int r2 = r0 + row;
            if (r2 < p.NR && (!p.ever || p.ever[((r2 / p.na) % p.B) * p.na + r2 % p.na])) {
                const long orow = ((long)(r2 / p.na) * p.T1 + t) * p.na + r2 % p.na;
                *reinterpret_cast<float4*>(p.dgi + orow * (3 * GH) + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c4 < GH / 4) *reinterpret_cast<float4*>(p.dgh + orow * GH + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (tend <= 0) return;
    // inputs fetched PD steps ahead into a ring of register sets (loop unrolled by PD: static indices; see the forward kernel)
    constexpr int PD = PD_;
    float ring[PD][6];
    float carry = 0.f;
    const rsrc_t rs_gi = gru_rsrc(p.dgi, true), rs_gh = gru_rsrc(p.dgh, true), rs_none = gru_rsrc(p.dgi, false);
#pragma unroll
    for (int u = 0; u < PD; ++u) {                         // (issue order = ring order; dropped stores: see the forward kernel)
        fetch(ring[u], max(tend - 1 - u, 0));
#pragma unroll
        for (int k = 0; k < 4; ++k) stb32(rs_none, 64u * (u * 4 + k), 0.f);
        __builtin_amdgcn_sched_barrier(0);
    }
    int it = 0;
    // (whole ring periods, dead steps past t = 0 with the same vector-memory operations: see the forward kernel)
    for (int tb = tend - 1; tb >= 0; tb -= PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = tb - u;
            int t_op = t;                                  // (uniform; opaque: see the forward kernel)
            asm volatile("" : "+s"(t_op));
            const bool live = t_op >= 0;
            float cur[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { asm volatile("" : "+v"(ring[u][k])); cur[k] = ring[u][k]; }
            fetch(ring[u], max(t - PD, 0));                // (unconditional: a branch here costs register copies and waits)
            __builtin_amdgcn_sched_barrier(0);             // the loads are ISSUED here, ahead of the step's arithmetic
            float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dghn = 0.f;
            if (live) {
                float* gw = gbuf[it & 1];
                ++it;
                const float dh = carry + cur[0];
                const float rg = cur[1], zg = cur[2], ng = cur[3], ghn = cur[4], hp = cur[5];
                const float dn = dh * (1.0f - zg);
                const float dz = dh * (hp - ng);
                const float dhz = dh * zg;
                dn_pre = dn * (1.0f - ng * ng);
                const float dr = dn_pre * ghn;
                dghn = dn_pre * rg;
                dr_pre = dr * rg * (1.0f - rg);
                dz_pre = dz * zg * (1.0f - zg);
                float* row = gw + ks * GP;
                row[c] = dr_pre; row[GH + c] = dz_pre; row[2 * GH + c] = dghn;
                lds_barrier();                             // (not __syncthreads(): its fence would drain the step's loads and stores)
                if (VALU) {
                    // the lane's own carry element over all 3 GH gate gradients of its row (LDS broadcasts) against its column of
                    // W_hh in registers: packed FMAs on the vector ALUs, no matrix-core instruction (gru_valu_enabled())
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 b0 = {0.f, 0.f}, b1 = b0, b2 = b0, b3 = b0;
#pragma unroll
                    for (int s = 0; s < KS; s += 8) {
                        const float4 v0 = *reinterpret_cast<const float4*>(gw + ks * GP + s);
                        const float4 v1 = *reinterpret_cast<const float4*>(gw + ks * GP + s + 4);
                        b0 = __builtin_elementwise_fma(f32x2{v0.x, v0.y}, f32x2{bw[s], bw[s + 1]}, b0);
                        b1 = __builtin_elementwise_fma(f32x2{v0.z, v0.w}, f32x2{bw[s + 2], bw[s + 3]}, b1);
                        b2 = __builtin_elementwise_fma(f32x2{v1.x, v1.y}, f32x2{bw[s + 4], bw[s + 5]}, b2);
                        b3 = __builtin_elementwise_fma(f32x2{v1.z, v1.w}, f32x2{bw[s + 6], bw[s + 7]}, b3);
                    }
                    const f32x2 t2 = (b0 + b1) + (b2 + b3);
                    carry = dhz + (t2[0] + t2[1]);
                } else {
                f32x4 aa[GRU_NACC_BWD];
#pragma unroll
                for (int i = 0; i < GRU_NACC_BWD; ++i) aa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; s += 12) {
                    const float4 v0 = *reinterpret_cast<const float4*>(gw + j * GP + KS * ks + s);
                    const float4 v1 = *reinterpret_cast<const float4*>(gw + j * GP + KS * ks + s + 4);
                    const float4 v2 = *reinterpret_cast<const float4*>(gw + j * GP + KS * ks + s + 8);
                    const float av[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
#pragma unroll
                    for (int e = 0; e < 12; ++e) aa[(s + e) % GRU_NACC_BWD] = MFMA4(av[e], bw[s + e], aa[(s + e) % GRU_NACC_BWD]);
                }
                f32x4 v = aa[0] + aa[1] + aa[2];
                if (GRU_NACC_BWD >= 6) v = v + (aa[3 % GRU_NACC_BWD] + aa[4 % GRU_NACC_BWD] + aa[5 % GRU_NACC_BWD]);
                if (GRU_NACC_BWD == 12) v = v + ((aa[6 % GRU_NACC_BWD] + aa[7 % GRU_NACC_BWD] + aa[8 % GRU_NACC_BWD]) + (aa[9 % GRU_NACC_BWD] + aa[10 % GRU_NACC_BWD] + aa[11 % GRU_NACC_BWD]));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = quad4_sum(v[i]);
                carry = dhz + pick4(v, ks);
                }
            }
            {
                constexpr unsigned G1 = GH * sizeof(float);
                const unsigned o = (valid && live) ? so + 3u * (unsigned)t * row_step : GRU_BUF_DROP;
                // (the r / z blocks of d(gh) equal d(gi)'s: only the n block of d(gh) is stored, in its own [rows, GH] tensor)
                const unsigned oh = (valid && live) ? go + (unsigned)t * row_step : GRU_BUF_DROP;
                stb32(rs_gi, o, dr_pre); stb32(rs_gi, o + G1, dz_pre); stb32(rs_gi, o + 2 * G1, dn_pre);
                stb32(rs_gh, oh, dghn);
            }
        }
    }
}

// ================================================================================================
// 16-row tiles on v_mfma_f32_16x16x4_f32: the variant for LARGE batches. From three 4-row tiles per CU on (B = 64 at the
// north-star shape: three to four per CU) a CU works on 16 rows per step either way, and one 16-row workgroup does
// it with a quarter of the waves, barriers and k-slice reductions (measured at cfg3: 3.245 ms against 3.326 with 4-row tiles;
// cfg2 / cfg4 / cfg-T: 4-row tiles win by 12 / 7 / 1 %). Wave w owns hidden columns [16w, 16w+16) of r, z and n for all 16
// rows; fragment maps: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+reg][col=l&15].
// ================================================================================================
constexpr int GROWS = 16;     // rows per workgroup
// number of steps the rows [r0, r0 + GROWS) need: 1 + max over their episodes of t_last (all T1 without a bound)
__device__ inline int tile_steps16(const GruK& p, int r0) {
    if (!p.t_last) return p.T1;
    int tend = 0;
    const int rl = min(r0 + GROWS, p.NR) - 1;
    for (int gb = r0 / p.na; gb <= rl / p.na; ++gb) tend = max(tend, p.t_last[gb % p.B] + 1);
    return min(tend, p.T1);
}

template <bool SAVE, int GH>
__global__ __launch_bounds__(4 * GH) void gru_fwd16_kernel(GruK2 p2) {
    constexpr int HP = GH + 4, KQ = GH / 4;                // KQ: reduction indices per lane group
    __shared__ __attribute__((aligned(16))) float hbuf[2][GROWS * HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int c = wave * 16 + c16;           // hidden column owned by this lane
    const bool second = (int)blockIdx.x >= p2.nblk0;
    const GruK& p = second ? p2.b : p2.a;
    const int r0 = (second ? blockIdx.x - p2.nblk0 : blockIdx.x) * GROWS;
    const int tend = tile_steps16(p, r0);
    const bool save = SAVE && p.save_r != nullptr;

    // W_hh fragments: bw[g][s] = W_hh[g*GH + c][KQ q + s]. (The MFMA k order is a free permutation as long as both
    // operands agree: giving lane group q the CONTIGUOUS k range [KQ q, KQ q + KQ) turns the scalar LDS reads of the
    // h fragment into ds_read_b128.)
    float bw[3][KQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s = 0; s < KQ; ++s) bw[g][s] = p.w_hh[(long)(g * GH + c) * GH + KQ * q + s];
    const float bhr = p.b_hh[c], bhz = p.b_hh[GH + c], bhn = p.b_hh[2 * GH + c];

    long gi_base[4], hs_base[4];
    bool valid[4];
    float hold[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int rr = r0 + 4 * q + reg;
        valid[reg] = rr < p.NR;
        if (valid[reg] && p.ever) valid[reg] = p.ever[((rr / p.na) % p.B) * p.na + rr % p.na] != 0;      // (never-active agent: no traffic)
        const int gb = valid[reg] ? rr / p.na : 0, i = valid[reg] ? rr % p.na : 0;
        gi_base[reg] = (long)gb * p.T1 * p.na + i;
        hs_base[reg] = (long)gb * (p.T1 + 1) * p.na + i;
        hold[reg] = (valid[reg] && !p.zero_h0) ? p.hsx[hs_base[reg] * GH + c] : 0.f;
        if (valid[reg] && p.zero_h0) p.hsx[hs_base[reg] * GH + c] = 0.f;      // (the backward and the weight gradients read h_{-1} from slot 0)
        hbuf[0][(4 * q + reg) * HP + c] = hold[reg];
    }
    // byte offsets of this lane's four rows at step 0: gi [.., 3 GH], the saved gates [.., GH], hsx [.., GH] (slot t+1);
    // rows past the end alias row 0 of the tensor (loaded, never stored: a row's recurrence depends on that row only)
    unsigned go[4], so[4], ho[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        go[reg] = (unsigned)((gi_base[reg] * (3 * GH) + c) * sizeof(float));
        so[reg] = (unsigned)((gi_base[reg] * GH + c) * sizeof(float));
        ho[reg] = (unsigned)(((hs_base[reg] + p.na) * GH + c) * sizeof(float));
    }
    const unsigned gi_step = (unsigned)(p.na * 3 * GH * sizeof(float)), row_step = (unsigned)(p.na * GH * sizeof(float));
    float gcur[4][3], gnext[4][3];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int g = 0; g < 3; ++g) gcur[reg][g] = ldg32(p.gi, go[reg] + g * GH * (unsigned)sizeof(float));
    __syncthreads();

    for (int t = 0; t < tend; ++t) {
        const float* hb = hbuf[t & 1];
        float* hn = hbuf[(t + 1) & 1];
        {   // next step's gi (the last step re-reads its own: no branch around the loads)
            const unsigned adv = t + 1 < tend ? gi_step : 0u;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                go[reg] += adv;
#pragma unroll
                for (int g = 0; g < 3; ++g) gnext[reg][g] = ldg32(p.gi, go[reg] + g * GH * (unsigned)sizeof(float));
            }
        }
        __builtin_amdgcn_sched_barrier(0);                 // the loads are ISSUED here: they need the whole step to arrive
        float a[KQ];
#pragma unroll
        for (int s4 = 0; s4 < KQ / 4; ++s4) {
            const float4 v = *reinterpret_cast<const float4*>(hb + c16 * HP + KQ * q + 4 * s4);
            a[4 * s4] = v.x; a[4 * s4 + 1] = v.y; a[4 * s4 + 2] = v.z; a[4 * s4 + 3] = v.w;
        }
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
            ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[0][s], ar, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[1][s], az, 0, 0, 0);
            an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw[2][s], an, 0, 0, 0);
        }
        float rgv[4], zgv[4], ngv[4], ghv[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            rgv[reg] = fast_sigmoid(gcur[reg][0] + ar[reg] + bhr);
            zgv[reg] = fast_sigmoid(gcur[reg][1] + az[reg] + bhz);
            ghv[reg] = an[reg] + bhn;
            ngv[reg] = fast_tanh(gcur[reg][2] + rgv[reg] * ghv[reg]);
            const float hnew = (1.0f - zgv[reg]) * ngv[reg] + zgv[reg] * hold[reg];
            hold[reg] = hnew;
            hn[(4 * q + reg) * HP + c] = hnew;
        }
        // gfx9 counts loads and stores in one counter (vmcnt): awaiting the prefetched gi AFTER this step's stores
        // would stall every step on the store acknowledgements. So the prefetch is collected first (it had the whole
        // step to arrive) and the stores are issued last; they drain during the next step.
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                asm volatile("" : "+v"(gnext[reg][g]));
                gcur[reg][g] = gnext[reg][g];
            }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if (valid[reg]) {
                stg32(p.hsx, ho[reg], hold[reg]);
                if (save) {
                    stg32(p.save_r, so[reg], rgv[reg]); stg32(p.save_z, so[reg], zgv[reg]);
                    stg32(p.save_n, so[reg], ngv[reg]); stg32(p.save_ghn, so[reg], ghv[reg]);
                }
            }
            ho[reg] += row_step; so[reg] += row_step;
        }
        lds_barrier();
    }
}

// BPTT. For t = T1-1 .. 0:
//   dh = carry + dhs[t];  dn = dh (1-z);  dz = dh (h_{t-1} - n);  carry' = dh z
//   dn_pre = dn (1-n^2);  dr = dn_pre ghn;  dgh_n = dn_pre r
//   dr_pre = dr r (1-r);  dz_pre = dz z (1-z)
//   dgi[t] = (dr_pre, dz_pre, dn_pre);  d(gh)[t] = (dr_pre, dz_pre, dgh_n) -- only dgh_n is STORED (p.dgh: [rows, GH]; the r / z
//   blocks are dgi's: 41 MB of the 122 MB the recurrence wrote per launch at cfg-T);  carry' += d(gh)[t] W_hh
template <int GH>
__global__ __launch_bounds__(4 * GH) void gru_bwd16_kernel(GruK p) {
    constexpr int GP = 3 * GH + 4, KQ = 3 * GH / 4, NW = GH / 16;
    __shared__ __attribute__((aligned(16))) float gbuf[2][GROWS * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int c = wave * 16 + c16;
    const int r0 = blockIdx.x * GROWS;
    const int tend = tile_steps16(p, r0);

    // W_hh^T fragments: carry[row][c] += sum_k dgh[row][k] W_hh[k][c];  bw[s] = W_hh[KQ q + s][c]  (contiguous k range per lane group, see the forward kernel)
    float bw[KQ];
#pragma unroll
    for (int s = 0; s < KQ; ++s) bw[s] = p.w_hh[(long)(KQ * q + s) * GH + c];

    long gi_base[4], hs_base[4];
    bool valid[4];
    float carry[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int rr = r0 + 4 * q + reg;
        valid[reg] = rr < p.NR;
        if (valid[reg] && p.ever) valid[reg] = p.ever[((rr / p.na) % p.B) * p.na + rr % p.na] != 0;      // (never-active agent: no traffic)
        const int gb = valid[reg] ? rr / p.na : 0, i = valid[reg] ? rr % p.na : 0;
        gi_base[reg] = (long)gb * p.T1 * p.na + i;
        hs_base[reg] = (long)gb * (p.T1 + 1) * p.na + i;
        carry[reg] = 0.f;
    }
    // per-step inputs: 0 dhs, 1 r, 2 z, 3 n, 4 ghn, 5 h_{t-1}
    float cur[4][6];
    // (rows past the end read row r0's values instead of branching around the loads: nothing of theirs is stored, and a
    // row's recurrence depends on nothing but that row)
    unsigned go[4], ho[4], so[4];      // byte offsets at step 0: [.., GH] tensors, hsx (slot t), dgi / dgh [.., 3 GH]
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        go[reg] = (unsigned)((gi_base[reg] * GH + c) * sizeof(float));
        ho[reg] = (unsigned)((hs_base[reg] * GH + c) * sizeof(float));
        so[reg] = (unsigned)((gi_base[reg] * (3 * GH) + c) * sizeof(float));
    }
    const unsigned row_step = (unsigned)(p.na * GH * sizeof(float));
    auto fetch = [&](float (&dst)[4][6], int t) __attribute__((always_inline)) {
        const unsigned adv = (unsigned)t * row_step;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            dst[reg][0] = ldg32(p.dhs, go[reg] + adv);
            dst[reg][1] = ldg32(p.save_r, go[reg] + adv);
            dst[reg][2] = ldg32(p.save_z, go[reg] + adv);
            dst[reg][3] = ldg32(p.save_n, go[reg] + adv);
            dst[reg][4] = ldg32(p.save_ghn, go[reg] + adv);
            dst[reg][5] = ldg32(p.hsx, ho[reg] + adv);
        }
    };
    // steps the episode's loss cannot reach (t >= tend): exact zeros, what the full recurrence would have produced
    for (int t = tend + (tid >> 6); t < p.T1; t += NW) {
        for (int idx = lane; idx < GROWS * (3 * GH / 4); idx += 64) {
            const int row = idx / (3 * GH / 4), c4 = idx % (3 * GH / 4);
            const int rr = r0 + row;
            if (rr < p.NR && (!p.ever || p.ever[((rr / p.na) % p.B) * p.na + rr % p.na])) {
                const long orow = ((long)(rr / p.na) * p.T1 + t) * p.na + rr % p.na;
                *reinterpret_cast<float4*>(p.dgi + orow * (3 * GH) + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c4 < GH / 4) *reinterpret_cast<float4*>(p.dgh + orow * GH + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (tend <= 0) return;
    // The per-step inputs come from HBM: their latency (> 1 us under load) exceeds a step's compute, so they are fetched TWO
    // steps ahead into two alternating register sets (the loop is unrolled by two: no moves of in-flight registers).
    float pa[4][6], pb[4][6] = {};
    fetch(cur, tend - 1);
    fetch(pa, max(tend - 2, 0));
    int it = 0;
    auto step = [&](int t, float (&issue)[4][6], float (&next)[4][6]) __attribute__((always_inline)) {
        float* gb_w = gbuf[it & 1];
        ++it;
        fetch(issue, max(t - 2, 0));                       // (unconditional: a branch here costs register copies and waits)
        __builtin_amdgcn_sched_barrier(0);                 // the loads are ISSUED here, ahead of the step's arithmetic
        float dhz[4], sv[4][4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float dh = carry[reg] + cur[reg][0];
            const float rg = cur[reg][1], zg = cur[reg][2], ng = cur[reg][3], ghn = cur[reg][4], hp = cur[reg][5];
            const float dn = dh * (1.0f - zg);
            const float dz = dh * (hp - ng);
            dhz[reg] = dh * zg;
            const float dn_pre = dn * (1.0f - ng * ng);
            const float dr = dn_pre * ghn;
            const float dghn = dn_pre * rg;
            const float dr_pre = dr * rg * (1.0f - rg);
            const float dz_pre = dz * zg * (1.0f - zg);
            float* row = gb_w + (4 * q + reg) * GP;
            row[c] = dr_pre; row[GH + c] = dz_pre; row[2 * GH + c] = dghn;
            sv[reg][0] = dr_pre; sv[reg][1] = dz_pre; sv[reg][2] = dn_pre; sv[reg][3] = dghn;
        }
        lds_barrier();
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
        for (int s = 0; s < KQ; s += 12) {
            const float4 v0 = *reinterpret_cast<const float4*>(gb_w + c16 * GP + KQ * q + s);
            const float4 v1 = *reinterpret_cast<const float4*>(gb_w + c16 * GP + KQ * q + s + 4);
            const float4 v2 = *reinterpret_cast<const float4*>(gb_w + c16 * GP + KQ * q + s + 8);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.x, bw[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.y, bw[s + 1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.z, bw[s + 2], a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.w, bw[s + 3], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.x, bw[s + 4], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.y, bw[s + 5], a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.z, bw[s + 6], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.w, bw[s + 7], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v2.x, bw[s + 8], a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v2.y, bw[s + 9], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v2.z, bw[s + 10], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v2.w, bw[s + 11], a2, 0, 0, 0);
        }
        // (same ordering rule as the forward kernel: collect the prefetch, then issue this step's stores)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            carry[reg] = dhz[reg] + (a0[reg] + a1[reg] + a2[reg]);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                asm volatile("" : "+v"(next[reg][k]));
                cur[reg][k] = next[reg][k];
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if (valid[reg]) {
                constexpr unsigned G1 = GH * sizeof(float);
                const unsigned o = so[reg] + 3u * (unsigned)t * row_step;
                stg32(p.dgi, o, sv[reg][0]); stg32(p.dgi, o + G1, sv[reg][1]); stg32(p.dgi, o + 2 * G1, sv[reg][2]);
                stg32(p.dgh, go[reg] + (unsigned)t * row_step, sv[reg][3]);      // n block of d(gh) only: [rows, GH]
            }
        }
    };
    for (int t = tend - 1; t >= 0; t -= 2) {
        step(t, pb, pa);                       // pa holds step t-1 (in flight), pb receives step t-2
        if (t >= 1) step(t - 1, pa, pb);       // pb holds step t-2, pa receives step t-3
    }
}

// REFIL_GRU_VALU (opt-in, read per call): bit 0 = backward, bit 1 = forward recurrence of the 4-row kernels with the recurrent
// product on the vector ALUs (packed FMAs, the lane's own element over the whole reduction: no matrix-core instruction, no
// k-slice reduction; rnn_hidden_dim 64). Measured at cfg-T: the same time alone (85 / 102 us against 89 / 100), 20 % / 11 %
// FASTER in situ (171 / 174 us against 213 / 196: the 8-cycle 4x4x1 MFMAs queue behind the other chain's 64-cycle ones) -- and
// the step 0.9-1.5 % SLOWER (cfg2: +2.7 %): what the recurrences no longer wait for, the other chain's kernels now do.
static int gru_valu_bits() {
    const char* e = getenv("REFIL_GRU_VALU");
    return e ? atoi(e) : 0;
}
static bool gru_valu_enabled() { return gru_valu_bits() & 1; }
static bool gru_valu_fwd_enabled() { return gru_valu_bits() & 2; }
// rows per workgroup: 4-row tiles while there are fewer than three per CU (REFIL_GRU_ROWS=4 / 16 forces one). Measured: cfg3's
// backward (768 tiles on 256 CUs) 38 us faster on 16-row tiles, its forward (1024) 45 us; cfg5 (768 / 576) 20 us faster on 4-row tiles
static int gru_rows_per_wg(long rows) {
    static const int forced = [] { const char* e = getenv("REFIL_GRU_ROWS"); return e ? atoi(e) : 0; }();
    if (forced == 4 || forced == 16) return forced;
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return cdivl(rows, GR4) >= 3L * cus ? GROWS : GR4;
}

static int gru_check_fwd(const refil_gru_desc& d) {
    REFIL_CHECK(d.H == 32 || d.H == 64 || d.H == 128, "refil_gru: rnn_hidden_dim must be 32, 64 or 128 (got %d)", d.H);
    REFIL_CHECK(d.gi && d.hsx && d.w_hh && d.b_hh, "refil_gru_forward: null pointer");
    REFIL_CHECK(d.NR > 0 && d.T1 > 0 && d.na > 0, "refil_gru_forward: bad sizes");
    REFIL_CHECK(!d.save_r || (d.save_z && d.save_n && d.save_ghn), "refil_gru_forward: all four save buffers or none");
    REFIL_CHECK((!d.t_last && !d.ever) || d.B > 0, "refil_gru: t_last / ever need B");
    return 0;
}
static GruK gru_k(const refil_gru_desc& d) {
    return GruK{d.gi, d.hsx, d.w_hh, d.b_hh, d.save_r, d.save_z, d.save_n, d.save_ghn, d.dhs, d.dgi, d.dgh, d.NR, d.T1, d.na, d.t_last, d.B, d.zero_h0, d.ever};
}

// `second` (may be NULL): another, independent recurrence run by the same launch
int gru_forward_launch2(const refil_gru_desc& d, const refil_gru_desc* second, hipStream_t st) {
    if (int e = gru_check_fwd(d)) return e;
    if (second) {
        if (int e = gru_check_fwd(*second)) return e;
        REFIL_CHECK(second->H == d.H, "refil_gru: the two recurrences of one launch need the same hidden size");
    }
    const int GH = d.H;
    REFIL_CHECK((double)d.NR * (d.T1 + 1) * 3 * GH * sizeof(float) < 4.0e9 && (!second || (double)second->NR * (second->T1 + 1) * 3 * GH * sizeof(float) < 4.0e9),
                "refil_gru: NR * T1 too large for the 32-bit offsets of the recurrence kernels (split the batch)");
    GruK2 k;
    k.a = gru_k(d); k.b = second ? gru_k(*second) : k.a;
    const int rows_wg = gru_rows_per_wg(d.NR + (second ? second->NR : 0));
    k.nblk0 = cdiv(d.NR, rows_wg);
    const bool save = d.save_r != nullptr || (second && second->save_r != nullptr);
    const long rows_t = (long)d.NR * d.T1 + (second ? (long)second->NR * second->T1 : 0);
    dim3 grid(k.nblk0 + (second ? cdiv(second->NR, rows_wg) : 0));
    ProfScope prof(save ? "gru_fwd_kernel<true>" : "gru_fwd_kernel<false>", 2.0 * rows_t * GH * 3 * GH,
                   4.0 * rows_t * GH * (save ? 8.0 : 4.0), st);
#define GRU_FWD(HH) do { if (rows_wg == GROWS) { if (save) hipLaunchKernelGGL((gru_fwd16_kernel<true, HH>), grid, dim3(4 * HH), 0, st, k); \
                                                  else hipLaunchKernelGGL((gru_fwd16_kernel<false, HH>), grid, dim3(4 * HH), 0, st, k); } \
                        else if (pd2) { if (save) hipLaunchKernelGGL((gru_fwd4_kernel<true, HH, false, 2>), grid, dim3(4 * HH), 0, st, k); \
                                        else hipLaunchKernelGGL((gru_fwd4_kernel<false, HH, false, 2>), grid, dim3(4 * HH), 0, st, k); } \
                        else { if (save) hipLaunchKernelGGL((gru_fwd4_kernel<true, HH>), grid, dim3(4 * HH), 0, st, k); \
                               else hipLaunchKernelGGL((gru_fwd4_kernel<false, HH>), grid, dim3(4 * HH), 0, st, k); } } while (0)
    // prefetch distance of the step inputs: 4 steps (REFIL_GRU_PD) or 2 ("gru_pd", refil_set_tuning / REFIL_GRU_PD_RT: fewer
    // registers in flight; cfg-T -1.1 %, cfg2 +0.6 % -- one of the knobs QLearner's first call measures per shape)
    static const int pd_env = [] { const char* e = getenv("REFIL_GRU_PD_RT"); return e ? atoi(e) : -1; }();
    const bool pd2 = (g_tuning.gru_pd > 0 ? g_tuning.gru_pd : pd_env) == 2;
    if (GH == 64 && rows_wg != GROWS && gru_valu_fwd_enabled()) {
        if (save) hipLaunchKernelGGL((gru_fwd4_kernel<true, 64, true>), grid, dim3(256), 0, st, k);
        else hipLaunchKernelGGL((gru_fwd4_kernel<false, 64, true>), grid, dim3(256), 0, st, k);
    } else
    if (GH == 32) GRU_FWD(32); else if (GH == 64) GRU_FWD(64); else GRU_FWD(128);
#undef GRU_FWD
    REFIL_LAUNCH_CHECK();
    return 0;
}
int gru_forward_launch(const refil_gru_desc& d, hipStream_t st) { return gru_forward_launch2(d, nullptr, st); }

int gru_backward_launch(const refil_gru_desc& d, hipStream_t st) {
    REFIL_CHECK(d.H == 32 || d.H == 64 || d.H == 128, "refil_gru: rnn_hidden_dim must be 32, 64 or 128 (got %d)", d.H);
    const int GH = d.H;
    REFIL_CHECK((double)d.NR * (d.T1 + 1) * 3 * GH * sizeof(float) < 4.0e9, "refil_gru: NR * T1 too large for the 32-bit offsets of the recurrence kernels (split the batch)");
    REFIL_CHECK(d.hsx && d.w_hh && d.save_r && d.save_z && d.save_n && d.save_ghn && d.dhs && d.dgi && d.dgh,
                "refil_gru_backward: null pointer");
    REFIL_CHECK(d.NR > 0 && d.T1 > 0 && d.na > 0, "refil_gru_backward: bad sizes");
    REFIL_CHECK((!d.t_last && !d.ever) || d.B > 0, "refil_gru: t_last / ever need B");
    GruK k = gru_k(d);
    ProfScope prof("gru_bwd_kernel", 2.0 * d.NR * d.T1 * GH * 3 * GH, 4.0 * d.NR * d.T1 * GH * 10.0, st);
    if (gru_rows_per_wg(d.NR) == GROWS) {
        if (GH == 32) hipLaunchKernelGGL(gru_bwd16_kernel<32>, dim3(cdiv(d.NR, GROWS)), dim3(128), 0, st, k);
        else if (GH == 64) hipLaunchKernelGGL(gru_bwd16_kernel<64>, dim3(cdiv(d.NR, GROWS)), dim3(256), 0, st, k);
        else hipLaunchKernelGGL(gru_bwd16_kernel<128>, dim3(cdiv(d.NR, GROWS)), dim3(512), 0, st, k);
    } else {
        static const int pd_env = [] { const char* e = getenv("REFIL_GRU_PD_RT"); return e ? atoi(e) : -1; }();
        const bool pd2 = (g_tuning.gru_pd > 0 ? g_tuning.gru_pd : pd_env) == 2;
        if (pd2 && !(GH == 64 && gru_valu_enabled())) {
            if (GH == 32) hipLaunchKernelGGL((gru_bwd4_kernel<32, false, 2>), dim3(cdiv(d.NR, GR4)), dim3(128), 0, st, k);
            else if (GH == 64) hipLaunchKernelGGL((gru_bwd4_kernel<64, false, 2>), dim3(cdiv(d.NR, GR4)), dim3(256), 0, st, k);
            else hipLaunchKernelGGL((gru_bwd4_kernel<128, false, 2>), dim3(cdiv(d.NR, GR4)), dim3(512), 0, st, k);
        } else
        if (GH == 32) hipLaunchKernelGGL(gru_bwd4_kernel<32>, dim3(cdiv(d.NR, GR4)), dim3(128), 0, st, k);
        else if (GH == 64 && gru_valu_enabled()) hipLaunchKernelGGL((gru_bwd4_kernel<64, true>), dim3(cdiv(d.NR, GR4)), dim3(256), 0, st, k);
        else if (GH == 64) hipLaunchKernelGGL(gru_bwd4_kernel<64>, dim3(cdiv(d.NR, GR4)), dim3(256), 0, st, k);
        else hipLaunchKernelGGL(gru_bwd4_kernel<128>, dim3(cdiv(d.NR, GR4)), dim3(512), 0, st, k);
    }
    REFIL_LAUNCH_CHECK();
    return 0;
}

}  // namespace refil

extern "C" int refil_gru_forward(const refil_gru_desc* desc, void* stream) {
    REFIL_CHECK(desc, "refil_gru_forward: null desc");
    return refil::gru_forward_launch(*desc, (hipStream_t)stream);
}
extern "C" int refil_gru_backward(const refil_gru_desc* desc, void* stream) {
    REFIL_CHECK(desc, "refil_gru_backward: null desc");
    return refil::gru_backward_launch(*desc, (hipStream_t)stream);
}

#ifdef REFIL_GRU_TIMING
extern "C" int refil_debug_gru_timing(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(refil::g_gru_dbg), 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
