// Weight-gradient GEMM, second generation:  dW[N_out, K_in] = dy[R, N_out]^T x[R, K_in]  (+ db = column sums of dy).
//
// gemm_dw.hip showed that for these products (tens of thousands of reduction rows, outputs of at most a few hundred
// rows / columns) the operands can go from global memory straight into the MFMA operand registers -- the reduction
// row is the slow index of both, so a plain global_load_dword per lane is coalesced AND already in
// v_mfma_f32_32x32x2_f32 operand layout. What bounded that kernel was its register-level reuse: a wave owned 2 x 2
// MFMA tiles, i.e. 4 operand loads per 4 MFMAs, and with 8 waves per CU the vector-memory pipe (256-byte wave loads),
// not the matrix pipe, set the pace (0.3-0.35 of the fp32 MFMA rate).
//
// Here a wave owns TI x TJ tiles (up to 4 x 4 = 128 x 128 outputs, 256 accumulator registers -- gfx950's unified
// 512-entry register file makes that a one-wave-per-SIMD kernel): 8 operand loads feed 16 MFMAs (1024 matrix-pipe
// cycles), a quarter of the load traffic per FLOP, and the prefetch ring alone (D steps = D x 1024 cycles) covers
// the HBM latency, so one wave per SIMD is enough. The four waves of a workgroup work on the SAME output tile and
// split the rows of the workgroup's range between them; their accumulators are summed through LDS (fixed order:
// deterministic) so that only ONE partial tile per workgroup goes to memory -- 4 x fewer partial bytes than one per
// wave. The per-split partials are added up by reduce_partials_kernel (gemm.hip), as before.
//
// Operand loads are VECTORS along the output index: a lane fetches the TI (TJ) consecutive columns
// m0 + TI * (lane % 32) + {0..TI-1} of its row with ONE 4/8/12/16-byte load and spreads them over its TI tiles -- tile i
// then holds the output rows m0 + TI * k + i, a permutation that is undone for free when the tile is stored (and makes
// the partial-tile stores 16-byte vectors as well). Two loads per step instead of eight: the 63 vector-memory
// operations a wave can have in flight (vmcnt) now cover a 12-step prefetch ring instead of 6.
//
// Row lists (refil_gemm_desc.row_index): the reduction rows come from a device-side list whose length is read from
// device memory; the indices travel through their own ring, one period ahead of the operand prefetch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bufops.h"
#include "common.h"
#include "kernels.h"
#include "profile.h"
#include "split.h"

namespace refil {


constexpr int DW4_PASS = 128;     // accumulator registers summed through LDS per pass (at most 3 writer waves x 128 x 64 x 4 B = 96 KB)

// prefetch ring depth (steps of 2 rows): bounded by the registers of the ring (D * (TI + TJ + 1))
constexpr int dw4_depth(int ti, int tj) { return ti * tj >= 12 ? 12 : 16; }
constexpr int dw4_pass(int ti, int tj) { return (ti * tj * 16 + ti + 7) / 8 * 8 < DW4_PASS ? (ti * tj * 16 + ti + 7) / 8 * 8 : DW4_PASS; }

template <int T> struct VecT;
template <> struct VecT<1> { float v[1]; };
template <> struct __attribute__((aligned(8))) VecT<2> { float v[2]; };
template <> struct VecT<3> { float v[3]; };
template <> struct __attribute__((aligned(16))) VecT<4> { float v[4]; };
// T consecutive floats at base[off] (base uniform, off a 32-bit element offset: the saddr + voffset addressing form, no
// 64-bit address arithmetic per lane). 12-byte vectors are three dword loads (global_load_dwordx3 measured 1.7x slower).
template <int T>
__device__ inline VecT<T> ldv(const float* __restrict__ base, unsigned off) {
    if (T == 3) {
        VecT<T> r;
        r.v[0] = base[off]; r.v[1] = base[off + 1]; r.v[T - 1] = base[off + T - 1];
        return r;
    }
    return *reinterpret_cast<const VecT<T>*>(base + off);
}

// 32-bit row map: physical row of logical row r, quotient by multiplication (exact for r * grp < 2^32, checked on the
// host); v_mul_hi_u32 + two 24-bit multiplies instead of the 64-bit sequence of RowMap
struct RowMap32 {
    unsigned grp, gstride, off, magic;
    __device__ inline unsigned operator()(unsigned r) const {
        const unsigned q = __umulhi(r, magic);
        return __umul24(q, gstride) + (r - __umul24(q, grp)) + off;
    }
};

struct Dw4K {
    const float* A; const float* B; float* partial;
    int M, N, R, lda, ldb;
    long sA, sB;
    RowMap32 amap, bmap;
    int splits, batch, colsum;
    const int* ridx; const int* rcount;
};

// WMT: waves of a workgroup side by side along the output rows (they walk the SAME reduction rows, so the x operand is
// fetched from HBM once and hits the vector cache for the others); the remaining factor WKS = 4 / WMT splits the rows.
template <int TI, int TJ, int WMT, bool IDX>
__global__ __launch_bounds__(256, 1) void gemm_dw4_kernel(Dw4K p) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [<= 3 writer waves][PASS][64]
    constexpr int D = dw4_depth(TI, TJ), WKS = 4 / WMT, PASS = dw4_pass(TI, TJ);
    constexpr int NM = TI * TJ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane31 = lane & 31, hf = lane >> 5;
    const int wm = wave / WKS, wk = wave % WKS;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z % p.splits;
    const int m0 = (blockIdx.y * WMT + wm) * 32 * TI, n0 = blockIdx.x * 32 * TJ;
    const float* __restrict__ A = p.A + bz * p.sA;
    const float* __restrict__ B = p.B + bz * p.sB;

    // rows of this split (a multiple of WKS * 2 D so that every wave's part runs whole ring periods), then of this wave
    const int R = IDX ? *p.rcount : p.R;
    const int chunk = cdiv(cdiv(R, p.splits), WKS * 2 * D) * WKS * 2 * D;
    const int wchunk = chunk / WKS;
    const int rbeg = min(R, sp * chunk + wk * wchunk), rend = min(R, rbeg + wchunk);
    const int nfull = (rend - rbeg) / (2 * D);

    // first column of this lane's TI (TJ) consecutive ones; lanes past the matrix edge re-read the last whole group
    // (M % TI == 0, N % TJ == 0) and their results are simply not stored
    const unsigned ca = min(m0 + TI * lane31, p.M - TI), cb = min(n0 + TJ * lane31, p.N - TJ);
    const unsigned lda = p.lda, ldb = p.ldb;
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) csum[i] = 0.f;

    if (nfull > 0) {
        // Software pipeline over steps g = 0, 1, .. (2 rows each): step g multiplies ring slot g % D and, in the middle of
        // its MFMAs, fills slot (g - 1) % D -- consumed one step earlier -- with the rows of step g + D - 1. Loading the
        // PREVIOUS slot leaves no register dependency between a step's loads and its MFMAs. All address arithmetic is
        // 32-bit (element offsets from a uniform base, 24-bit multiplies): with one wave per SIMD every VALU instruction
        // competes with the MFMA issue, and the 64-bit row-map / address sequences cost more cycles than the 16 MFMAs.
        // Row lists: ri[s] = list entry slot s is filled with next (fetched one ring period ahead). The list is padded
        // with scratch-row indices far enough for the prefetch to run past the end.
        const unsigned rlast = (unsigned)(R - 1);
        unsigned pf = rbeg + hf;                                          // position (list index / row) of the next fill
        auto offsets = [&](unsigned row, unsigned& oa, unsigned& ob) {
            oa = __umul24(p.amap(row), lda) + ca;
            ob = __umul24(p.bmap(row), ldb) + cb;
        };
        unsigned ri[IDX ? D : 1];
        VecT<TI> ra[D];
        VecT<TJ> rb[D];
#pragma unroll
        for (int s = 0; s < D - 1; ++s) {                                 // steps 0 .. D-2
            unsigned oa, ob;
            offsets(IDX ? (unsigned)p.ridx[pf] : pf, oa, ob);
            ra[s] = ldv<TI>(A, oa); rb[s] = ldv<TJ>(B, ob);
            pf += 2;
        }
        if (IDX) {
#pragma unroll
            for (int s = 0; s < D; ++s) ri[(s + D - 1) % D] = p.ridx[pf + 2 * s];    // entries of steps D-1 .. 2D-2
        }
        // The prologue's loads are COMPLETE before the ring starts: the compiler orders them freely (the slot consumed first
        // was fetched last), and at the loop head it takes the more conservative of the two incoming paths for every register --
        // with anything of the prologue still in flight that is vmcnt(0) at the head of EVERY period, i.e. the ring runs empty
        // once per D steps. With nothing pending on the entry path the waits inside the loop count the ring's loads exactly.
        __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
        for (int it = 0; it < nfull; ++it) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int sp_ = (s + D - 1) % D;                          // slot consumed by the previous step
                unsigned oa, ob;
                offsets(IDX ? ri[IDX ? sp_ : 0] : min(pf, rlast), oa, ob);
#pragma unroll
                for (int k = 0; k < NM; ++k)
                    acc[k / TJ][k % TJ] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s].v[k / TJ], rb[s].v[k % TJ], acc[k / TJ][k % TJ], 0, 0, 0);
                ra[sp_] = ldv<TI>(A, oa);
                rb[sp_] = ldv<TJ>(B, ob);
                if (IDX) ri[IDX ? sp_ : 0] = p.ridx[pf + 2 * D];
                pf += 2;
                // the schedule of this step: half of the MFMAs, the step's loads, the other half
                __builtin_amdgcn_sched_group_barrier(0x008, (NM + 1) / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, (TI == 3 ? 3 : 1) + (TJ == 3 ? 3 : 1) + (IDX ? 1 : 0), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NM / 2, 0);
                // column sums of dy (the bias gradient). As plain C++ adds the scheduler gathers the adds of ALL D slots at the head
                // of the ring period -- where every slot but one is still in flight, i.e. s_waitcnt vmcnt(0) once per period and
                // the prefetch ring runs empty. A volatile asm stays between the loads of its own step.
#pragma unroll
                for (int i = 0; i < TI; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(csum[i]) : "v"(ra[s].v[i]));
            }
        }
    }
    // tail rows (fewer than 2 D): plain loop with zero fill
    for (int r2 = rbeg + nfull * 2 * D; r2 < rend; r2 += 2) {      // wave-uniform trip count: MFMAs ignore EXEC
        const int r = r2 + hf;
        const bool ok = r < rend;
        unsigned rr = ok ? r : rbeg;
        if (IDX) rr = p.ridx[rr];
        VecT<TI> va = ldv<TI>(A, __umul24(p.amap(rr), lda) + ca);
        VecT<TJ> vb = ldv<TJ>(B, __umul24(p.bmap(rr), ldb) + cb);
#pragma unroll
        for (int i = 0; i < TI; ++i) va.v[i] = ok ? va.v[i] : 0.f;
#pragma unroll
        for (int j = 0; j < TJ; ++j) vb.v[j] = ok ? vb.v[j] : 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va.v[i], vb.v[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TI; ++i) csum[i] += va.v[i];
    }

    // ---- sum the WKS row-split waves' accumulators (and column sums) of each output tile through LDS into its wk = 0
    //      wave, PASS registers per pass, fixed order (deterministic) ----
    constexpr int NREG = TI * TJ * 16;
    if (WKS > 1) {
        float* mine = red + (size_t)(wm * (WKS - 1) + (wk - 1)) * PASS * 64;      // writer slot (wk > 0)
        const float* theirs = red + (size_t)(wm * (WKS - 1)) * PASS * 64;         // first writer slot of this output tile
#pragma unroll
        for (int pass = 0; pass * PASS < NREG + TI; ++pass) {
            if (wk > 0) {
#pragma unroll
                for (int k = 0; k < PASS; ++k) {
                    const int f = pass * PASS + k;          // flat register index: accumulators first, then the column sums
                    if (f < NREG) mine[k * 64 + lane] = acc[f / (16 * TJ)][(f / 16) % TJ][f % 16];
                    else if (f < NREG + TI) mine[k * 64 + lane] = csum[f - NREG];
                }
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int k = 0; k < PASS; ++k) {
                    const int f = pass * PASS + k;
                    if (f < NREG + TI) {
                        float t = theirs[k * 64 + lane];
#pragma unroll
                        for (int w = 1; w < WKS - 1; ++w) t += theirs[(w * PASS + k) * 64 + lane];
                        if (f < NREG) acc[f / (16 * TJ)][(f / 16) % TJ][f % 16] += t;
                        else csum[f - NREG] += t;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (wk != 0 || m0 >= p.M) return;

    // ---- partial tile of this split: partial[(bz * splits + sp)][M][N]. Register r of tile (i, j) is the output element
    //      (row m0 + TI * ((r&3) + 8 (r>>2) + 4 hf) + i, column n0 + TJ * lane31 + j): TJ consecutive columns per lane ----
    float* P = p.partial + ((long)bz * p.splits + sp) * p.M * p.N;
    const int ncol = n0 + TJ * lane31;
    const bool col_ok = ncol + TJ <= p.N;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + TI * ((r & 3) + 8 * (r >> 2) + 4 * hf) + i;
            if (m < p.M && col_ok) {
                VecT<TJ> o;
#pragma unroll
                for (int j = 0; j < TJ; ++j) o.v[j] = acc[i][j][r];
                *reinterpret_cast<VecT<TJ>*>(P + (long)m * p.N + ncol) = o;
            }
        }
    if (p.colsum && blockIdx.x == 0) {
        float* CS = p.partial + (long)p.batch * p.splits * p.M * p.N + ((long)bz * p.splits + sp) * p.M;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const float t = csum[i] + __shfl_xor(csum[i], 32, 64);      // rows of both halves
            const int m = m0 + TI * lane31 + i;
            if (hf == 0 && m0 + TI * lane31 + TI <= p.M) CS[m] = t;
        }
    }
}

// ---- bf16 x 6 form (gemm_wres.hip: wr_split) of the weight gradient for the widest outputs: N = 128, M a multiple of 128 --------
// A 16-row reduction step of a 32 x 32 output tile is 6 x v_mfma_f32_32x32x16_bf16 (192 matrix-pipe cycles) instead of
// 8 x v_mfma_f32_32x32x2_f32 (512). The operand layout of that instruction wants, per lane, 8 CONSECUTIVE reduction rows of one
// output column -- and every element split into three bf16 pieces (~5.5 VALU operations). Done per wave in registers that is
// 44 (TI + TJ) operations against 6 TI TJ MFMAs: it hides behind the matrix pipe (<= 5 single-issue fillers per MFMA at one wave
// per SIMD) for 4 x 4 tiles only, which leave no registers for it. Here the split is done ONCE per workgroup and shared through LDS:
//   * the workgroup (4 waves) owns a 128 TI x 128 output tile and a range of reduction rows; wave w multiplies the tile rows
//     w TI .. w TI + TI - 1 (TI x 4 MFMA tiles, 64 TI accumulator registers);
//   * producer role (every thread): unit (column c, half hf) = rows 8 hf .. 8 hf + 7 of one of the 128 TI + 128 operand columns of
//     the step: 8 dword loads (coalesced along the columns), 20 micro-steps of splitting, three 16-byte LDS writes -- one per
//     plane, already in MFMA operand layout (column pitch 48 bytes: conflict-free for the writes and the reads). TI + 1 units per
//     thread and step; the raw rows travel through a 4-deep register ring (3 steps = ~5 k matrix-pipe cycles of prefetch);
//   * consumer role: 3 (TI + 4) LDS reads of 16 bytes at the head of a step, then 24 TI MFMAs round-robin over the accumulators
//     (never two in a row on the same one), with the producer work of the NEXT step dealt out behind them, pinned by
//     sched_barriers (gemm_wres.hip, DESIGN.md lesson 29); one LDS-only barrier per step, two plane buffers;
//   * rows that do not fill a 64-row period of the ring (the end of the last workgroup's range) go through the fp32 instruction on
//     the same accumulators; column sums of dy (the bias gradient) are taken from the raw rows by the producers.
// Partial tiles / column sums have the layout of gemm_dw4_kernel: the same reduction kernels add them up.
// NJ: 32-column tiles of the x operand (N <= 32 NJ; columns past N are requested through an out-of-range offset: zeros). BMAP: the x
// rows go through a row map (agents' rows inside entity-major storage): physical row = r + (r / grp) (gstride - grp) + off, folded into
// the request's offset as one extra multiply-add per request plus one v_mul_hi per row.
constexpr int dws_units(int ti, int nj) { return (4 * ti + nj + 3) / 4; }                       // producer units (32-column chunks) per thread
#ifndef DWS_DEPTH
#define DWS_DEPTH 4
#endif
#ifndef DWS_PERIOD
#define DWS_PERIOD 4
#endif
// ring depth D (steps of 16 rows requested ahead) and steps per trip of the main loop P (a multiple of D, even: two plane buffers)
constexpr int DWS_PB = 48, DWS_D = DWS_DEPTH, DWS_P = DWS_PERIOD;
static_assert(DWS_P % DWS_D == 0 && DWS_P % 2 == 0 && DWS_D >= 3, "gemm_dws_kernel: period must be an even multiple of the ring depth");
constexpr size_t dws_smem(int ti, int nj) { return (size_t)2 * 3 * 128 * dws_units(ti, nj) * DWS_PB + 2 * 128 * ti * sizeof(float); }

template <int TI, int NJ, bool IDX, bool BMAP>
__global__ __launch_bounds__(256, 1) void gemm_dws_kernel(Dw4K p) {
    extern __shared__ __attribute__((aligned(16))) char dws_lds[];
    constexpr int U = dws_units(TI, NJ), NCOL = 128 * U, PB = DWS_PB, PS = NCOL * PB, BUF = 3 * PS, D = DWS_D, PER = DWS_P, NG = 6 * NJ * TI;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave: provably uniform)
    const int lane31 = lane & 31, hf = lane >> 5;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z % p.splits;
    const int m0 = blockIdx.y * 128 * TI;
    const float* __restrict__ A = p.A + bz * p.sA;
    const float* __restrict__ B = p.B + bz * p.sB;
    const int R = IDX ? *p.rcount : p.R;
    const int chunk = cdiv(cdiv(R, p.splits), 16 * PER) * 16 * PER;
    const int rbeg = min(R, sp * chunk), rend = min(R, rbeg + chunk);
    const int nper = (rend - rbeg) / (16 * PER);                 // whole periods (P steps of 16 rows)

    // producer units of this thread: column chunk q = wave U + u of [A columns m0 .. m0 + 128 TI | B columns 0 .. 127]. All requests are
    // buffer loads (wave-uniform resource + 32-bit byte offset; an offset past the resource returns zeros: steps past the range are
    // requested and converted like any other and never multiplied) -- and, unlike plain loads through __restrict__ pointers, they stay
    // where the schedule below puts them
    const long phys = IDX ? (1L << 24) : (long)R;              // rows the operands may be read at
    const rsrc_t rsA = mk_rsrc(A, ((phys - 1) * p.lda + p.M) * 4), rsB = mk_rsrc(B, (((BMAP ? (1L << 24) : phys) - 1) * p.ldb + p.N) * 4);
    const rsrc_t rsI = mk_rsrc(IDX ? p.ridx : nullptr, IDX ? (long)R * 4 : 0);
    const rsrc_t rs_none = mk_rsrc(A, 0);
    rsrc_t urs[U];
    unsigned uld4[U], ucol4[U], udelta4[U];
    bool uisa[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = wave * U + u;
        uisa[u] = q < 4 * TI;
        urs[u] = uisa[u] ? rsA : (q < 4 * TI + NJ ? rsB : rs_none);
        uld4[u] = 4u * (uisa[u] ? p.lda : p.ldb);
        const int bcol = 32 * (q - 4 * TI) + lane31;
        ucol4[u] = uisa[u] ? 4u * (m0 + 32 * q + lane31) : (bcol < p.N ? 4u * bcol + (BMAP ? p.bmap.off * 4u * p.ldb : 0u) : 0x80000000u);
        udelta4[u] = (BMAP && !uisa[u]) ? (p.bmap.gstride - p.bmap.grp) * 4u * p.ldb : 0u;
    }
    char* const wr0 = dws_lds + (32 * wave * U + lane31) * PB + 16 * hf;            // + 32 u PB + plane PS + buffer BUF
    const char* const rdA = dws_lds + (32 * wave * TI + lane31) * PB + 16 * hf;      // + 32 i PB
    const char* const rdB = dws_lds + (128 * TI + lane31) * PB + 16 * hf;            // + 32 j PB (chunks 4 TI .. 4 TI + NJ - 1)

    f32x16 acc[TI][NJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum[U];
#pragma unroll
    for (int u = 0; u < U; ++u) csum[u] = 0.f;

    if (nper > 0) {
        float raw[D][U][8];
        int ri[D][8];                                           // IDX: list entries of the steps whose rows are requested next (set = step % D)
        // list entries / row numbers of step t for this lane's half (8 consecutive positions)
        auto load_idx = [&](int t, int* o) {
            const int off = 4 * (rbeg + 16 * t + 8 * hf);
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsI, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rsI, off + 16, 0, 0);
            o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
        };
        auto row_of = [&](int t, int k, const int* idx) -> unsigned { return IDX ? (unsigned)idx[k] : (unsigned)(rbeg + 16 * t + 8 * hf + k); };
        // (steps past the range are requested through an EMPTY resource: zeros come back, nothing moves -- their column sums add nothing
        // and their planes are never multiplied; t and nsteps are uniform, the choice is four scalar selects)
        const int nsteps = nper * PER;
        auto load_raw = [&](int t, int k, int u, const int* idx, float* o) {
            const rsrc_t rs = t < nsteps ? urs[u] : rs_none;
            const unsigned r = row_of(t, k, idx);
            unsigned off = __umul24(r, uld4[u]) + ucol4[u];
            if (BMAP) off += __umul24(__umulhi(r, p.bmap.magic), udelta4[u]);
            o[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
        };
        // split state of the unit being converted
        float sx[4], sy[4];
        unsigned sh[4], sm[4], sl[4];
        auto split_micro = [&](auto k_, const float* rw, int u) {
            constexpr int k = decltype(k_)::value, q = k % 4, t = k / 4;      // (the four pairs' chains interleaved: consecutive micro-steps are independent)
            if constexpr (t == 0) {
                sx[q] = rw[2 * q]; sy[q] = rw[2 * q + 1];
                if (uisa[u]) csum[u] += sx[q] + sy[q];
                sh[q] = wr_pk(sx[q], sy[q]);
            } else if constexpr (t == 1) { sx[q] = wr_sub(sx[q], __uint_as_float(sh[q] << 16)); sy[q] = wr_sub(sy[q], __uint_as_float(sh[q] & 0xFFFF0000u)); }
            else if constexpr (t == 2) sm[q] = wr_pk(sx[q], sy[q]);
            else if constexpr (t == 3) { sx[q] = wr_sub(sx[q], __uint_as_float(sm[q] << 16)); sy[q] = wr_sub(sy[q], __uint_as_float(sm[q] & 0xFFFF0000u)); }
            else sl[q] = wr_pk(sx[q], sy[q]);
        };
        auto write_planes = [&](int u, int buf) {
            char* w = wr0 + buf * BUF + 32 * u * PB;
            *reinterpret_cast<wr_u32x4*>(w) = wr_u32x4{sh[0], sh[1], sh[2], sh[3]};
            *reinterpret_cast<wr_u32x4*>(w + PS) = wr_u32x4{sm[0], sm[1], sm[2], sm[3]};
            *reinterpret_cast<wr_u32x4*>(w + 2 * PS) = wr_u32x4{sl[0], sl[1], sl[2], sl[3]};
        };

        // prologue: rows of steps 0 .. D - 1 requested, step 0 split into buffer 0
#pragma unroll
        for (int t = 0; t < D; ++t) {
            if (IDX) load_idx(t, ri[t]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) load_raw(t, k, u, ri[t], raw[t][u]);
        }
        if (IDX) { load_idx(D, ri[0]); load_idx(D + 1, ri[1]); }     // (entries of steps D, D + 1: requested two steps ahead of their rows)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            static_for<20>([&](auto k_) { split_micro(k_, raw[0][u], u); });
            write_planes(u, 0);
        }
        // everything of the prologue is complete before the ring starts: the waits inside the loop then count its loads exactly
        __builtin_amdgcn_s_waitcnt(vmcnt_only(0));
        lds_barrier();

        for (int it = 0; it < nper; ++it) {
            static_for<PER>([&](auto s_) {
                constexpr int s = decltype(s_)::value;             // step it PER + s: multiplies buffer s & 1, converts step + 1 into the other
                const int step = it * PER + s;
                // operand planes of this step
                wr_u32x4 ap[TI][3], bp[NJ][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < TI; ++i) ap[i][pl] = *reinterpret_cast<const wr_u32x4*>(rdA + (s & 1) * BUF + pl * PS + 32 * i * PB);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bp[j][pl] = *reinterpret_cast<const wr_u32x4*>(rdB + (s & 1) * BUF + pl * PS + 32 * j * PB);
                }
                // Work dealt out behind the MFMAs, in this order: the list entries of step + D + 1, the row requests of step + D into the ring
                // slot converted one step EARLIER (slot s: no register dependency on this step's conversions), the conversion of step + 1
                // (slot s + 1) into the other plane buffer. Items: 1 + 8 U requests, U x (20 micro-steps + 1 write).
                constexpr int NREQ = 8 * U, NCONV = 21 * U, NW = NREQ + NCONV + 1;
                // (list entries: set s holds those of step + D, requested TWO steps ago: the wait for them counts past the row requests
                // of the two steps in between instead of draining them -- the 6-bit vmcnt holds ~2.4 steps of requests)
                auto item = [&](auto w_) {
                    constexpr int w = decltype(w_)::value;
                    if constexpr (w == 0) {
                        if (IDX) load_idx(step + D + 2, ri[(s + 2) % D]);
                    } else if constexpr (w <= NREQ) {
                        constexpr int u = (w - 1) / 8, k = (w - 1) % 8;
                        load_raw(step + D, k, u, ri[s % D], raw[s % D][u]);
                    } else {
#ifndef DWS_DEBUG_NO_CONV
                        constexpr int c = w - NREQ - 1, u = c / 21, k = c % 21;
                        if constexpr (k < 20) split_micro(std::integral_constant<int, k>{}, raw[(s + 1) % D][u], u);
                        else write_planes(u, (s + 1) & 1);
#else
                        constexpr int c = w - NREQ - 1, u = c / 21, k = c % 21;      // (timing probe: the rows are consumed, nothing is converted)
                        if constexpr (k == 0) csum[u] += raw[(s + 1) % D][u][0] + raw[(s + 1) % D][u][7];
#endif
                    }
                };
                __builtin_amdgcn_sched_barrier(0);
                static_for<NG>([&](auto g_) {
                    constexpr int g = decltype(g_)::value, pr = g / (NJ * TI), j = (g / TI) % NJ, i = g % TI;
                    constexpr int apl = pr == 0 ? 2 : ((pr == 2 || pr == 3) ? 1 : 0);            // lo, hi, mid, mid, hi, hi
                    constexpr int bpl = pr == 0 ? 0 : (pr == 1 ? 2 : (pr == 2 ? 1 : (pr == 3 ? 0 : (pr == 4 ? 1 : 0))));
#ifndef DWS_DEBUG_NO_MFMA
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wr_bf16x8, ap[i][apl]), __builtin_bit_cast(wr_bf16x8, bp[j][bpl]),
                                                                        acc[i][j], 0, 0, 0);
#else
                    if (g < TI * NJ) acc[i][j][0] += __uint_as_float(ap[i][apl][0] ^ bp[j][bpl][0]);      // (timing probe)
#endif
                    constexpr int w0 = g * NW / NG, w1 = (g + 1) * NW / NG;
                    static_for<w1 - w0>([&](auto d_) { item(std::integral_constant<int, w0 + decltype(d_)::value>{}); });
                    __builtin_amdgcn_sched_barrier(0);
                });
                lds_barrier();
            });
        }
    }

    // ---- rows past the last whole period: the fp32 instruction on the same accumulators (2 rows per step) ----
    const int tbeg = rbeg + nper * 16 * PER;
    for (int r2 = tbeg; r2 < rend; r2 += 2) {                    // workgroup-uniform trip count
        const int r = r2 + hf;
        const bool ok = r < rend;
        unsigned rr = ok ? r : rbeg;
        if (IDX) rr = p.ridx[rr];
        float va[TI], vb[NJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) { const float v = A[__umul24(rr, (unsigned)p.lda) + m0 + 32 * (wave * TI + i) + lane31]; va[i] = ok ? v : 0.f; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool okb = ok && 32 * j + lane31 < p.N;
            const float v = B[__umul24(BMAP ? p.bmap(rr) : rr, (unsigned)p.ldb) + (okb ? 32 * j + lane31 : 0)];
            vb[j] = okb ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i], vb[j], acc[i][j], 0, 0, 0);
    }

    // ---- partial tile of this split: partial[(bz * splits + sp)][M][N]; register r of tile (i, j) is the output element
    //      (row m0 + 32 (wave TI + i) + (r & 3) + 8 (r >> 2) + 4 hf, column 32 j + lane31) ----
    float* P = p.partial + ((long)bz * p.splits + sp) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * (wave * TI + i) + (r & 3) + 8 * (r >> 2) + 4 * hf;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (32 * j + lane31 < p.N) P[(long)m * p.N + 32 * j + lane31] = acc[i][j][r];
        }
    if (p.colsum) {
        // column sums: the producers' sums of the whole periods (two halves per column, fixed order) + the tail rows
        float* cs = reinterpret_cast<float*>(dws_lds + 2 * BUF);         // [2][128 TI]
        lds_barrier();
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (uisa[u]) cs[hf * 128 * TI + 32 * (wave * U + u) + lane31] = csum[u];
        lds_barrier();
        if (tid < 128 * TI) {
            float t = cs[tid] + cs[128 * TI + tid];
            for (int r = tbeg; r < rend; ++r) t += A[__umul24(IDX ? (unsigned)p.ridx[r] : (unsigned)r, (unsigned)p.lda) + m0 + tid];
            float* CS = p.partial + (long)p.batch * p.splits * p.M * p.N + ((long)bz * p.splits + sp) * p.M;
            CS[m0 + tid] = t;
        }
    }
}

// wave tile (ti x tj MFMA tiles: the count in [1,4] that wastes the fewest padded rows / columns, the larger on ties) and
// waves side by side along the output rows
static void dw4_shape(int M, int N, int& ti, int& tj, int& wmt) {
    auto pick = [](int n) {
        int best = 1, waste = 1 << 30;
        for (int t = 1; t <= 4; ++t) {
            const int w = cdiv(n, 32 * t) * 32 * t - n;
            if (n % t == 0 && w <= waste) { waste = w; best = t; }       // (a lane owns t consecutive columns: n % t == 0)
        }
        return best;
    };
    ti = pick(M); tj = pick(N);
    // (4 x 4 tiles = 256 accumulators: with the ring and the column sums beside them the 512-entry register file spills 32-139
    // registers -- code-object metadata, round 4 -- so a 128 x 128 wave tile is walked as two 128 x 64 ones; only the fp32-instruction
    // form of these gradients gets here, the bf16 x 6 kernel takes them by default)
    if (ti * tj == 16) tj = 2;
    const int mt = cdiv(M, 32 * ti);
    wmt = (mt >= 4 && ti == 4) ? 4 : ((mt >= 2 && ti >= 2) ? 2 : 1);       // (only these (ti, wmt) pairs are instantiated)
}

bool gemm_dw4_enabled() {
    static const bool on = [] { const char* e = getenv("REFIL_GEMM_DW4"); return !(e && e[0] == '0'); }();
    return on;
}

bool gemm_dw4_eligible(const refil_gemm_desc& d) {
    const int f = d.flags;
    if (!(f & REFIL_GEMM_A_OUTC) || !(f & REFIL_GEMM_B_OUTC)) return false;
    if (f & (REFIL_GEMM_RELU | REFIL_GEMM_RELU_BWD)) return false;
    if (d.splits < 2 || !d.partial) return false;
    if (d.N < 16 || d.K < 2048) return false;         // (bias-like outputs / short reductions: the LDS-tiled kernel)
    // Measured on MI355X (cfg-T shapes): the big wave tiles win when the output is large enough to give every CU a
    // workgroup with a long row range (the four hypernets' in_trans / fc1 gradients: 181 -> 119, 70 -> 52, 96 -> 70 us);
    // small outputs (agent nets, GRU, thin tails) are faster on the 2 x 2-tile streaming kernel / the LDS-tiled one.
    // (re-swept after the ring fix, tools/sweep.sh: with fewer than ~64 k reduction rows -- cfg2, the per-net launches -- the thin
    // outputs are faster here as well: cfg2 0.767 -> 0.749 ms; with more rows everything from 10 k outputs on: cfg-T 1.750 -> 1.745)
    static const long min_out = [] { const char* e = getenv("REFIL_DW4_MIN_OUT"); return e ? atol(e) : -1L; }();
    const long min_out_t = g_tuning.dw4_min_out >= 0 ? g_tuning.dw4_min_out : min_out;
    if ((long)d.batch * d.M * d.N < (min_out_t >= 0 ? min_out_t : (d.K >= 65536 ? 10000L : 2000L))) return false;
    int ti, tj, wmt;
    dw4_shape(d.M, d.N, ti, tj, wmt);
    // the operand vectors (ti resp. tj consecutive floats; 3 floats need 4-byte alignment only) must be naturally aligned
    auto vec_ok = [](const float* p, int ld, long sb, int t, int n) {
        const int al = t == 4 ? 4 : (t == 2 ? 2 : 1);
        return n >= t && (reinterpret_cast<uintptr_t>(p) % (4 * al)) == 0 && ld % al == 0 && sb % al == 0;
    };
    if (!vec_ok(d.A, d.lda, d.sA, ti, d.M) || !vec_ok(d.B, d.ldb, d.sB, tj, d.N)) return false;
    // 32-bit element offsets, 24-bit multiplies, multiplicative row-map quotients
    auto map_ok = [&](const refil_rowmap& m, int ld) {
        const long rows = (long)d.K + 64;
        const long phys = m.grp ? (rows / m.grp + 1) * (long)m.gstride + m.grp + m.off : rows;
        return ld < (1 << 24) && phys < (1L << 24) && phys * ld < (1L << 32) && (!m.grp || (rows * m.grp < (1L << 32) && m.gstride < (1 << 24)));
    };
    if (!map_ok(d.a_map, d.lda) || !map_ok(d.b_map, d.ldb)) return false;
    if (d.row_index && !d.row_count) return false;
    return true;
}

static bool dws_eligible(const refil_gemm_desc& d);

// reduction splits of a weight-gradient launch (d.M x d.N outputs, d.K rows, d.batch nets): workgroups = tiles x splits
int gemm_dw4_splits(const refil_gemm_desc& d) {
    const long R = d.K;
    if (dws_eligible(d)) {
        // the bf16 x 6 kernel: one workgroup per 128 ti x (<= 128) output tile and split. Alone on the GPU (tools/dws_bench.py) the four
        // hypernets' K / V gradient takes 192 us on 64 workgroups, 119 on 128, 93 on 256 -- a workgroup's pace (1.1 us per 16-row step) is
        // its own instruction stream's, not the memory system's: cache-resident operands change nothing. Inside the step the order is the
        // other way round (tools/sweep.sh, one box, profiles/r05_dws_target.txt): 64 workgroups 1.406 ms, 128 1.415, 192 1.423, 256 1.426 --
        // these launches run beside three other streams, what they cost the step is their CU-time, and the narrow launch leaves the other
        // CUs to the chains. REFIL_DWS_TARGET / refil_set_tuning("dws_target") = the workgroups a launch aims for
        const int ti = (d.M % 256) == 0 ? 2 : 1;
        const long tiles = (long)(d.M / (128 * ti)) * d.batch;
        static const long target = [] { const char* e = getenv("REFIL_DWS_TARGET"); return e ? atol(e) : 64L; }();
        long splits = max(2L, (g_tuning.dws_target > 0 ? g_tuning.dws_target : target) / tiles);
        splits = min(splits, max(2L, R / 1024));          // >= 1024 rows (16 ring periods) per workgroup
        return (int)splits;
    }
    int ti, tj, wmt;
    dw4_shape(d.M, d.N, ti, tj, wmt);
    const long tiles = (long)cdiv(d.M, 32 * ti * wmt) * cdiv(d.N, 32 * tj) * d.batch;
    // (swept with the step's four streams running, tools/sweep.sh: 128 workgroups beat one per CU by 1 % of the step, 64 lose 5 %)
    static const long target = [] { const char* e = getenv("REFIL_DW4_TARGET"); return e ? atol(e) : 128L; }();
    long splits = max(2L, (g_tuning.dw4_target > 0 ? g_tuning.dw4_target : target) / tiles);
    splits = min(splits, max(2L, R / 512));          // >= 512 rows per workgroup: the LDS reduction + partial tile are amortised
    return (int)splits;
}

template <int TI, int TJ, int WMT>
static int dw4_launch_w(const Dw4K& k, dim3 grid, hipStream_t st) {
    constexpr size_t smem = (size_t)(4 - WMT) * dw4_pass(TI, TJ) * 64 * sizeof(float) + 16;
    static bool raised = false;
    if (!raised) {
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_dw4_kernel<TI, TJ, WMT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_dw4_kernel<TI, TJ, WMT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        raised = true;
    }
    if (k.ridx) hipLaunchKernelGGL((gemm_dw4_kernel<TI, TJ, WMT, true>), grid, dim3(256), smem, st, k);
    else hipLaunchKernelGGL((gemm_dw4_kernel<TI, TJ, WMT, false>), grid, dim3(256), smem, st, k);
    return 0;
}
template <int TI, int TJ>
static int dw4_launch_t(const Dw4K& k, int wmt, dim3 grid, hipStream_t st) {
    if (TI == 4 && wmt == 4) return dw4_launch_w<TI, TJ, (TI == 4 ? 4 : 1)>(k, grid, st);
    if (TI >= 2 && wmt == 2) return dw4_launch_w<TI, TJ, (TI >= 2 ? 2 : 1)>(k, grid, st);
    return dw4_launch_w<TI, TJ, 1>(k, grid, st);
}

// REFIL_DW_SPLIT=0 / refil_set_tuning("dw_split", 0): the fp32-instruction kernel for every shape
static bool dws_on() {
    if (g_tuning.dw_split >= 0) return g_tuning.dw_split == 6;
    static const bool env = [] { const char* e = getenv("REFIL_DW_SPLIT"); return !(e && atoi(e) == 0); }();
    return env;
}
static bool dws_eligible(const refil_gemm_desc& d) {
    // (outputs 65 .. 128 columns wide. Two column tiles -- 33 .. 64 columns: the recurrent / tail layers, cfg2's in_trans -- were built and
    // measured in round 5 (tools/dws_bench.py "thin" shapes, profiles/r05_dws_target.txt): 12 TI MFMAs per step leave the launch bound by
    // the 3-way split, and the fp32-instruction kernels are as fast or faster there: 256 x 52: 48 vs 58 us on 64 workgroups but 49 vs 39
    // on 128; 128 x 64 x 4 nets: 51 vs 44 us)
    if (!dws_on() || d.N > 128 || d.N <= 64 || (d.M % 128) != 0 || d.a_map.grp) return false;
    if (d.K < 4096) return false;                                        // (short reductions: all prologue)
    if (((long)d.K + 64) * d.lda * 4 >= (1L << 31) || ((long)d.K + 64) * d.ldb * 4 >= (1L << 31) || d.lda >= (1 << 22) || d.ldb >= (1 << 22)) return false;
    if (d.b_map.grp) {        // physical row = r + (r / grp) (gstride - grp) + off: multiplier and the furthest offset must fit the 24-bit multiplies / 2 GB
        const long delta = (long)d.b_map.gstride - d.b_map.grp;
        const long phys = ((long)d.K + 64) / d.b_map.grp * d.b_map.gstride + d.b_map.grp + d.b_map.off;
        if (delta < 0 || delta * d.ldb * 4 >= (1L << 24) || phys * d.ldb * 4 >= (1L << 31) || d.b_map.off < 0) return false;
    }
    if (d.row_index && (reinterpret_cast<uintptr_t>(d.row_index) & 15)) return false;
    return true;
}
template <int TI, int NJ>
static int dws_launch_t(const Dw4K& k, bool bmap, dim3 grid, hipStream_t st) {
    constexpr size_t smem = dws_smem(TI, NJ);
    static bool raised = false;
    if (!raised) {
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_dws_kernel<TI, NJ, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_dws_kernel<TI, NJ, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_dws_kernel<TI, NJ, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_dws_kernel<TI, NJ, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        raised = true;
    }
    if (k.ridx && bmap) hipLaunchKernelGGL((gemm_dws_kernel<TI, NJ, true, true>), grid, dim3(256), smem, st, k);
    else if (k.ridx) hipLaunchKernelGGL((gemm_dws_kernel<TI, NJ, true, false>), grid, dim3(256), smem, st, k);
    else if (bmap) hipLaunchKernelGGL((gemm_dws_kernel<TI, NJ, false, true>), grid, dim3(256), smem, st, k);
    else hipLaunchKernelGGL((gemm_dws_kernel<TI, NJ, false, false>), grid, dim3(256), smem, st, k);
    return 0;
}

int gemm_dw4_launch(const refil_gemm_desc& d, hipStream_t st) {
    Dw4K k;
    k.A = d.A; k.B = d.B; k.partial = d.partial;
    k.M = d.M; k.N = d.N; k.R = d.K; k.lda = d.lda; k.ldb = d.ldb; k.sA = d.sA; k.sB = d.sB;
    auto mk = [](const refil_rowmap& m) {
        const unsigned grp = m.grp ? m.grp : (1u << 30);
        return RowMap32{grp, m.grp ? (unsigned)m.gstride : 0u, m.grp ? (unsigned)m.off : 0u, (unsigned)(((1ull << 32) + grp - 1) / grp)};
    };
    k.amap = mk(d.a_map); k.bmap = mk(d.b_map);
    k.splits = d.splits; k.batch = d.batch; k.colsum = (d.flags & REFIL_GEMM_COLSUM_A) ? 1 : 0;
    k.ridx = d.row_index; k.rcount = d.row_index ? d.row_count : nullptr;
    static const bool dbg = getenv("REFIL_DEBUG_DW") != nullptr;
    if (dbg) fprintf(stderr, "dw4: M=%d N=%d R=%d batch=%d splits=%d lda=%d ldb=%d amap=(%d,%d,%d) bmap=(%d,%d,%d) list=%d colsum=%d dws=%d\n", d.M, d.N, d.K, d.batch,
                     d.splits, d.lda, d.ldb, d.a_map.grp, (int)d.a_map.gstride, (int)d.a_map.off, d.b_map.grp, (int)d.b_map.gstride, (int)d.b_map.off,
                     d.row_index ? 1 : 0, (d.flags & REFIL_GEMM_COLSUM_A) ? 1 : 0, dws_eligible(d) ? 1 : 0);
    if (dws_eligible(d)) {
        const int ti = (d.M % 256) == 0 ? 2 : 1, nj = cdiv(d.N, 32);
        dim3 grid(1, d.M / (128 * ti), d.batch * d.splits);
        static thread_local char names[16][40];
        static thread_local int n_names = 0;
        char nm[40];
        snprintf(nm, sizeof(nm), "gemm_dws_kernel<%d,%d,%d,%d>", ti, nj, d.row_index ? 1 : 0, d.b_map.grp ? 1 : 0);
        const char* pname = nullptr;
        for (int i = 0; i < n_names; ++i)
            if (!strcmp(names[i], nm)) pname = names[i];
        if (!pname && n_names < 16) { strcpy(names[n_names], nm); pname = names[n_names++]; }
        if (!pname) pname = "gemm_dws_kernel";
        ProfScope prof(pname, 2.0 * d.M * d.N * d.K * d.batch, 4.0 * d.batch * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N), st,
                       d.row_index ? d.row_count : nullptr, (double)d.K, 2.0 * d.M * d.N * d.K * d.batch);
        const bool bm = d.b_map.grp != 0;
        const int rc = ti == 2 ? (nj == 4 ? dws_launch_t<2, 4>(k, bm, grid, st) : dws_launch_t<2, 3>(k, bm, grid, st))
                               : (nj == 4 ? dws_launch_t<1, 4>(k, bm, grid, st) : dws_launch_t<1, 3>(k, bm, grid, st));
        if (rc) return rc;
        REFIL_LAUNCH_CHECK();
        return 0;
    }
    int ti, tj, wmt;
    dw4_shape(d.M, d.N, ti, tj, wmt);
    dim3 grid(cdiv(d.N, 32 * tj), cdiv(d.M, 32 * ti * wmt), d.batch * d.splits);
    static thread_local char names[48][40];
    static thread_local int n_names = 0;
    char nm[40];
    snprintf(nm, sizeof(nm), "gemm_dw4_kernel<%d,%d,%d,%d>", ti, tj, wmt, d.row_index ? 1 : 0);
    const char* pname = nullptr;
    for (int i = 0; i < n_names; ++i)
        if (!strcmp(names[i], nm)) pname = names[i];
    if (!pname && n_names < 48) { strcpy(names[n_names], nm); pname = names[n_names++]; }
    if (!pname) pname = "gemm_dw4_kernel";
    ProfScope prof(pname, 2.0 * d.M * d.N * d.K * d.batch, 4.0 * d.batch * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N), st,
                   d.row_index ? d.row_count : nullptr, (double)d.K);
    int rc = 1;
#define CASE(I, J) if (ti == I && tj == J) rc = dw4_launch_t<I, J>(k, wmt, grid, st)
    CASE(1, 1); CASE(1, 2); CASE(1, 3); CASE(1, 4); CASE(2, 1); CASE(2, 2); CASE(2, 3); CASE(2, 4);
    CASE(3, 1); CASE(3, 2); CASE(3, 3); CASE(3, 4); CASE(4, 1); CASE(4, 2); CASE(4, 3);
#undef CASE
    if (rc) return rc;
    REFIL_LAUNCH_CHECK();
    return 0;
}

}  // namespace refil
