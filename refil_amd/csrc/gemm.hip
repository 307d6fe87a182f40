// fp32 matrix-core GEMM for the REFIL projections (nn.Linear forward, dX, dW) on gfx950.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
//
// * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): 4 waves per workgroup, each wave owns a
//   (32*TM) x (32*TN) block of C in accumulator VGPRs.
// * K is consumed in tiles of 32 staged through LDS. An operand whose reduction index is contiguous
//   in memory ("RED": x [rows,in], W [out,in]) is stored [out][32+4] and read with one ds_read_b128
//   per 4 MFMAs (conflict-free: 36*r mod 64 is a bijection on r mod 16); an operand whose OUTPUT
//   index is contiguous ("OUTC": W in dX = dY W, both operands of dW = dY^T X) is stored
//   [32][out+4] and read with conflict-free ds_read_b32. The MFMA k order is permuted identically
//   for A and B (virtual k of MFMA j in group g, lane half hf  ->  k = 8g + 4hf + j).
// * global loads are 16 B per lane and coalesced along whichever index is contiguous; the next
//   K tile is prefetched into registers while the current one is multiplied.
// * the reduction can be split across workgroups (dW: reduction = rows, tens of thousands) into
//   deterministic partial sums that reduce_partials_kernel adds up -- no atomics, bit-reproducible.
//
// Replaces the implicit aten::mm / addmm calls of the reference.
#include <stdlib.h>

#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "profile.h"
#include "kernels.h"

namespace refil {

Tuning g_tuning;

struct GemmK {
    const float* A; const float* B; float* C;
    const float* bias; const float* aux; const uint8_t* rowmask; float* colsum; float* partial;
    int M, N, K, lda, ldb, ldc;
    long sA, sB, sC, sBias, sColsum;
    RowMap amap, bmap, cmap;
    int rowmask_mod, batch, splits, flags;
    int vecA, vecB;   // 16-byte global loads legal for this operand
    int vecC;         // 16-byte stores (and aux / accumulate loads) legal for C
    const float* bias2; const float* rowscale; int rowscale_mod;   // EPI 0: + rowscale[r % mod] * bias2
};

constexpr int BK = 32;
constexpr int PITCH_RED = BK + 4;   // floats

template <int ROWS, bool OUTC, int NT>      // NT = threads per workgroup
struct Tile {
    static constexpr int NV = ROWS * BK / 4 / NT;            // float4 per thread
    static constexpr int PITCH = OUTC ? ROWS + 4 : PITCH_RED;
    static constexpr int FLOATS = OUTC ? BK * (ROWS + 4) : ROWS * PITCH_RED;

    // Per-thread invariants of the NV loads of a tile: the position along the OUTPUT index never changes
    // across K tiles, so its bounds check (and, for reduction-contiguous operands, the row map) is hoisted.
    struct Pos {
        long off[NV];    // RED: map(o)*ld ; OUTC: o
        bool ok[NV];     // o < OUT
    };
    __device__ static inline void prepare(Pos& ps, int ld, const RowMap& map, int o0, int OUT, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT;
            const int o = OUTC ? o0 + (f % (ROWS / 4)) * 4 : o0 + (f >> 3);
            ps.ok[i] = o < OUT;
            const int oc = ps.ok[i] ? o : 0;
            ps.off[i] = OUTC ? (long)oc : map(oc) * (long)ld;
        }
    }

    // global -> registers, BRANCH-FREE: out-of-range elements read a clamped (valid) address and are
    // zeroed by a select, so all NV loads of a tile issue back to back (a divergent if-ladder here
    // serialised the loads: ~8 dependent HBM latencies per K tile). VEC: 16-byte loads are legal
    // (aligned base, ld % 4 == 0, extent of the contiguous dim % 4 == 0); otherwise 4 scalar loads.
    // Element (o,k) lives at base[map(o)*ld + k] (RED) or base[map(k)*ld + o] (OUTC).
    template <bool VEC>
    __device__ static inline void load(float4 (&v)[NV], const float* __restrict__ base, int ld, const RowMap& map,
                                       const Pos& ps, int OUT, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT;
            const int k = OUTC ? k0 + f / (ROWS / 4) : k0 + (f & 7) * 4;
            const bool in = ps.ok[i] && (k < kend);
            const int kc = in ? k : 0;
            const float* p = OUTC ? base + map(kc) * (long)ld + (in ? ps.off[i] : 0) : base + (in ? ps.off[i] : 0) + kc;
            float4 r;
            if (VEC) {
                r = *reinterpret_cast<const float4*>(p);
                r.x = in ? r.x : 0.f; r.y = in ? r.y : 0.f; r.z = in ? r.z : 0.f; r.w = in ? r.w : 0.f;
            } else {
                // tail of the contiguous dim: element e valid iff (contiguous index + e) < limit
                const int c = OUTC ? (int)ps.off[i] : k, lim = OUTC ? OUT : kend;
                const bool i1 = in && c + 1 < lim, i2 = in && c + 2 < lim, i3 = in && c + 3 < lim;
                const float x0 = p[0], x1 = p[i1 ? 1 : 0], x2 = p[i2 ? 2 : 0], x3 = p[i3 ? 3 : 0];
                r.x = in ? x0 : 0.f; r.y = i1 ? x1 : 0.f; r.z = i2 ? x2 : 0.f; r.w = i3 ? x3 : 0.f;
            }
            v[i] = r;
        }
    }

    __device__ static inline void store(const float4 (&v)[NV], float* lds, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT;
            if (!OUTC) {
                const int row = f >> 3, c4 = f & 7;
                *reinterpret_cast<float4*>(lds + row * PITCH_RED + c4 * 4) = v[i];
            } else {
                constexpr int O4 = ROWS / 4;
                const int red = f / O4, o4 = f % O4;
                *reinterpret_cast<float4*>(lds + red * (ROWS + 4) + o4 * 4) = v[i];
            }
        }
    }

    // the 4 operand values of k-group g for the 32 outputs starting at `ob` (lane = l&31, half = l>>5)
    __device__ static inline float4 frag(const float* lds, int ob, int g, int lane31, int hf) {
        if (!OUTC) {
            return *reinterpret_cast<const float4*>(lds + (ob + lane31) * PITCH_RED + g * 8 + hf * 4);
        } else {
            const float* p = lds + (g * 8 + hf * 4) * (ROWS + 4) + ob + lane31;
            return make_float4(p[0], p[ROWS + 4], p[2 * (ROWS + 4)], p[3 * (ROWS + 4)]);
        }
    }
};

// EPI: 0 = store act(acc + bias) (+ optional row mask), 1 = relu-backward (x aux>0, optional +=C),
//      2 = raw partial store of a split reduction. Compile-time so that the epilogue is branch-free and
//      its loads (mask bytes / aux / C) are issued in batches instead of one dependent load per element.
// VEC: both operands admit 16-byte global loads (compile-time so that the loads stay straight-line code).
template <int WAVES_M, int WAVES_N, int TM, int TN, bool A_OUTC, bool B_OUTC, int EPI, bool VEC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_M * WAVES_N == 8 ? 4 : 3) void gemm_kernel(GemmK p) {
    constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N, NT = 64 * WAVES_M * WAVES_N, NW = WAVES_M * WAVES_N;
    using TA = Tile<BM, A_OUTC, NT>;
    using TB = Tile<BN, B_OUTC, NT>;
    constexpr int SLAB_W = 32 * TN, SLAB_P = SLAB_W + 4;          // epilogue staging: one 32 x (32*TN) slab per wave
    constexpr int LDS_FLOATS = (TA::FLOATS + TB::FLOATS) > NW * 32 * SLAB_P ? (TA::FLOATS + TB::FLOATS) : NW * 32 * SLAB_P;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float* As = lds;
    float* Bs = lds + TA::FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lane31 = lane & 31, hf = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z % p.splits;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const float* A = p.A + bz * p.sA;
    const float* B = p.B + bz * p.sB;

    const int kchunk = cdiv(cdiv(p.K, p.splits), BK) * BK;
    const int kbeg = sp * kchunk;
    const int kend = min(p.K, kbeg + kchunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float csum = 0.f;
    const bool do_colsum = (p.flags & REFIL_GEMM_COLSUM_A) && blockIdx.x == 0;

    float4 va[TA::NV], vb[TB::NV];
    typename TA::Pos pa;
    typename TB::Pos pb;
    TA::prepare(pa, p.lda, p.amap, m0, p.M, tid);
    TB::prepare(pb, p.ldb, p.bmap, n0, p.N, tid);
    if (kbeg < kend) {
        TA::template load<VEC>(va, A, p.lda, p.amap, pa, p.M, kbeg, kend, tid);
        TB::template load<VEC>(vb, B, p.ldb, p.bmap, pb, p.N, kbeg, kend, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        TA::store(va, As, tid);
        TB::store(vb, Bs, tid);
        __syncthreads();
        if (k0 + BK < kend) {
            TA::template load<VEC>(va, A, p.lda, p.amap, pa, p.M, k0 + BK, kend, tid);
            TB::template load<VEC>(vb, B, p.ldb, p.bmap, pb, p.N, k0 + BK, kend, tid);
        }
        if (do_colsum && tid < BM) {
            float s = 0.f;
            if (A_OUTC) {
#pragma unroll 8
                for (int k = 0; k < BK; ++k) s += As[k * (BM + 4) + tid];
            } else {
#pragma unroll 8
                for (int k = 0; k < BK; ++k) s += As[tid * PITCH_RED + k];
            }
            csum += s;
        }
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = TA::frag(As, (wm * TM + i) * 32, g, lane31, hf);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = TB::frag(Bs, (wn * TN + j) * 32, g, lane31, hf);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // The MFMA accumulator layout gives a lane ONE column and 16 rows of a 32x32 tile, i.e. 4-byte
    // stores scattered over 16 rows; issued directly they are store-issue bound (measured: 71 -> 103
    // TFLOP/s with the stores removed). So each wave transposes its 32 x (32*TN) slab through LDS and
    // writes whole 16-byte vectors, 16 (TN=2) or 8 (TN=1) lanes per contiguous row segment; bias, ReLU,
    // row masks and the ReLU-backward / accumulate reads are applied on that vector path (coalesced).
    if (do_colsum && tid < BM && m0 + tid < p.M) {
        if (EPI == 2) p.partial[(long)p.batch * p.splits * p.M * p.N + ((long)bz * p.splits + sp) * p.M + m0 + tid] = csum;
        else p.colsum[bz * p.sColsum + m0 + tid] = csum;
    }
    float* Cb = EPI == 2 ? p.partial + ((long)bz * p.splits + sp) * p.M * p.N : p.C + bz * p.sC;
    const int ldc = EPI == 2 ? p.N : p.ldc;
    const float* aux = EPI == 1 ? p.aux + bz * p.sC : nullptr;
    const bool relu = p.flags & REFIL_GEMM_RELU;
    const bool accum = p.flags & REFIL_GEMM_ACCUM;
    float* slab = lds + wave * 32 * SLAB_P;
    constexpr int LPR = SLAB_W / 4;            // lanes per slab row (float4 each)
    constexpr int RPP = 64 / LPR;              // rows per pass
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int ncol = n0 + wn * SLAB_W + c4 * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == 0 && p.bias) {
        const float* bp = p.bias + bz * p.sBias;
        bias4.x = ncol < p.N ? bp[ncol] : 0.f;
        bias4.y = ncol + 1 < p.N ? bp[ncol + 1] : 0.f;
        bias4.z = ncol + 2 < p.N ? bp[ncol + 2] : 0.f;
        bias4.w = ncol + 3 < p.N ? bp[ncol + 3] : 0.f;
    }
    float4 bias24 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == 0 && p.bias2) {
        const float* bp = p.bias2 + bz * p.sBias;
        bias24.x = ncol < p.N ? bp[ncol] : 0.f;
        bias24.y = ncol + 1 < p.N ? bp[ncol + 1] : 0.f;
        bias24.z = ncol + 2 < p.N ? bp[ncol + 2] : 0.f;
        bias24.w = ncol + 3 < p.N ? bp[ncol + 3] : 0.f;
    }
    const bool vecc = p.vecC;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (i > 0) __syncthreads();             // the wave's slab is reused
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hf) * SLAB_P + j * 32 + lane31] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int rr = ps * RPP + rsub;
            const int m = m0 + (wm * TM + i) * 32 + rr;
            const bool rok = m < p.M;
            float4 v = *reinterpret_cast<const float4*>(slab + rr * SLAB_P + c4 * 4);
            const long off = (EPI == 2 ? (long)(rok ? m : 0) : p.cmap(rok ? m : 0)) * ldc + ncol;
            if (EPI == 0) {
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                if (p.bias2) {
                    const float rs = p.rowscale[(rok ? m : 0) % p.rowscale_mod];
                    v.x = fmaf(rs, bias24.x, v.x); v.y = fmaf(rs, bias24.y, v.y); v.z = fmaf(rs, bias24.z, v.z); v.w = fmaf(rs, bias24.w, v.w);
                }
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (p.rowmask) {
                    const bool dead = rok && p.rowmask[m % p.rowmask_mod] != 0;
                    if (dead) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (vecc) {
                const bool ok = rok && ncol < p.N;       // N % 4 == 0 on this path: a vector is all-in or all-out
                if (EPI == 1) {
                    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), cx = ax;
                    if (ok) ax = *reinterpret_cast<const float4*>(aux + off);
                    if (ok && accum) cx = *reinterpret_cast<const float4*>(Cb + off);
                    v.x = (ax.x > 0.f ? v.x : 0.f) + cx.x; v.y = (ax.y > 0.f ? v.y : 0.f) + cx.y;
                    v.z = (ax.z > 0.f ? v.z : 0.f) + cx.z; v.w = (ax.w > 0.f ? v.w : 0.f) + cx.w;
                }
                if (ok) *reinterpret_cast<float4*>(Cb + off) = v;
            } else {
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = rok && ncol + e < p.N;
                    if (EPI == 1) {
                        const float a = ok ? aux[off + e] : 0.f;
                        const float c = (ok && accum) ? Cb[off + e] : 0.f;
                        vv[e] = (a > 0.f ? vv[e] : 0.f) + c;
                    }
                    if (ok) Cb[off + e] = vv[e];
                }
            }
        }
    }
}

// sum the split partials (plain store or +=). 16 waves per workgroup: lane = output element (coalesced),
// wave w adds splits w, w+16, ... (independent loads in flight), then a 16-way LDS reduction.
constexpr int RED_WAVES = 16;
template <class P>
__device__ inline void reduce_chunk(const P& p, long chunk, float (*red)[64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long MN = (long)p.M * p.N;
    const long total = (long)p.batch * MN;
    const long nchunk = cdivl(total, 64);
    const bool with_cs = p.flags & REFIL_GEMM_COLSUM_A;
    const long tot2 = with_cs ? (long)p.batch * p.M : 0;
    const bool cs = chunk >= nchunk;
    const long idx = (cs ? chunk - nchunk : chunk) * 64 + lane;
    const long lim = cs ? tot2 : total;
    float s = 0.f;
    int b = 0, m = 0, n = 0;
    if (idx < lim) {
        const float* src;
        long sstride;
        if (!cs) {
            b = idx / MN;
            const long rem = idx - (long)b * MN;
            m = rem / p.N; n = rem % p.N;
            src = p.partial + (long)b * p.splits * MN + rem;
            sstride = MN;
        } else {
            b = idx / p.M; m = idx % p.M;
            src = p.partial + (long)p.batch * p.splits * MN + (long)b * p.splits * p.M + m;
            sstride = p.M;
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = w;
        for (; sp + 3 * RED_WAVES < p.splits; sp += 4 * RED_WAVES) {
            s0 += src[(long)sp * sstride];
            s1 += src[(long)(sp + RED_WAVES) * sstride];
            s2 += src[(long)(sp + 2 * RED_WAVES) * sstride];
            s3 += src[(long)(sp + 3 * RED_WAVES) * sstride];
        }
        for (; sp < p.splits; sp += RED_WAVES) s0 += src[(long)sp * sstride];
        s = (s0 + s1) + (s2 + s3);
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && idx < lim) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RED_WAVES; ++k) t += red[k][lane];
        if (!cs) {
            float* dst = p.C + b * p.sC + p.cmap(m) * (long)p.ldc + n;
            if (p.flags & REFIL_GEMM_ACCUM) t += *dst;
            *dst = t;
        } else {
            p.colsum[b * p.sColsum + m] = t;
        }
    }
    __syncthreads();
}
template <class P>
__host__ __device__ inline long reduce_chunks(const P& p) {
    return cdivl((long)p.batch * p.M * p.N, 64) + ((p.flags & REFIL_GEMM_COLSUM_A) ? cdivl((long)p.batch * p.M, 64) : 0);
}
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_partials_kernel(GemmK p) {
    __shared__ float red[RED_WAVES][64];
    const long n = reduce_chunks(p);
    for (long chunk = blockIdx.x; chunk < n; chunk += gridDim.x) reduce_chunk(p, chunk, red);
}
// The reductions of many split launches in one launch (the step's weight gradients: nothing but the optimiser reads them,
// so their reductions wait for the end of the step instead of sitting between the GEMMs on the weight-gradient streams).
// Same summation order per output element as reduce_partials_kernel: the results are bit-identical.
struct ReduceMulti { ReduceK r[RED_MULTI]; int first[RED_MULTI + 1]; int n; };
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_multi_kernel(ReduceMulti m) {
    __shared__ float red[RED_WAVES][64];
    const int total = m.first[m.n];
    int i = 0;
    for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        while (chunk >= m.first[i + 1]) ++i;            // (workgroup-uniform: scalar loads from the kernel arguments)
        reduce_chunk(m.r[i], (long)(chunk - m.first[i]), red);
    }
}
int reduce_multi_launch(const ReduceK* r, int n, hipStream_t st) {
    for (int at = 0; at < n; at += RED_MULTI) {
        ReduceMulti m = {};
        m.n = min(RED_MULTI, n - at);
        long chunks = 0, bytes = 0;
        for (int i = 0; i < m.n; ++i) {
            m.r[i] = r[at + i];
            m.first[i] = (int)chunks;
            chunks += reduce_chunks(m.r[i]);
            bytes += 4L * m.r[i].batch * m.r[i].M * m.r[i].N * (m.r[i].splits + 1);
        }
        REFIL_CHECK(chunks < (1L << 31), "refil: reduce_multi: too many output elements");
        m.first[m.n] = (int)chunks;
        if (!chunks) continue;
        ProfScope prof("reduce_multi_kernel", 0.0, (double)bytes, st);
        hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)min(chunks, 4096L)), dim3(64 * RED_WAVES), 0, st, m);
        REFIL_LAUNCH_CHECK();
    }
    return 0;
}

static const char* gemm_name(int wm, int tm, int tn, bool ao, bool bo) {
    static const char* names[3][4] = {
        {"gemm_kernel<2,2,2,2,false,false>", "gemm_kernel<2,2,2,2,false,true>", "gemm_kernel<2,2,2,2,true,false>", "gemm_kernel<2,2,2,2,true,true>"},
        {"gemm_kernel<4,1,1,2,false,false>", "gemm_kernel<4,1,1,2,false,true>", "gemm_kernel<4,1,1,2,true,false>", "gemm_kernel<4,1,1,2,true,true>"},
        {"gemm_kernel<4,1,1,1,false,false>", "gemm_kernel<4,1,1,1,false,true>", "gemm_kernel<4,1,1,1,true,false>", "gemm_kernel<4,1,1,1,true,true>"}};
    if (wm == 1) return "gemm_kernel<1,4,1,1,true,true>";
    if (wm == 2 && tm == 1) return "gemm_kernel<2,2,1,2,true,true>";
    if (wm == 4 && tm == 2) {
        static const char* big[4] = {"gemm_kernel<4,2,2,2,false,false>", "gemm_kernel<4,2,2,2,false,true>",
                                     "gemm_kernel<4,2,2,2,true,false>", "gemm_kernel<4,2,2,2,true,true>"};
        return big[(ao ? 2 : 0) + (bo ? 1 : 0)];
    }
    const int c = wm == 2 ? 0 : (tn == 2 ? 1 : 2);
    return names[c][(ao ? 2 : 0) + (bo ? 1 : 0)];
}

template <int WM, int WN, int TM, int TN, bool AO, bool BO, int EPI>
static void launch_vec(const GemmK& k, dim3 grid, hipStream_t st) {
    if (k.vecA && k.vecB) hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, AO, BO, EPI, true>), grid, dim3(64 * WM * WN), 0, st, k);
    else hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, AO, BO, EPI, false>), grid, dim3(64 * WM * WN), 0, st, k);
}

// Only the operand-layout x epilogue combinations the learner schedule uses are instantiated:
//   x W^T (RED,RED): store | dY W (RED,OUTC): store, relu-bwd | dY^T X (OUTC,OUTC): split partial, store
template <int WM, int WN, int TM, int TN>
static int launch_cfg(const GemmK& k, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    dim3 grid(cdiv(k.N, BN), cdiv(k.M, BM), k.batch * k.splits);
    const bool ao = k.flags & REFIL_GEMM_A_OUTC, bo = k.flags & REFIL_GEMM_B_OUTC;
    const int epi = k.splits > 1 ? 2 : ((k.flags & REFIL_GEMM_RELU_BWD) ? 1 : 0);
    ProfScope prof(gemm_name(WM, TM, TN, ao, bo), 2.0 * k.M * k.N * k.K * k.batch,
                   4.0 * k.batch * ((double)k.M * k.K + (double)k.N * k.K + (double)k.M * k.N), st);
    if (!ao && !bo && epi == 0) launch_vec<WM, WN, TM, TN, false, false, 0>(k, grid, st);
    else if (!ao && bo && epi == 0) launch_vec<WM, WN, TM, TN, false, true, 0>(k, grid, st);
    else if (!ao && bo && epi == 1) launch_vec<WM, WN, TM, TN, false, true, 1>(k, grid, st);
    else if (ao && bo && epi == 2) launch_vec<WM, WN, TM, TN, true, true, 2>(k, grid, st);
    else if (ao && bo && epi == 0) launch_vec<WM, WN, TM, TN, true, true, 0>(k, grid, st);
    else {
        set_error("refil_gemm: unsupported combination (A_OUTC=%d, B_OUTC=%d, epilogue=%d)", (int)ao, (int)bo, epi);
        return 1;
    }
    return 0;
}

// narrow-M tiles for the weight gradients of thin layers (dW[N_out <= 64, K]): only the (OUTC, OUTC) combinations
template <int WM, int WN, int TM, int TN>
static int launch_cfg_dw(const GemmK& k, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    dim3 grid(cdiv(k.N, BN), cdiv(k.M, BM), k.batch * k.splits);
    ProfScope prof(gemm_name(WM, TM, TN, true, true), 2.0 * k.M * k.N * k.K * k.batch,
                   4.0 * k.batch * ((double)k.M * k.K + (double)k.N * k.K + (double)k.M * k.N), st);
    if (k.splits > 1) launch_vec<WM, WN, TM, TN, true, true, 2>(k, grid, st);
    else launch_vec<WM, WN, TM, TN, true, true, 0>(k, grid, st);
    return 0;
}

static bool dw_stream_on() {
    static const bool on = []() { const char* e = getenv("REFIL_GEMM_DWSTREAM"); return !(e && e[0] == '0'); }();
    return on;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int gemm_launch(const refil_gemm_desc& d, hipStream_t st, ReduceK* defer) {
    if (defer) defer->splits = 0;
    static const bool log_shapes = [] { const char* e = getenv("REFIL_GEMM_LOG"); return e && e[0] == '1'; }();
    if (log_shapes)   // debugging aid: one line per launch, in launch order (pair with a rocprofv3 kernel trace)
        fprintf(stderr, "refil_gemm M=%d N=%d K=%d batch=%d splits=%d flags=0x%x\n", d.M, d.N, d.K, d.batch, d.splits, d.flags);
    REFIL_CHECK(d.A && d.B && d.C, "refil_gemm: null operand");
    REFIL_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "refil_gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    REFIL_CHECK(d.batch >= 1 && d.splits >= 1, "refil_gemm: batch/splits must be >= 1");
    REFIL_CHECK(d.splits == 1 || d.partial, "refil_gemm: splits > 1 needs a partial buffer");
    REFIL_CHECK(d.splits == 1 || !(d.flags & (REFIL_GEMM_RELU | REFIL_GEMM_RELU_BWD)) , "refil_gemm: split GEMM has no activation epilogue");
    REFIL_CHECK(d.splits == 1 || (!d.bias && !d.rowmask), "refil_gemm: split GEMM supports no bias / rowmask");
    REFIL_CHECK(!(d.flags & REFIL_GEMM_RELU_BWD) || d.aux, "refil_gemm: RELU_BWD needs aux");
    REFIL_CHECK(d.splits > 1 || !(d.flags & REFIL_GEMM_ACCUM) || (d.flags & REFIL_GEMM_RELU_BWD),
                "refil_gemm: ACCUM is supported with RELU_BWD or with a split reduction");
    REFIL_CHECK(!(d.flags & REFIL_GEMM_RELU_BWD) || (!d.bias && !d.rowmask && !(d.flags & REFIL_GEMM_RELU)),
                "refil_gemm: RELU_BWD excludes bias / rowmask / RELU");
    REFIL_CHECK(!(d.flags & REFIL_GEMM_COLSUM_A) || d.colsum, "refil_gemm: COLSUM_A needs colsum");
    REFIL_CHECK(!d.bias2 || (d.rowscale && d.rowscale_mod > 0 && d.splits == 1 && !(d.flags & (REFIL_GEMM_RELU_BWD | REFIL_GEMM_A_OUTC | REFIL_GEMM_B_OUTC))),
                "refil_gemm: bias2 needs rowscale / rowscale_mod and a plain x W^T product");
    REFIL_CHECK(!d.rowmask || d.rowmask_mod > 0, "refil_gemm: rowmask_mod must be > 0");
    {
        const long rows = max((long)max(d.M, d.N), (long)d.K) + 64;
        REFIL_CHECK(rowmap_exact(d.a_map.grp, rows) && rowmap_exact(d.b_map.grp, rows) && rowmap_exact(d.c_map.grp, rows),
                    "refil_gemm: row map group too large for %ld rows", rows);
    }
    {
        const long rows = d.c_map.grp ? ((long)(d.M - 1) / d.c_map.grp) * d.c_map.gstride + d.c_map.grp + d.c_map.off : d.M;
        REFIL_CHECK(rows * (long)(d.splits > 1 ? d.N : d.ldc) < (1L << 32), "refil_gemm: C exceeds 2^32 elements per batch");
    }
    static const bool wres = []() { const char* e = getenv("REFIL_GEMM_WRES"); return !(e && e[0] == '0'); }();
    if (wres && gemm_wres_eligible(d)) return gemm_wres_launch(d, st);
    const bool dw4 = gemm_dw4_enabled() && gemm_dw4_eligible(d);
    REFIL_CHECK(!d.row_index || dw4 || (dw_stream_on() && gemm_dw_stream_eligible(d)),
                "refil_gemm: row lists need the weight-resident or the streaming-dW kernel (M=%d N=%d K=%d flags=0x%x splits=%d)",
                d.M, d.N, d.K, d.flags, d.splits);
    GemmK k;
    k.A = d.A; k.B = d.B; k.C = d.C; k.bias = d.bias; k.aux = d.aux; k.rowmask = d.rowmask;
    k.colsum = d.colsum; k.partial = d.partial;
    k.M = d.M; k.N = d.N; k.K = d.K; k.lda = d.lda; k.ldb = d.ldb; k.ldc = d.ldc;
    k.sA = d.sA; k.sB = d.sB; k.sC = d.sC; k.sBias = d.sBias; k.sColsum = d.sColsum;
    auto mk = [](const refil_rowmap& m) { return make_rowmap(m.grp, m.gstride, m.off); };
    k.amap = mk(d.a_map); k.bmap = mk(d.b_map); k.cmap = mk(d.c_map);
    k.rowmask_mod = d.rowmask_mod; k.batch = d.batch; k.splits = d.splits; k.flags = d.flags;
    k.bias2 = d.bias2; k.rowscale = d.rowscale; k.rowscale_mod = d.rowscale_mod > 0 ? d.rowscale_mod : 1;
    // 16-byte loads: aligned base/strides and the contiguous extent (K for reduction-contiguous operands,
    // M resp. N for output-contiguous ones) a multiple of 4, so a float4 is never partially valid
    const int extA = (d.flags & REFIL_GEMM_A_OUTC) ? d.M : d.K, extB = (d.flags & REFIL_GEMM_B_OUTC) ? d.N : d.K;
    k.vecA = aligned16(d.A) && (d.lda % 4 == 0) && (d.sA % 4 == 0) && (extA % 4 == 0);
    k.vecB = aligned16(d.B) && (d.ldb % 4 == 0) && (d.sB % 4 == 0) && (extB % 4 == 0);
    if (d.splits > 1) k.vecC = aligned16(d.partial) && (d.N % 4 == 0) && (((long)d.M * d.N) % 4 == 0);
    else k.vecC = aligned16(d.C) && (d.ldc % 4 == 0) && (d.sC % 4 == 0) && (d.N % 4 == 0) &&
                  (!d.aux || aligned16(d.aux));
    int rc;
    static const bool big_tile = []() { const char* e = getenv("REFIL_GEMM_BIG"); return !(e && e[0] == '0'); }();
    // 8 waves, 256x128 tile: wins when the reduction is long enough to amortise the bigger prologue (measured: K>=128
    // shapes +20 %, K=84 shapes -10 %)
    static const long big_min_blocks = []() { const char* e = getenv("REFIL_GEMM_BIG_MINBLK"); return e ? atol(e) : 200L; }();   // fewer big tiles than this leave CUs idle: use 128x128
    const long big_blocks = (long)cdiv(d.M, 256) * cdiv(d.N, 128) * d.batch * d.splits;
    const bool dw_stream = dw_stream_on();
    const bool dw = (d.flags & REFIL_GEMM_A_OUTC) && (d.flags & REFIL_GEMM_B_OUTC) && !(d.flags & (REFIL_GEMM_RELU | REFIL_GEMM_RELU_BWD));
    if (dw4) rc = gemm_dw4_launch(d, st);
    else if (dw_stream && gemm_dw_stream_eligible(d)) rc = gemm_dw_stream_launch(d, st);
    else if (dw && d.M <= 32 && d.N > 64) rc = launch_cfg_dw<1, 4, 1, 1>(k, st);
    else if (dw && d.M <= 64 && d.N > 64) rc = launch_cfg_dw<2, 2, 1, 2>(k, st);
    else if (d.N > 64 && d.M >= 8192 && d.K >= 96 && big_tile && big_blocks >= big_min_blocks) rc = launch_cfg<4, 2, 2, 2>(k, st);
    else if (d.N > 64) rc = launch_cfg<2, 2, 2, 2>(k, st);
    else if (d.N > 32) rc = launch_cfg<4, 1, 1, 2>(k, st);
    else rc = launch_cfg<4, 1, 1, 1>(k, st);
    if (rc) return rc;
    REFIL_LAUNCH_CHECK();
    if (d.splits > 1 && defer) {
        *defer = ReduceK{k.partial, k.C, k.colsum, k.sC, k.sColsum, k.cmap, k.ldc, k.M, k.N, k.batch, k.splits, k.flags};
    } else if (d.splits > 1) {
        const long total = (long)d.batch * d.M * d.N;
        const int blocks = (int)min((long)2048, cdivl(total, 64) + cdivl((long)d.batch * d.M, 64));
        ProfScope prof("reduce_partials_kernel", 0.0, 4.0 * total * (d.splits + 1), st);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(64 * RED_WAVES), 0, st, k);
        REFIL_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace refil

extern "C" int refil_gemm(const refil_gemm_desc* desc, void* stream) {
    REFIL_CHECK(desc, "refil_gemm: null desc");
    return refil::gemm_launch(*desc, (hipStream_t)stream);
}
