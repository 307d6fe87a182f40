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
#include "common.h"
#include "profile.h"
#include "../../include/refil_hip.h"

namespace refil {

struct GemmK {
    const float* A; const float* B; float* C;
    const float* bias; const float* aux; const uint8_t* rowmask; float* colsum; float* partial;
    int M, N, K, lda, ldb, ldc;
    long sA, sB, sC, sBias, sColsum;
    RowMap amap, bmap, cmap;
    int rowmask_mod, batch, splits, flags;
    int vecA, vecB;   // 16-byte global loads legal for this operand
};

constexpr int BK = 32;
constexpr int PITCH_RED = BK + 4;   // floats

template <int ROWS, bool OUTC>
struct Tile {
    static constexpr int NV = ROWS * BK / 4 / 256;           // float4 per thread
    static constexpr int PITCH = OUTC ? ROWS + 4 : PITCH_RED;
    static constexpr int FLOATS = OUTC ? BK * (ROWS + 4) : ROWS * PITCH_RED;

    // global -> registers. `o0` first output index of the tile, `k0` first reduction index,
    // OUT/kend bounds. Element (o,k) lives at base[map(o)*ld + k] (RED) or base[map(k)*ld + o] (OUTC).
    __device__ static inline void load(float4 (&v)[NV], const float* __restrict__ base, int ld, const RowMap& map,
                                       int o0, int OUT, int k0, int kend, int vec, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * 256;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!OUTC) {
                const int row = f >> 3, c4 = f & 7;
                const int o = o0 + row, k = k0 + c4 * 4;
                if (o < OUT && k < kend) {
                    const float* p = base + map(o) * (long)ld + k;
                    if (vec && k + 3 < kend) {
                        r = *reinterpret_cast<const float4*>(p);
                    } else {
                        r.x = p[0];
                        if (k + 1 < kend) r.y = p[1];
                        if (k + 2 < kend) r.z = p[2];
                        if (k + 3 < kend) r.w = p[3];
                    }
                }
            } else {
                constexpr int O4 = ROWS / 4;
                const int red = f / O4, o4 = f % O4;
                const int k = k0 + red, o = o0 + o4 * 4;
                if (k < kend && o < OUT) {
                    const float* p = base + map(k) * (long)ld + o;
                    if (vec && o + 3 < OUT) {
                        r = *reinterpret_cast<const float4*>(p);
                    } else {
                        r.x = p[0];
                        if (o + 1 < OUT) r.y = p[1];
                        if (o + 2 < OUT) r.z = p[2];
                        if (o + 3 < OUT) r.w = p[3];
                    }
                }
            }
            v[i] = r;
        }
    }

    __device__ static inline void store(const float4 (&v)[NV], float* lds, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * 256;
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
template <int WAVES_M, int WAVES_N, int TM, int TN, bool A_OUTC, bool B_OUTC, int EPI>
__global__ __launch_bounds__(256, 3) void gemm_kernel(GemmK p) {
    constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
    using TA = Tile<BM, A_OUTC>;
    using TB = Tile<BN, B_OUTC>;
    __shared__ __attribute__((aligned(16))) float lds[TA::FLOATS + TB::FLOATS];
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
    if (kbeg < kend) {
        TA::load(va, A, p.lda, p.amap, m0, p.M, kbeg, kend, p.vecA, tid);
        TB::load(vb, B, p.ldb, p.bmap, n0, p.N, kbeg, kend, p.vecB, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        TA::store(va, As, tid);
        TB::store(vb, Bs, tid);
        __syncthreads();
        if (k0 + BK < kend) {
            TA::load(va, A, p.lda, p.amap, m0, p.M, k0 + BK, kend, p.vecA, tid);
            TB::load(vb, B, p.ldb, p.bmap, n0, p.N, k0 + BK, kend, p.vecB, tid);
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
    if (do_colsum && tid < BM && m0 + tid < p.M) {
        if (EPI == 2) p.partial[(long)p.batch * p.splits * p.M * p.N + ((long)bz * p.splits + sp) * p.M + m0 + tid] = csum;
        else p.colsum[bz * p.sColsum + m0 + tid] = csum;
    }
    float* Cb = EPI == 2 ? p.partial + ((long)bz * p.splits + sp) * p.M * p.N : p.C + bz * p.sC;
    const int ldc = EPI == 2 ? p.N : p.ldc;
    int ncol[TN];
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ncol[j] = n0 + (wn * TN + j) * 32 + lane31;
        bv[j] = (EPI == 0 && p.bias && ncol[j] < p.N) ? p.bias[bz * p.sBias + ncol[j]] : 0.f;
    }
    const bool relu = p.flags & REFIL_GEMM_RELU;
    const bool accum = p.flags & REFIL_GEMM_ACCUM;
    const float* aux = EPI == 1 ? p.aux + bz * p.sC : nullptr;
    // accumulator register r of a 32x32 tile holds row (r&3) + 8*(r>>2) + 4*hf: walk the 4 row quads
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned roff[4];   // element offsets fit 32 bits (checked on the host)
            bool rok[4], dead[4];
            uint8_t mb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + (wm * TM + i) * 32 + q + 8 * c + 4 * hf;
                rok[q] = m < p.M;
                roff[q] = (unsigned)((EPI == 2 ? (long)m : p.cmap(m)) * ldc);
                mb[q] = (EPI == 0 && p.rowmask && rok[q]) ? p.rowmask[m % p.rowmask_mod] : 0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) dead[q] = mb[q] != 0;
            if (EPI == 1) {
                float ax[TN][4], cx[TN][4];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = rok[q] && ncol[j] < p.N;
                        ax[j][q] = ok ? aux[roff[q] + ncol[j]] : 0.f;
                        cx[j][q] = (ok && accum) ? Cb[roff[q] + ncol[j]] : 0.f;
                    }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = (ax[j][q] > 0.f ? acc[i][j][4 * c + q] : 0.f) + cx[j][q];
                        if (rok[q] && ncol[j] < p.N) Cb[roff[q] + ncol[j]] = v;
                    }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = acc[i][j][4 * c + q];
                        if (EPI == 0) {
                            v += bv[j];
                            v = relu ? fmaxf(v, 0.f) : v;
                            v = dead[q] ? 0.f : v;
                        }
                        if (rok[q] && ncol[j] < p.N) Cb[roff[q] + ncol[j]] = v;
                    }
            }
        }
    }
}

// sum the split partials (plain store or +=). 16 waves per workgroup: lane = output element (coalesced),
// wave w adds splits w, w+16, ... (independent loads in flight), then a 16-way LDS reduction.
constexpr int RED_WAVES = 16;
__global__ __launch_bounds__(64 * RED_WAVES) void reduce_partials_kernel(GemmK p) {
    __shared__ float red[RED_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long MN = (long)p.M * p.N;
    const long total = (long)p.batch * MN;
    const long nchunk = cdivl(total, 64);
    const bool with_cs = p.flags & REFIL_GEMM_COLSUM_A;
    const long tot2 = with_cs ? (long)p.batch * p.M : 0;
    const long nchunk2 = cdivl(tot2, 64);
    for (long chunk = blockIdx.x; chunk < nchunk + nchunk2; chunk += gridDim.x) {
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
}

static const char* gemm_name(int wm, int tm, int tn, bool ao, bool bo) {
    static const char* names[3][4] = {
        {"gemm_kernel<2,2,2,2,false,false>", "gemm_kernel<2,2,2,2,false,true>", "gemm_kernel<2,2,2,2,true,false>", "gemm_kernel<2,2,2,2,true,true>"},
        {"gemm_kernel<4,1,1,2,false,false>", "gemm_kernel<4,1,1,2,false,true>", "gemm_kernel<4,1,1,2,true,false>", "gemm_kernel<4,1,1,2,true,true>"},
        {"gemm_kernel<4,1,1,1,false,false>", "gemm_kernel<4,1,1,1,false,true>", "gemm_kernel<4,1,1,1,true,false>", "gemm_kernel<4,1,1,1,true,true>"}};
    const int c = wm == 2 ? 0 : (tn == 2 ? 1 : 2);
    return names[c][(ao ? 2 : 0) + (bo ? 1 : 0)];
}

template <int WM, int WN, int TM, int TN, int EPI>
static void launch_epi(const GemmK& k, dim3 grid, hipStream_t st) {
    const bool ao = k.flags & REFIL_GEMM_A_OUTC, bo = k.flags & REFIL_GEMM_B_OUTC;
    if (!ao && !bo) hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, false, false, EPI>), grid, dim3(256), 0, st, k);
    else if (!ao && bo) hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, false, true, EPI>), grid, dim3(256), 0, st, k);
    else if (ao && bo) hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, true, true, EPI>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((gemm_kernel<WM, WN, TM, TN, true, false, EPI>), grid, dim3(256), 0, st, k);
}

template <int WM, int WN, int TM, int TN>
static void launch_cfg(const GemmK& k, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    dim3 grid(cdiv(k.N, BN), cdiv(k.M, BM), k.batch * k.splits);
    const bool ao = k.flags & REFIL_GEMM_A_OUTC, bo = k.flags & REFIL_GEMM_B_OUTC;
    ProfScope prof(gemm_name(WM, TM, TN, ao, bo), 2.0 * k.M * k.N * k.K * k.batch,
                   4.0 * k.batch * ((double)k.M * k.K + (double)k.N * k.K + (double)k.M * k.N), st);
    if (k.splits > 1) launch_epi<WM, WN, TM, TN, 2>(k, grid, st);
    else if (k.flags & REFIL_GEMM_RELU_BWD) launch_epi<WM, WN, TM, TN, 1>(k, grid, st);
    else launch_epi<WM, WN, TM, TN, 0>(k, grid, st);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int gemm_launch(const refil_gemm_desc& d, hipStream_t st) {
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
    REFIL_CHECK(!d.rowmask || d.rowmask_mod > 0, "refil_gemm: rowmask_mod must be > 0");
    {
        const long rows = d.c_map.grp ? ((long)(d.M - 1) / d.c_map.grp) * d.c_map.gstride + d.c_map.grp + d.c_map.off : d.M;
        REFIL_CHECK(rows * (long)(d.splits > 1 ? d.N : d.ldc) < (1L << 32), "refil_gemm: C exceeds 2^32 elements per batch");
    }
    GemmK k;
    k.A = d.A; k.B = d.B; k.C = d.C; k.bias = d.bias; k.aux = d.aux; k.rowmask = d.rowmask;
    k.colsum = d.colsum; k.partial = d.partial;
    k.M = d.M; k.N = d.N; k.K = d.K; k.lda = d.lda; k.ldb = d.ldb; k.ldc = d.ldc;
    k.sA = d.sA; k.sB = d.sB; k.sC = d.sC; k.sBias = d.sBias; k.sColsum = d.sColsum;
    k.amap = RowMap{d.a_map.grp, d.a_map.gstride, d.a_map.off};
    k.bmap = RowMap{d.b_map.grp, d.b_map.gstride, d.b_map.off};
    k.cmap = RowMap{d.c_map.grp, d.c_map.gstride, d.c_map.off};
    k.rowmask_mod = d.rowmask_mod; k.batch = d.batch; k.splits = d.splits; k.flags = d.flags;
    k.vecA = aligned16(d.A) && (d.lda % 4 == 0) && (d.sA % 4 == 0);
    k.vecB = aligned16(d.B) && (d.ldb % 4 == 0) && (d.sB % 4 == 0);
    if (d.N > 64) launch_cfg<2, 2, 2, 2>(k, st);
    else if (d.N > 32) launch_cfg<4, 1, 1, 2>(k, st);
    else launch_cfg<4, 1, 1, 1>(k, st);
    REFIL_LAUNCH_CHECK();
    if (d.splits > 1) {
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
