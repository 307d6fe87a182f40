// Weight-gradient GEMM  dW[N_out, K_in] = dy[R, N_out]^T x[R, K_in]  (+ db = column sums of dy) for long reductions
// (R = tens of thousands of rows, N_out/K_in <= 512): the autograd of every nn.Linear of the REFIL nets.
//
// The reduction index (the row) is the SLOW index of both operands, so in the v_mfma_f32_32x32x2_f32 operand layouts
// (A: lane -> output row m = lane%32, k = lane/32;  B: lane -> output column n = lane%32, k = lane/32) the 32 lanes
// of a half-wave read 32 CONSECUTIVE floats of one row of dy resp. x: a plain global_load_dword per lane is already
// perfectly coalesced (128 bytes per half-wave) and lands exactly in the register the MFMA reads. So this kernel
// has no LDS staging, no transposition and no workgroup barrier at all: a wave owns a 64 x 64 output tile (2 x 2
// MFMA tiles, 64 accumulator registers) and walks down its row range two rows per step -- 4 loads feed 4 MFMAs --
// with a ring of D prefetched steps (each slot is refilled right after it has been consumed; only loads are ever
// outstanding, so the compiler's vmcnt bookkeeping stays exact). Rows are split over workgroups; the per-split
// partial tiles go to the same partial buffer / reduce_partials_kernel as gemm.hip's split path (deterministic).
// Out-of-range output rows/columns read a clamped address and are simply not stored.
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace refil {

struct DwStreamK {
    const float* A; const float* B; float* partial;
    int M, N, R, lda, ldb;
    long sA, sB;
    RowMap amap, bmap;
    int splits, batch, colsum;
    const int* ridx; const int* rcount;     // IDX: reduction row r < *rcount lives at row ridx[r] of both operands (before the maps)
};

// incremental version of RowMap: physical row offset (in elements) of logical rows r, r+2, r+4, ...
struct RowWalk {
    long off;      // element offset of the current row start
    int rem;       // current row % grp
    int grp; long step2, wrap;
    __device__ inline void init(const RowMap& m, int r, int ld) {
        grp = m.grp;
        const int q = r / m.grp;
        rem = r - q * m.grp;
        off = ((long)q * m.gstride + rem + m.off) * ld;
        step2 = 2L * ld;
        wrap = (long)(m.gstride - m.grp) * ld;
    }
    __device__ inline void advance2() {
        rem += 2; off += step2;
        const bool w = rem >= grp;
        rem -= w ? grp : 0;
        off += w ? wrap : 0;
    }
};

// IDX: the reduction runs over an index list (rows that cannot contribute are skipped); its length is read from device
// memory. The indices travel through their own ring, one ring period ahead of the operand prefetch, so the extra
// indirection never sits between a request and its use.
template <int WM, int WK, int D, bool IDX = false>
__global__ __launch_bounds__(64 * WM * WK) void gemm_dw_stream_kernel(DwStreamK p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane31 = lane & 31, hf = lane >> 5;
    const int wm = wave / WK, wk = wave % WK;
    const int bz = blockIdx.z / p.splits, sp = blockIdx.z % p.splits;
    const int m0 = (blockIdx.y * WM + wm) * 64, n0 = (blockIdx.x * WK + wk) * 64;
    if (m0 >= p.M || n0 >= p.N) return;            // (no barriers in this kernel: whole waves may leave)
    const float* __restrict__ A = p.A + bz * p.sA;
    const float* __restrict__ B = p.B + bz * p.sB;

    // rows of this split: a multiple of 2 D per split so that the pipelined loop runs whole iterations
    const int R = IDX ? *p.rcount : p.R;
    const int chunk = cdiv(cdiv(R, p.splits), 2 * D) * 2 * D;
    const int rbeg = sp * chunk, rend = min(R, rbeg + chunk);
    const int nfull = rend > rbeg ? (rend - rbeg) / (2 * D) : 0;       // iterations of D steps x 2 rows

    int ca[2], cb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ca[i] = min(m0 + 32 * i + lane31, p.M - 1);
        cb[i] = min(n0 + 32 * i + lane31, p.N - 1);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum[2] = {0.f, 0.f};

    if (IDX && nfull > 0) {
        // ri[s]: list entry of the row that slot s is refilled with NEXT (position rbeg + 2 (D it + s) + hf, one
        // period ahead); positions past the split re-read its first entry (never consumed)
        int ri[D];
        const int pend = rbeg + nfull * 2 * D;
        float ra[D][2], rb[D][2];
#pragma unroll
        for (int s = 0; s < D; ++s) ri[s] = p.ridx[rbeg + 2 * s + hf];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const long oa = p.amap(ri[s]) * (long)p.lda, ob = p.bmap(ri[s]) * (long)p.ldb;
#pragma unroll
            for (int i = 0; i < 2; ++i) { ra[s][i] = A[oa + ca[i]]; rb[s][i] = B[ob + cb[i]]; }
            const int pos = rbeg + 2 * (D + s) + hf;
            ri[s] = p.ridx[pos < pend ? pos : rbeg];
        }
        for (int it = 0; it < nfull; ++it) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s][i], rb[s][j], acc[i][j], 0, 0, 0);
                csum[0] += ra[s][0]; csum[1] += ra[s][1];
                __builtin_amdgcn_sched_barrier(0);
                const long oa = p.amap(ri[s]) * (long)p.lda, ob = p.bmap(ri[s]) * (long)p.ldb;
#pragma unroll
                for (int i = 0; i < 2; ++i) { ra[s][i] = A[oa + ca[i]]; rb[s][i] = B[ob + cb[i]]; }
                const int pos = rbeg + 2 * (D * (it + 2) + s) + hf;
                ri[s] = p.ridx[pos < pend ? pos : rbeg];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (!IDX && nfull > 0) {
        RowWalk wa, wb;                            // rows rbeg + hf, + 2, + 4, ... (the PREFETCH position)
        wa.init(p.amap, rbeg + hf, p.lda);
        wb.init(p.bmap, rbeg + hf, p.ldb);
        float ra[D][2], rb[D][2];
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { ra[s][i] = A[wa.off + ca[i]]; rb[s][i] = B[wb.off + cb[i]]; }
            wa.advance2(); wb.advance2();
        }
        const long safe_a = p.amap(rbeg) * (long)p.lda, safe_b = p.bmap(rbeg) * (long)p.ldb;
        for (int it = 0; it < nfull; ++it) {
            const bool more = it + 1 < nfull;      // uniform
#pragma unroll
            for (int s = 0; s < D; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s][i], rb[s][j], acc[i][j], 0, 0, 0);
                csum[0] += ra[s][0]; csum[1] += ra[s][1];
                __builtin_amdgcn_sched_barrier(0);
                // refill the slot with the rows D steps ahead. The last iteration has nothing to prefetch: it re-reads
                // the split's first row instead (a select on the ADDRESS keeps the loop free of branches around loads)
                const long oa = more ? wa.off : safe_a, ob = more ? wb.off : safe_b;
#pragma unroll
                for (int i = 0; i < 2; ++i) { ra[s][i] = A[oa + ca[i]]; rb[s][i] = B[ob + cb[i]]; }
                wa.advance2(); wb.advance2();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // tail rows (fewer than 2 D): plain loop with zero fill
    for (int r2 = rbeg + nfull * 2 * D; r2 < rend; r2 += 2) {      // uniform trip count: MFMAs ignore EXEC
        const int r = r2 + hf;
        const bool ok = r < rend;
        int rr = ok ? r : rbeg;
        if (IDX) rr = p.ridx[rr];
        const long oa = p.amap(rr) * (long)p.lda, ob = p.bmap(rr) * (long)p.ldb;
        float va[2], vb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            va[i] = A[oa + ca[i]]; vb[i] = B[ob + cb[i]];
            va[i] = ok ? va[i] : 0.f; vb[i] = ok ? vb[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i], vb[j], acc[i][j], 0, 0, 0);
        csum[0] += va[0]; csum[1] += va[1];
    }

    // ---- partial tile of this split: partial[(bz * splits + sp)][M][N]; D layout: row (r&3)+8(r>>2)+4hf, column lane31 ----
    float* P = p.partial + ((long)bz * p.splits + sp) * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 32 * j + lane31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (m < p.M && n < p.N) P[(long)m * p.N + n] = acc[i][j][r];
            }
        }
    if (p.colsum && wk == 0 && blockIdx.x == 0) {
        float* CS = p.partial + (long)p.batch * p.splits * p.M * p.N + ((long)bz * p.splits + sp) * p.M;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float t = csum[i] + __shfl_xor(csum[i], 32, 64);      // rows of both halves
            const int m = m0 + 32 * i + lane31;
            if (hf == 0 && m < p.M) CS[m] = t;
        }
    }
}

bool gemm_dw_stream_eligible(const refil_gemm_desc& d) {
    const int f = d.flags;
    if (!(f & REFIL_GEMM_A_OUTC) || !(f & REFIL_GEMM_B_OUTC)) return false;
    if (f & (REFIL_GEMM_RELU | REFIL_GEMM_RELU_BWD)) return false;
    if (d.splits < 2 || !d.partial) return false;
    if (d.row_index) return d.row_count != nullptr;  // a row list is only understood by this kernel
    if (d.M < 96 || d.N < 48) return false;         // thin outputs: the narrow LDS-tiled configurations of gemm.hip
    if (d.K < 4096) return false;
    return true;
}

int gemm_dw_stream_launch(const refil_gemm_desc& d, hipStream_t st) {
    DwStreamK k;
    k.A = d.A; k.B = d.B; k.partial = d.partial;
    k.M = d.M; k.N = d.N; k.R = d.K; k.lda = d.lda; k.ldb = d.ldb; k.sA = d.sA; k.sB = d.sB;
    auto mk = [](const refil_rowmap& m) { return make_rowmap(m.grp, m.gstride, m.off); };
    k.amap = mk(d.a_map); k.bmap = mk(d.b_map);
    k.splits = d.splits; k.batch = d.batch; k.colsum = (d.flags & REFIL_GEMM_COLSUM_A) ? 1 : 0;
    k.ridx = d.row_index; k.rcount = d.row_index ? d.row_count : nullptr;
    const bool idx = d.row_index != nullptr;
    static const bool dbg = getenv("REFIL_DEBUG_DW") != nullptr;
    if (dbg) fprintf(stderr, "dw_stream: M=%d N=%d R=%d batch=%d splits=%d lda=%d ldb=%d amap=(%d,%d,%d) bmap=(%d,%d,%d) list=%d colsum=%d\n", d.M, d.N, d.K, d.batch,
                     d.splits, d.lda, d.ldb, d.a_map.grp, (int)d.a_map.gstride, (int)d.a_map.off, d.b_map.grp, (int)d.b_map.gstride, (int)d.b_map.off,
                     idx ? 1 : 0, (d.flags & REFIL_GEMM_COLSUM_A) ? 1 : 0);
    const bool wide = d.N > 64, tall = d.M >= 256;
    // wide + tall: 8 waves cover 256 x 128 of the output so that x (the B operand) is read once per workgroup -- two
    // 128-row workgroups would land on different XCDs and each fetch its own copy from HBM
    const char* nm = idx ? (wide ? (tall ? "gemm_dw_stream_kernel<4,2,16,1>" : "gemm_dw_stream_kernel<2,2,16,1>") : "gemm_dw_stream_kernel<4,1,16,1>")
                         : (wide ? (tall ? "gemm_dw_stream_kernel<4,2,16>" : "gemm_dw_stream_kernel<2,2,16>") : "gemm_dw_stream_kernel<4,1,16>");
    ProfScope prof(nm, 2.0 * d.M * d.N * d.K * d.batch,
                   4.0 * d.batch * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N), st,
                   idx ? d.row_count : nullptr, (double)d.K);
    if (wide && tall) {
        dim3 grid(cdiv(d.N, 128), cdiv(d.M, 256), d.batch * d.splits);
        if (idx) hipLaunchKernelGGL((gemm_dw_stream_kernel<4, 2, 16, true>), grid, dim3(512), 0, st, k);
        else hipLaunchKernelGGL((gemm_dw_stream_kernel<4, 2, 16>), grid, dim3(512), 0, st, k);
    } else if (wide) {
        dim3 grid(cdiv(d.N, 128), cdiv(d.M, 128), d.batch * d.splits);
        if (idx) hipLaunchKernelGGL((gemm_dw_stream_kernel<2, 2, 16, true>), grid, dim3(256), 0, st, k);
        else hipLaunchKernelGGL((gemm_dw_stream_kernel<2, 2, 16>), grid, dim3(256), 0, st, k);
    } else {
        dim3 grid(1, cdiv(d.M, 256), d.batch * d.splits);
        if (idx) hipLaunchKernelGGL((gemm_dw_stream_kernel<4, 1, 16, true>), grid, dim3(256), 0, st, k);
        else hipLaunchKernelGGL((gemm_dw_stream_kernel<4, 1, 16>), grid, dim3(256), 0, st, k);
    }
    REFIL_LAUNCH_CHECK();
    return 0;
}

}  // namespace refil
