// Weight-resident GEMM for the projections with a SHORT reduction (<= 256):
//   forward   y  = act(x W^T + b)            every nn.Linear forward of the REFIL nets (fc1, in_trans, out_trans,
//                                            fc2, GRU input gates; entity_rnn_agent.py:38-58, attention.py:46,65,
//                                            flex_qmix.py:41-50)
//   backward  dx = dy W  [* relu'(x)] [+ dx] the matching input gradients (autograd of the same layers)
//
// Why a second GEMM kernel. With a reduction of 64..256 a tiled GEMM writes an output tile after only 2-8 K tiles,
// so its main loop never reaches steady state: per K tile it pays a global->register->LDS hop for BOTH operands and
// two workgroup barriers, and the 8 waves of a tile stall on them together (gemm.hip reaches 60-85 TFLOP/s on these
// shapes, rocBLAS / hipBLASLt 36-87). Here the roles are split instead:
//   * W (<= 128 x 128 or 64 x 256 fp32 = 64 KB) is staged ONCE per workgroup into LDS and stays there (for the
//     backward product it is transposed while staging); workgroups are persistent (one per CU) and loop over row
//     tiles, so the staging is amortised over ~10 tiles per wave;
//   * x / dy never touches LDS: in the v_mfma_f32_32x32x2_f32 A-operand layout lane l owns row l%32, so each lane
//     streams its own row with 16-byte loads straight into the registers the MFMA reads. The k order of an MFMA is a
//     free permutation as long as A and B agree: lane (row, half) takes k = 8c + 4*half + {0..3} of chunk c, and
//     reads W[col][8c + 4*half ..+3] with one conflict-free ds_read_b128;
//   * waves are independent: no __syncthreads in the main loop. As soon as chunk c has been issued to the MFMAs its
//     registers are refilled with the chunk needed one pass later (second half of the row, or the wave's next tile),
//     so HBM latency hides behind ~16k MFMA cycles without a second register buffer;
//   * the 32x32 accumulator tiles are transposed through a wave-private LDS slab into 16-byte row-contiguous
//     stores with bias / ReLU / row mask, or relu'(aux) and accumulate, fused.
// The loop is straight-line code (compile-time chunk counts, whole tiles only, no predicated memory operation):
// gfx9 counts loads and stores in ONE counter (vmcnt) and the compiler has to assume they complete out of order, so
// every place where a load is awaited while a store is in flight degenerates to vmcnt(0) -- see the loop tail.
#include <string.h>

#include "bufops.h"
#include "common.h"
#include "kernels.h"
#include "profile.h"
#include "split.h"

namespace refil {

#ifdef REFIL_WR_TIMING
// debug build only (REFIL_EXTRA_FLAGS=-DREFIL_WR_TIMING; tools/probes/wres_timing.py): per-wave cycle sums (shader clock) of the
// tile phases, read back through refil_debug_wres_timing
__device__ unsigned long long g_wr_dbg[8192 * 4];
#define WR_TICK(acc_) { const unsigned long long t1_ = __builtin_readcyclecounter(); acc_ += t1_ - t0_; t0_ = t1_; }
#else
#define WR_TICK(acc_)
#endif

struct WresK {
    const float* A; const float* W; float* C; const float* bias; const uint8_t* rowmask; const float* aux;
    int M, N, K, lda, ldw, ldc;
    long sA, sW, sC, sBias;
    RowMap amap, cmap;
    int rowmask_mod, relu, ntiles;
    const float* bias2; const float* rowscale; int rowscale_mod;   // EPI 0: + rowscale[r % mod] * bias2
    const int* ridx; const int* rcount;     // IDX: logical row r < *rcount lives at row ridx[r] (before amap / cmap); the list is
                                            // padded to whole 32-row tiles with the index of a scratch row
};

// waves per workgroup. Swept with the step's four streams running (tools/sweep.sh, builds selected with REFIL_LIB_PATH):
// 4 waves beat 8 by 1.2 % of the step (2: +22 %, 3 / 6: +7 %, 16: +40 %) -- one wave per SIMD keeps the matrix pipe fed
// because the x rows stream straight into the operand registers, and the smaller epilogue slabs (18 instead of 37 KB of
// LDS) leave room for an attention workgroup of the other chain on the same CU
#ifndef REFIL_WR_WAVES
#define REFIL_WR_WAVES 4
#endif
constexpr int WR_WAVES = REFIL_WR_WAVES;
constexpr int WR_SLAB_P = 36;      // floats per row of the 32 x 32 epilogue slab

__device__ inline float4 keep_if(bool in, float4 v) {
    v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
    return v;
}

// TN: 32-column tiles per workgroup. NC x NPASS: 8-wide k chunks of the reduction (K <= 8 NC NPASS, zero padded);
// NPASS = 2 walks a row in two halves through the same NC register chunks. BT: W is [reduction][out] in memory
// (backward product) and is transposed while staging. EPI 0: + bias, ReLU, row mask; EPI 1: * relu'(aux) (+ C if ACC).
// B2: the row-scaled second bias of EPI 0 (only instantiated for TN <= 2: it costs registers the 128-wide tile needs).
// IDX: the rows come from an index list whose length is read from device memory (rows that cannot influence the
// step are skipped without a host round trip). A wave fetches the 32 indices of a tile with ONE load per lane, two
// tiles ahead of their use, and the epilogue gets its row indices from the neighbouring lanes (ds_bpermute), so the
// list adds no dependent load to the pipeline.
// bytes of LDS of an instantiation: the W slice + the waves' epilogue slabs
constexpr size_t wres_smem(int tn, int nc, int npass) { return ((size_t)32 * tn * (8 * nc * npass + 4) + (size_t)WR_WAVES * 32 * WR_SLAB_P) * sizeof(float); }

// ---- SPLIT = 6: the fp32 product on the bf16 matrix pipe, without giving up fp32 accuracy --------------------------------------
// Every fp32 operand is written as hi + mid + lo, three bf16 numbers (8 significant bits each, nearest rounding, the residuals
// x - hi and x - hi - mid are exact in fp32): the split is exact up to the last bit of the fp32 significand. A product a b is
// then nine bf16 x bf16 products, each EXACT in the pipe's fp32 accumulate; the three smallest (mid lo, lo mid, lo lo) are below
// 2^-26 |a b| together -- a quarter of the unit roundoff the fp32 accumulation applies to every partial sum anyway -- and are left
// out: v_mfma_f32_32x32x16_bf16 x 6 instead of v_mfma_f32_32x32x2_f32 x 8 for the same 16 reduction indices, 192 instead of 512
// matrix-pipe cycles. (Measured against an fp64 product: tests/test_gpu_ops.py::test_wres_split_accuracy -- the error of the two
// paths is the same.) W is split once per workgroup while it is staged into LDS (three bf16 planes, 1.5 x the fp32 bytes); the x rows
// are split in registers right before use, ~5 VALU operations per element, each element feeding 6 TN MFMAs.
// split layout of the W slice: 3 planes x (32 TN columns) x PB bytes; a column holds, per 16-index super-chunk S and lane half hf,
// the 8 bf16 of k = 16 S + 8 (e >> 2) + 4 hf + (e & 3), e = 0..7 -- the indices lane half hf holds of chunks 2S and 2S + 1
constexpr int wres_nse(int nc, int npass) { return npass * ((nc + 1) / 2); }                 // super-chunks
constexpr int wres_pb(int nc, int npass) { return 32 * wres_nse(nc, npass) + 16; }           // column pitch in bytes (/16 odd)
constexpr size_t wres_smem_split(int tn, int nc, int npass) { return (size_t)3 * 32 * tn * wres_pb(nc, npass) + (size_t)WR_WAVES * 32 * WR_SLAB_P * sizeof(float); }
constexpr size_t wres_smem_x(int tn, int nc, int npass, int split) { return split ? wres_smem_split(tn, nc, npass) : wres_smem(tn, nc, npass); }
// (SPLIT: the double-buffered W planes and the split x pieces cost ~60 registers more: the 256-register cap only for the small shapes)
constexpr int wres_min_wg(int tn, int nc, int npass, int split) { return wres_smem_x(tn, nc, npass, split) <= 80 * 1024 && (!split || tn * nc * npass < 32) ? 2 : 1; }
constexpr bool wres_split_ok(int tn, int nc, int npass) { return wres_smem_split(tn, nc, npass) <= 160 * 1024 && (npass == 1 || nc % 2 == 0); }

// (a second workgroup per CU -- and with it the 256-register cap -- only where two W slices fit the CU's 160 KB of LDS)
template <int TN, int NC, int NPASS, bool BT, int EPI, bool ACC, bool RMASK, bool B2, bool IDX = false, int SPLIT = 0>
__global__ __launch_bounds__(64 * WR_WAVES, wres_min_wg(TN, NC, NPASS, SPLIT)) void gemm_wres_kernel(WresK p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane31 = lane & 31, hf = lane >> 5;
    const int bz = blockIdx.z, n0 = blockIdx.y * 32 * TN;
    constexpr int NCT = NC * NPASS;
    constexpr int KP = 8 * NCT + 4;                    // LDS pitch of a W row: KP/4 odd -> conflict-free b128 reads
    const float* __restrict__ A = p.A + bz * p.sA;
    const float* __restrict__ W = p.W + bz * p.sW;
    float* __restrict__ C = p.C + bz * p.sC;
    const float* __restrict__ AUX = EPI == 1 ? p.aux + bz * p.sC : nullptr;
    float* Ws = lds;
    constexpr int NSE = wres_nse(NC, NPASS), PB = wres_pb(NC, NPASS), PS = 32 * TN * PB;      // (SPLIT) super-chunks, column pitch, plane size [bytes]
    char* Wb = reinterpret_cast<char*>(lds);
    float* slab = (SPLIT ? reinterpret_cast<float*>(Wb + 3 * PS) : lds + 32 * TN * KP) + wave * 32 * WR_SLAB_P;
    const int ntiles = IDX ? (*p.rcount + 31) >> 5 : p.ntiles;
    if (IDX && ntiles == 0) return;                    // (uniform: before any barrier)

    // ---- stage the W slice (32 TN output columns x 8 NCT reduction indices, zero padded) once; loads batched 8 deep ----
    {
        constexpr int NT = 64 * WR_WAVES, U = 8;
        constexpr int NCS = SPLIT ? 2 * NSE : NCT;                                // chunks staged (SPLIT: whole super-chunks, zero padded)
        constexpr int TOTAL = BT ? 8 * NCS * 8 * TN : 32 * TN * 2 * NCS;          // float4 count
        for (int base = 0; base < TOTAL; base += NT * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT + tid;
                if (BT) {        // memory [r][o]: 16 bytes along the output index
                    const int r = idx / (8 * TN), o = (idx - r * (8 * TN)) * 4;
                    const bool in = idx < TOTAL && r < p.K && (n0 + o) < p.N;     // N % 4 == 0
                    v[u] = keep_if(in, *reinterpret_cast<const float4*>(W + (long)(in ? r : 0) * p.ldw + (in ? n0 + o : 0)));
                } else {         // memory [o][r]: 16 bytes along the reduction index
                    const int n = idx / (2 * NCS), k = (idx - n * (2 * NCS)) * 4;
                    const bool in = idx < TOTAL && (n0 + n < p.N) && (k < p.K);   // K % 4 == 0
                    v[u] = keep_if(in, *reinterpret_cast<const float4*>(W + (long)(in ? n0 + n : 0) * p.ldw + (in ? k : 0)));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT + tid;
                if (idx < TOTAL) {
                    if (SPLIT) {
                        unsigned h0, m0, l0, h1, m1, l1;
                        if (BT) {   // four columns o..o+3 at reduction index r: one bf16 per plane and column
                            const int r = idx / (8 * TN), o = (idx - r * (8 * TN)) * 4;
                            wr_split(v[u].x, v[u].y, h0, m0, l0); wr_split(v[u].z, v[u].w, h1, m1, l1);
                            const int off = (2 * (r >> 4) + ((r >> 2) & 1)) * 16 + 2 * (4 * ((r >> 3) & 1) + (r & 3));
                            unsigned short* q0 = reinterpret_cast<unsigned short*>(Wb + (o + 0) * PB + off);
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) {
                                const unsigned w0 = pl == 0 ? h0 : (pl == 1 ? m0 : l0), w1 = pl == 0 ? h1 : (pl == 1 ? m1 : l1);
                                unsigned short* q = q0 + pl * (PS / 2);
                                q[0] = (unsigned short)w0; q[PB / 2] = (unsigned short)(w0 >> 16);
                                q[2 * (PB / 2)] = (unsigned short)w1; q[3 * (PB / 2)] = (unsigned short)(w1 >> 16);
                            }
                        } else {    // four reduction indices k..k+3 of column n: 8 bytes per plane
                            const int n = idx / (2 * NCS), k = (idx - n * (2 * NCS)) * 4;
                            wr_split(v[u].x, v[u].y, h0, m0, l0); wr_split(v[u].z, v[u].w, h1, m1, l1);
                            char* q = Wb + n * PB + (2 * (k >> 4) + ((k >> 2) & 1)) * 16 + 8 * ((k >> 3) & 1);
                            *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
                            *reinterpret_cast<uint2*>(q + PS) = make_uint2(m0, m1);
                            *reinterpret_cast<uint2*>(q + 2 * PS) = make_uint2(l0, l1);
                        }
                    } else if (BT) {
                        const int r = idx / (8 * TN), o = (idx - r * (8 * TN)) * 4;
                        Ws[(o + 0) * KP + r] = v[u].x; Ws[(o + 1) * KP + r] = v[u].y;
                        Ws[(o + 2) * KP + r] = v[u].z; Ws[(o + 3) * KP + r] = v[u].w;
                    } else {
                        const int n = idx / (2 * NCT), k = (idx - n * (2 * NCT)) * 4;
                        *reinterpret_cast<float4*>(Ws + n * KP + k) = v[u];
                    }
                }
            }
        }
    }
    __syncthreads();

    // per-lane constants of the epilogue: this lane stores columns n0 + 32 j + 4 (lane % 8) .. +3 of rows 8 ps + lane / 8
    const int c4 = lane & 7, rsub = lane >> 3;
    float4 bias4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == 0 && p.bias) bias4[j] = *reinterpret_cast<const float4*>(p.bias + bz * p.sBias + n0 + 32 * j + 4 * c4);
    }
    constexpr bool has_b2 = EPI == 0 && B2;
    float4 bias24[B2 ? TN : 1];
#pragma unroll
    for (int j = 0; j < (B2 ? TN : 1); ++j) {
        bias24[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_b2) bias24[j] = *reinterpret_cast<const float4*>(p.bias2 + bz * p.sBias + n0 + 32 * j + 4 * c4);
    }
    const float* wb = Ws + lane31 * KP + 4 * hf;       // + 32 j KP + 8 c
    const bool relu = p.relu != 0;

    const int stride = gridDim.x * WR_WAVES;
    int tile = blockIdx.x * WR_WAVES + wave;
    const int last = ntiles - 1;
    // k >= K (padding of the last chunk): the matching W columns in LDS are zero, so x only has to be FINITE there.
    // No select on the loaded value (it would make the compiler wait for every prefetch at the loop head): the
    // address is clamped to the row's last 16 bytes instead and those x values are multiplied by zeros.
    const int kmax = p.K - 4 - 4 * hf;
    float4 a[NC];
    // IDX: list entries of this wave's current / next tile (row of lane31); the one after next is requested in the loop
    int currow = 0, nrow = 0;
    if (IDX) { currow = p.ridx[min(tile, last) * 32 + lane31]; nrow = p.ridx[min(tile + stride, last) * 32 + lane31]; }
    const float* csrc = A + p.amap(IDX ? currow : min(tile, last) * 32 + lane31) * (long)p.lda + 4 * hf;
#pragma unroll
    for (int c = 0; c < NC; ++c) a[c] = *reinterpret_cast<const float4*>(csrc + min(8 * c, kmax));
    // vmcnt counts loads and stores together, in order, and at the loop head the compiler takes the SMALLER count of the two
    // incoming paths: with no stores behind the prologue's loads the wait for chunk 0 would be vmcnt(NC - 1), which on the
    // back edge (NC refills, then the tile's 4 TN stores) drains everything -- including the refill issued last, a full HBM
    // latency ago at most. As many dropped stores (zero-sized buffer, distinct offsets so that they are not merged) as a tile
    // issues make both paths look the same, and every wait in the loop counts exactly.
    {
        const rsrc_t none = mk_rsrc(p.C, 0);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4 * TN; ++i) buf_st4(none, BUF_OOB - 16 * i, z);
    }

#ifdef REFIL_WR_TIMING
    unsigned long long t_top = 0, t_mf = 0, t_ep = 0, n_t = 0, t0_ = __builtin_readcyclecounter();
#endif
    while (tile < ntiles) {
        const int next = tile + stride;
        int nnrow = 0;
        if (IDX) nnrow = p.ridx[min(next + stride, last) * 32 + lane31];
        // (a wave's last prefetch re-reads its own last tile: always a legal address, never consumed)
        const float* nsrc = A + p.amap(IDX ? nrow : min(next, last) * 32 + lane31) * (long)p.lda + 4 * hf;
        // epilogue addressing and operands of THIS tile, requested before the MFMAs so that nothing in the epilogue
        // waits on memory
        const int m0 = tile * 32;
        long coff[4];
        uint8_t dead[4];
        float rsc[4];
        float4 ax[EPI == 1 ? TN : 1][4], cx[ACC ? TN : 1][4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = IDX ? __shfl(currow, ps * 8 + rsub, 64) : m0 + ps * 8 + rsub;
            coff[ps] = p.cmap(m) * (long)p.ldc + n0 + 4 * c4;
            dead[ps] = 0;
            if (RMASK) dead[ps] = p.rowmask[m % p.rowmask_mod];
            rsc[ps] = 0.f;
            if (has_b2) rsc[ps] = p.rowscale[m % p.rowscale_mod];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (EPI == 1) ax[j][ps] = *reinterpret_cast<const float4*>(AUX + coff[ps] + 32 * j);
                if (ACC) cx[j][ps] = *reinterpret_cast<const float4*>(C + coff[ps] + 32 * j);
            }
        }
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        WR_TICK(t_top)
        if constexpr (SPLIT == 0) {
            // W fragments are read one chunk ahead of the MFMAs that use them (LDS latency off the MFMA issue path)
            float4 bq[2][TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[0][j] = *reinterpret_cast<const float4*>(wb + 32 * j * KP);
#pragma unroll
            for (int ps_ = 0; ps_ < NPASS; ++ps_) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int ct = ps_ * NC + c;
                    if (ct + 1 < NCT) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) bq[(ct + 1) & 1][j] = *reinterpret_cast<const float4*>(wb + 32 * j * KP + 8 * (ct + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler sinks these reads to the end of the chunk)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float4 b = bq[ct & 1][j];
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, b.x, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, b.y, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, b.z, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, b.w, acc[j], 0, 0, 0);
                    }
                    // refill: the second half of this row (two-pass) or the same chunk of the wave's next tile. The
                    // sched_barriers pin the interleaving: left alone the scheduler sinks all loads below the MFMAs.
                    __builtin_amdgcn_sched_barrier(0);
                    // The four chunks that share a 128-byte line of the row are requested TOGETHER, behind the last of them: requested
                    // one per chunk (1024 matrix-pipe cycles apart, with the other waves' rows in between) the line had left the 32 KB
                    // vector cache by the time its next quarter was asked for (measured: 71.0 -> 69.3 us alone, -0.7 % of the step)
                    if ((c & 3) == 3 || c == NC - 1) {
#pragma unroll
                        for (int cc = (c & ~3); cc <= c; ++cc) {
                            if (ps_ + 1 < NPASS) a[cc] = *reinterpret_cast<const float4*>(csrc + min(8 * (ps_ * NC + cc + NC), kmax));
                            else a[cc] = *reinterpret_cast<const float4*>(nsrc + min(8 * cc, kmax));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // bf16 x 6 (see wr_split): per 16-index super-chunk and column tile, 3 LDS reads of 16 bytes (the W planes) feed 6 MFMAs;
            // the planes of the next super-chunk are read, and the x pieces of the next super-chunk are split (~45 VALU operations),
            // while this one's MFMAs run: the MFMAs go round-robin over the TN accumulators, product by product (smallest first per
            // accumulator), so two consecutive ones never depend on each other and the fillers between them cost nothing (a filler
            // between two MFMAs on the SAME accumulator breaks the back-to-back forwarding: DESIGN.md lesson 26).
            constexpr int NSP = (NC + 1) / 2;                      // super-chunks per pass
            const char* wbb = Wb + lane31 * PB + 16 * hf;         // + 32 j PB + 32 s (+ plane PS)
            wr_u32x4 bq[2][TN][3];
            wr_u32x4 as[2][3];                                    // hi / mid / lo pieces of the x operand, this and the next super-chunk
            // The split of one super-chunk (8 x values = 4 pairs) as 20 micro-steps (per pair: hi, residual, mid, residual, lo), so that it
            // can be dealt out in pieces of 1-3 VALU operations behind the MFMAs of the super-chunk before
            float sx[4], sy[4];
            unsigned sh[4], sm[4], sl[4];
            auto split_begin = [&](int sc) {
                const float4 a0 = a[2 * sc < NC ? 2 * sc : 0];
                float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (2 * sc + 1 < NC) a1 = a[2 * sc + 1 < NC ? 2 * sc + 1 : 0];
                sx[0] = a0.x; sy[0] = a0.y; sx[1] = a0.z; sy[1] = a0.w; sx[2] = a1.x; sy[2] = a1.y; sx[3] = a1.z; sy[3] = a1.w;
            };
            auto split_micro = [&](auto k_) {
                constexpr int k = decltype(k_)::value, q = k / 5, t = k % 5;
                if constexpr (t == 0) sh[q] = wr_pk(sx[q], sy[q]);
                else if constexpr (t == 1) { sx[q] = wr_sub(sx[q], __uint_as_float(sh[q] << 16)); sy[q] = wr_sub(sy[q], __uint_as_float(sh[q] & 0xFFFF0000u)); }
                else if constexpr (t == 2) sm[q] = wr_pk(sx[q], sy[q]);
                else if constexpr (t == 3) { sx[q] = wr_sub(sx[q], __uint_as_float(sm[q] << 16)); sy[q] = wr_sub(sy[q], __uint_as_float(sm[q] & 0xFFFF0000u)); }
                else sl[q] = wr_pk(sx[q], sy[q]);
            };
            auto split_end = [&](wr_u32x4* o) {
                o[0] = wr_u32x4{sh[0], sh[1], sh[2], sh[3]}; o[1] = wr_u32x4{sm[0], sm[1], sm[2], sm[3]}; o[2] = wr_u32x4{sl[0], sl[1], sl[2], sl[3]};
            };
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bq[0][j][pl] = *reinterpret_cast<const wr_u32x4*>(wbb + pl * PS + 32 * j * PB);
            split_begin(0);
            static_for<20>(split_micro);
            split_end(as[0]);
#pragma unroll
            for (int ps_ = 0; ps_ < NPASS; ++ps_) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if ((c & 1) || c == NC - 1) {
                        const int sc = c >> 1, st_ = ps_ * NSP + sc;
                        // the next super-chunk's x pieces: the same pass (its chunks are refilled later), or the next pass once its first
                        // chunks have been refilled (NC > 4: the refill of chunks 0..3 follows chunk 3); else behind this one's refills
                        constexpr bool across = NC > 4;
                        const bool nxt_now = sc + 1 < NSP || (ps_ + 1 < NPASS && across);
                        constexpr int NG = 6 * TN;                 // MFMAs of a super-chunk = filler slots
                        __builtin_amdgcn_sched_barrier(0);
                        if (nxt_now) split_begin(sc + 1 < NSP ? sc + 1 : 0);
                        const wr_bf16x8 Ah = __builtin_bit_cast(wr_bf16x8, as[st_ & 1][0]), Am = __builtin_bit_cast(wr_bf16x8, as[st_ & 1][1]),
                                        Al = __builtin_bit_cast(wr_bf16x8, as[st_ & 1][2]);
                        // product order (smallest first; the three of magnitude 2^-16 in the order that frees the W planes early): lo hi,
                        // hi lo, mid mid, mid hi, hi mid, hi hi. The lo planes of W are dead after the second product and the mid planes after
                        // the fifth: the next super-chunk's are read right there (the register allocator puts them into the same registers),
                        // only the hi planes are held twice.
                        static_for<NG>([&](auto g_) {
                            constexpr int g = decltype(g_)::value, pr = g / TN, j = g % TN;
                            constexpr int bpl = pr == 0 ? 0 : (pr == 1 ? 2 : (pr == 2 ? 1 : (pr == 3 ? 0 : (pr == 4 ? 1 : 0))));
                            const wr_bf16x8 Ap = pr == 0 ? Al : ((pr == 2 || pr == 3) ? Am : Ah);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ap, __builtin_bit_cast(wr_bf16x8, bq[st_ & 1][j][bpl]), acc[j], 0, 0, 0);
                            // fillers of this MFMA (the sched_barriers pin them here: left alone the scheduler gathers them behind the
                            // super-chunk's last MFMA): an LDS read of the next planes, a share of the next split
                            if constexpr (pr == 0 || pr == 2 || pr == 5) {
                                constexpr int pl = pr == 0 ? 0 : (pr == 2 ? 2 : 1);
                                if (st_ + 1 < NSE) bq[(st_ + 1) & 1][j][pl] = *reinterpret_cast<const wr_u32x4*>(wbb + pl * PS + 32 * j * PB + 32 * (st_ + 1));
                            }
                            if (nxt_now) {
                                constexpr int k0 = g * 20 / NG, k1 = (g + 1) * 20 / NG;
                                static_for<k1 - k0>([&](auto d_) { split_micro(std::integral_constant<int, k0 + decltype(d_)::value>{}); });
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        });
                        if (nxt_now) split_end(as[(st_ + 1) & 1]);
                    }
                    // refills: as in the fp32 loop (the four chunks of a 128-byte line together, behind the last of them)
                    if ((c & 3) == 3 || c == NC - 1) {
#pragma unroll
                        for (int cc = (c & ~3); cc <= c; ++cc) {
                            if (ps_ + 1 < NPASS) a[cc] = *reinterpret_cast<const float4*>(csrc + min(8 * (ps_ * NC + cc + NC), kmax));
                            else a[cc] = *reinterpret_cast<const float4*>(nsrc + min(8 * cc, kmax));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (c == NC - 1 && ps_ + 1 < NPASS && NC <= 4) {
                        split_begin(0);
                        static_for<20>(split_micro);
                        split_end(as[(ps_ * NSP + (c >> 1) + 1) & 1]);
                    }
                }
            }
        }
        WR_TICK(t_mf)
        // epilogue: one 32 x 32 tile at a time through the wave-private slab (same-wave LDS ops execute in order);
        // the transposed rows stay in registers (they replace the accumulators) until all of them are ready
        float4 outv[TN][4];
        // The epilogue operands stay opaque until here: left alone the compiler evaluates relu'(aux) right behind the loads at
        // the top of the tile (32 comparisons into scalar masks, to free the registers) and waits for them there with
        // vmcnt(0) -- a full memory latency per tile with only the first few MFMAs issued.
#pragma unroll
        for (int j = 0; j < (EPI == 1 ? TN : 0); ++j)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) asm volatile("" : "+v"(ax[j][ps].x), "+v"(ax[j][ps].y), "+v"(ax[j][ps].z), "+v"(ax[j][ps].w));
#pragma unroll
        for (int j = 0; j < (ACC ? TN : 0); ++j)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) asm volatile("" : "+v"(cx[j][ps].x), "+v"(cx[j][ps].y), "+v"(cx[j][ps].z), "+v"(cx[j][ps].w));
        auto finish = [&](float4 v, int j, int ps) {
            if (EPI == 0) {
                v.x += bias4[j].x; v.y += bias4[j].y; v.z += bias4[j].z; v.w += bias4[j].w;
                if (has_b2) {
                    const float4 b2 = bias24[B2 ? j : 0];
                    v.x = fmaf(rsc[ps], b2.x, v.x); v.y = fmaf(rsc[ps], b2.y, v.y);
                    v.z = fmaf(rsc[ps], b2.z, v.z); v.w = fmaf(rsc[ps], b2.w, v.w);
                }
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (RMASK) v = keep_if(dead[ps] == 0, v);
            } else {
                const float4 x = ax[EPI == 1 ? j : 0][ps];
                v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f; v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
                if (ACC) { const float4 o = cx[ACC ? j : 0][ps]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            }
            return v;
        };
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hf) * WR_SLAB_P + lane31] = acc[j][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const float4 v = *reinterpret_cast<const float4*>(slab + (ps * 8 + rsub) * WR_SLAB_P + 4 * c4);
                // SPLIT (the matrix part of a tile is short, the epilogue a third of it): the transposed rows of ALL column tiles are
                // requested before any is touched -- a wave's LDS operations execute in order, so tile j + 1 may be written behind the
                // reads of tile j without waiting for their data: one LDS latency per row tile instead of one per read
                outv[j][ps] = SPLIT ? v : finish(v, j, ps);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) outv[j][ps] = finish(outv[j][ps], j, ps);
        }
        // the stores drain during the next tile's MFMAs; nothing waits on them (the next tile's waits count past them)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                *reinterpret_cast<float4*>(C + coff[ps] + 32 * j) = outv[j][ps];
        tile = next;
        csrc = nsrc;
        if (IDX) { currow = nrow; nrow = nnrow; }
        WR_TICK(t_ep)
#ifdef REFIL_WR_TIMING
        ++n_t;
#endif
    }
#ifdef REFIL_WR_TIMING
    if (lane == 0) {
        const int w = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WR_WAVES + wave;
        if (w < 8192) { g_wr_dbg[4 * w] = t_top; g_wr_dbg[4 * w + 1] = t_mf; g_wr_dbg[4 * w + 2] = t_ep; g_wr_dbg[4 * w + 3] = n_t; }
    }
#endif
}

static inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool gemm_wres_eligible(const refil_gemm_desc& d) {
    const int f = d.flags;
    if (f & (REFIL_GEMM_A_OUTC | REFIL_GEMM_COLSUM_A)) return false;
    const bool bt = f & REFIL_GEMM_B_OUTC, rb = f & REFIL_GEMM_RELU_BWD;
    if ((f & REFIL_GEMM_ACCUM) && !rb) return false;
    if (rb && (!bt || !d.aux || d.bias || d.rowmask || (f & REFIL_GEMM_RELU) || (d.N % 64))) return false;
    if (d.splits != 1 || d.K < 8 || (d.K % 4) != 0) return false;
    // reductions up to 128 (one pass over the row); up to 256 in two passes through the same register chunks: the dX launches through a
    // ReLU, and the plain forward (no row mask / second bias: the fc1 layers of wide entity feature vectors -- 64 entities: E = 148)
    if (d.K > 256) return false;
    if (d.K > 128 && !rb && (bt || d.rowmask || d.bias2)) return false;
    if ((d.lda % 4) || (d.ldb % 4) || (d.sA % 4) || (d.sB % 4) || !al16(d.A) || !al16(d.B)) return false;
    if (d.M < (d.row_index ? 256 : 2048) || (d.N % 32) != 0) return false;  // tiny calls: tiled kernel
    if (d.row_index) {        // row list (padded to whole tiles by its producer)
        if (!d.row_count || (d.rowmask && d.bias2)) return false;
    } else if ((d.M % 32) != 0) return false;                              // whole 32 x 32 tiles only
    if (!al16(d.C) || (d.ldc % 4) || (d.sC % 4)) return false;
    if (d.bias && (!al16(d.bias) || (d.sBias % 4))) return false;
    if (d.bias2 && (!al16(d.bias2) || (d.sBias % 4) || (d.N % 128) == 0)) return false;     // (no B2 instantiation of the 128-wide tile)
    if (rb && !al16(d.aux)) return false;
    return true;
}

// The bf16 x 6 form of the product (see wr_split) is the default wherever its W planes fit the LDS: same accuracy as the fp32 matrix
// instruction (tests/test_gpu_ops.py::test_wres_split_accuracy), 3/8 of its matrix-pipe cycles. REFIL_WRES_SPLIT=0 /
// refil_set_tuning("wres_split", 0): the v_mfma_f32_32x32x2_f32 form.
static int wres_split_mode() {
    if (g_tuning.wres_split >= 0) return g_tuning.wres_split == 6 ? 6 : 0;
    static const int env = [] { const char* e = getenv("REFIL_WRES_SPLIT"); return e && atoi(e) == 0 ? 0 : 6; }();
    return env;
}
bool gemm_wres_split_on() { return wres_split_mode() == 6; }
template <int TN, int NC, int NPASS, bool BT, int EPI, bool ACC, bool RMASK, bool B2, bool IDX, int SPLIT>
static int wres_launch_s(const WresK& k, dim3 grid, hipStream_t st) {
    constexpr size_t smem = wres_smem_x(TN, NC, NPASS, SPLIT);
    static_assert(smem <= 160 * 1024, "W slice + slabs must fit the 160 KB LDS of a CU");
    static bool raised = false;                        // raise the dynamic-LDS cap of this instantiation once
    if (smem > 64 * 1024 && !raised) {
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_wres_kernel<TN, NC, NPASS, BT, EPI, ACC, RMASK, B2, IDX, SPLIT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        raised = true;
    }
    hipLaunchKernelGGL((gemm_wres_kernel<TN, NC, NPASS, BT, EPI, ACC, RMASK, B2, IDX, SPLIT>), grid, dim3(64 * WR_WAVES), smem, st, k);
    return 0;
}
template <int TN, int NC, int NPASS, bool BT, int EPI, bool ACC, bool RMASK, bool B2 = false, bool IDX = false>
static int wres_launch_i(const WresK& k, dim3 grid, hipStream_t st) {
    if constexpr (wres_split_ok(TN, NC, NPASS)) {
        if (wres_split_mode() == 6) return wres_launch_s<TN, NC, NPASS, BT, EPI, ACC, RMASK, B2, IDX, 6>(k, grid, st);
    }
    return wres_launch_s<TN, NC, NPASS, BT, EPI, ACC, RMASK, B2, IDX, 0>(k, grid, st);
}

// forward products: K <= 128, one pass
template <int TN>
static int wres_launch_fwd(const WresK& k, dim3 grid, hipStream_t st) {
    const int nc = cdiv(k.K, 8);
    const bool rm = k.rowmask != nullptr;
#define FWD(NC)                                                                                                     \
    do {                                                                                                            \
        if (k.ridx && TN <= 2 && k.bias2) return wres_launch_i<(TN <= 2 ? TN : 1), NC, 1, false, 0, false, false, true, true>(k, grid, st); \
        if (k.ridx) return rm ? wres_launch_i<TN, NC, 1, false, 0, false, true, false, true>(k, grid, st)          \
                              : wres_launch_i<TN, NC, 1, false, 0, false, false, false, true>(k, grid, st);        \
        if (TN <= 2 && k.bias2)                                                                                     \
            return rm ? wres_launch_i<(TN <= 2 ? TN : 1), NC, 1, false, 0, false, true, true>(k, grid, st)          \
                      : wres_launch_i<(TN <= 2 ? TN : 1), NC, 1, false, 0, false, false, true>(k, grid, st);        \
        return rm ? wres_launch_i<TN, NC, 1, false, 0, false, true>(k, grid, st) : wres_launch_i<TN, NC, 1, false, 0, false, false>(k, grid, st); \
    } while (0)
    if (nc <= 4) FWD(4);
    if (nc <= 8) FWD(8);
    if (nc <= 11) FWD(11);      // K = 84: the fc1 layers at the SC2 shape law
    if (nc <= 16) FWD(16);
#undef FWD
    // 128 < K <= 256: two passes (plain epilogue only, gemm_wres_eligible)
    if (nc <= 24) return k.ridx ? wres_launch_i<TN, 12, 2, false, 0, false, false, false, true>(k, grid, st)
                                : wres_launch_i<TN, 12, 2, false, 0, false, false>(k, grid, st);
    return k.ridx ? wres_launch_i<TN, 16, 2, false, 0, false, false, false, true>(k, grid, st)
                  : wres_launch_i<TN, 16, 2, false, 0, false, false>(k, grid, st);
}
// backward products dx = dy W (+ row mask), reduction <= 128
template <int TN>
static int wres_launch_bwd(const WresK& k, dim3 grid, hipStream_t st) {
    const int nc = cdiv(k.K, 8);
    const bool rm = k.rowmask != nullptr;
#define BWD(NC)                                                                                                     \
    do {                                                                                                            \
        if (k.ridx) return rm ? wres_launch_i<TN, NC, 1, true, 0, false, true, false, true>(k, grid, st)            \
                              : wres_launch_i<TN, NC, 1, true, 0, false, false, false, true>(k, grid, st);          \
        return rm ? wres_launch_i<TN, NC, 1, true, 0, false, true>(k, grid, st) : wres_launch_i<TN, NC, 1, true, 0, false, false>(k, grid, st); \
    } while (0)
    if (nc <= 4) BWD(4);
    if (nc <= 8) BWD(8);
    BWD(16);
#undef BWD
}
// backward products through a ReLU: dx = (dy W) * relu'(aux) (+ dx), reduction <= 256, TN = 2
static int wres_launch_rbwd(const WresK& k, bool acc, dim3 grid, hipStream_t st) {
    const int nc = cdiv(k.K, 8);
#define RB(NC, NP)                                                                                                  \
    do {                                                                                                            \
        if (k.ridx) return acc ? wres_launch_i<2, NC, NP, true, 1, true, false, false, true>(k, grid, st)           \
                               : wres_launch_i<2, NC, NP, true, 1, false, false, false, true>(k, grid, st);         \
        return acc ? wres_launch_i<2, NC, NP, true, 1, true, false>(k, grid, st) : wres_launch_i<2, NC, NP, true, 1, false, false>(k, grid, st); \
    } while (0)
    if (nc <= 8) RB(8, 1);
    if (nc <= 16) RB(16, 1);
    if (nc <= 24) RB(12, 2);
    RB(16, 2);
#undef RB
}

int gemm_wres_launch(const refil_gemm_desc& d, hipStream_t st) {
    WresK k;
    k.A = d.A; k.W = d.B; k.C = d.C; k.bias = d.bias; k.rowmask = d.rowmask; k.aux = d.aux;
    k.M = d.M; k.N = d.N; k.K = d.K; k.lda = d.lda; k.ldw = d.ldb; k.ldc = d.ldc;
    k.sA = d.sA; k.sW = d.sB; k.sC = d.sC; k.sBias = d.sBias;
    auto mk = [](const refil_rowmap& m) { return make_rowmap(m.grp, m.gstride, m.off); };
    k.amap = mk(d.a_map); k.cmap = mk(d.c_map);
    k.rowmask_mod = d.rowmask_mod; k.relu = (d.flags & REFIL_GEMM_RELU) ? 1 : 0;
    k.bias2 = d.bias2; k.rowscale = d.rowscale; k.rowscale_mod = d.rowscale_mod > 0 ? d.rowscale_mod : 1;
    k.ntiles = d.M / 32;
    k.ridx = d.row_index; k.rcount = d.row_index ? d.row_count : nullptr;
    const bool bt = d.flags & REFIL_GEMM_B_OUTC, rb = d.flags & REFIL_GEMM_RELU_BWD;
    int tn = (d.N % 128 == 0) ? 4 : ((d.N % 64 == 0) ? 2 : 1);
    if (rb) tn = 2;                                   // relu'(aux) (+ C) operands of a tile live in registers too
    // chunk counts of the instantiation this shape takes (wres_launch_fwd / _bwd / _rbwd)
    const int nc8 = cdiv(d.K, 8);
    int ncp, npass = 1;
    if (rb) { if (nc8 <= 8) ncp = 8; else if (nc8 <= 16) ncp = 16; else if (nc8 <= 24) { ncp = 12; npass = 2; } else { ncp = 16; npass = 2; } }
    else if (bt) ncp = nc8 <= 4 ? 4 : (nc8 <= 8 ? 8 : 16);
    else if (nc8 <= 16) ncp = nc8 <= 4 ? 4 : (nc8 <= 8 ? 8 : (nc8 <= 11 ? 11 : 16));
    else { ncp = nc8 <= 24 ? 12 : 16; npass = 2; }
    // The bf16 x 6 form keeps THREE planes of the W slice in LDS: a 128-column slice of a long reduction (the fc1 layers of wide entity
    // feature vectors: K = 164 at 48 entities -> 153 KB + slabs) does not fit where a 64-column one does. Two narrower column blocks on the
    // bf16 pipe beat one wide one on the fp32 instruction (cfg5: 96 -> see profiles/r05_cfg5_fc1_split.txt); the x rows' second read is an L2 hit
    // (column blocks of the same rows share an XCD, below)
    if (wres_split_mode() == 6 && !rb)               // (the dX-through-ReLU launches exist as 64-column tiles only)
        while (tn > 1 && !wres_split_ok(tn, ncp, npass) && wres_split_ok(tn / 2, ncp, npass)) tn >>= 1;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 256; }
        return n;
    }();
    // rows the launch is expected to process: the list length of the previous step when the caller passes it (a hint)
    long rows = d.M;
    if (d.row_index && d.row_count_hint > 0) rows = min((long)d.M, (long)d.row_count_hint + d.row_count_hint / 8 + 64);
    const long tiles = cdivl(rows, 32);
    // a (32 x 32 TN) tile occupies a wave for ~4 TN K microseconds: with fewer than ~1.5 tiles per wave of the chip the
    // launch is all prologue and quantisation -- use narrower tiles (more of them) instead
    if (!rb && d.row_index)
        while (tn > 1 && tiles * (d.N / (32 * tn)) * d.batch * 2 < 3L * n_cu * WR_WAVES && !(tn == 2 && d.bias2 && false)) tn >>= 1;
    const int gy = d.N / (32 * tn), gz = d.batch;
    int gx = (int)min(cdivl(tiles, WR_WAVES), (long)max(1, n_cu / (gy * gz)));
    // XCD-aware: workgroups are dealt round-robin to the 8 XCDs by linear id = x + gx (y + gy z). With gx a multiple
    // of 8 the column blocks y = 0, 1, .. that re-read the SAME rows of x (N > 128) land on the same XCD, so the
    // second read is an L2 hit instead of a second trip to HBM.
    if (gy > 1 && gx >= 8) gx &= ~7;
    dim3 grid(gx, gy, gz);
    // profiler name = the kernel symbol as rocprofv3 prints it (template arguments TN,NC,NPASS,BT,EPI,ACC,RMASK,B2,IDX,SPLIT)
    static thread_local char names[64][64];
    static thread_local int n_names = 0;
    char nm[64];
    const bool split = wres_split_mode() == 6 && wres_split_ok(tn, ncp, npass);
    snprintf(nm, sizeof(nm), "gemm_wres_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d,%d>", tn, ncp, npass, bt ? 1 : 0, rb ? 1 : 0, (d.flags & REFIL_GEMM_ACCUM) ? 1 : 0,
             d.rowmask ? 1 : 0, d.bias2 ? 1 : 0, d.row_index ? 1 : 0, split ? 6 : 0);
    const char* pname = nullptr;
    for (int i = 0; i < n_names; ++i)
        if (!strcmp(names[i], nm)) pname = names[i];
    if (!pname && n_names < 64) { strcpy(names[n_names], nm); pname = names[n_names++]; }
    if (!pname) pname = "gemm_wres_kernel";
    ProfScope prof(pname, 2.0 * d.M * d.N * d.K * d.batch,
                   4.0 * d.batch * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N * (rb ? ((d.flags & REFIL_GEMM_ACCUM) ? 3.0 : 2.0) : 1.0)), st,
                   d.row_index ? d.row_count : nullptr, (double)d.M, split ? 2.0 * d.M * d.N * d.K * d.batch : 0.0);
    int rc;
    if (rb) rc = wres_launch_rbwd(k, (d.flags & REFIL_GEMM_ACCUM) != 0, grid, st);
    else if (bt) rc = tn == 4 ? wres_launch_bwd<4>(k, grid, st) : (tn == 2 ? wres_launch_bwd<2>(k, grid, st) : wres_launch_bwd<1>(k, grid, st));
    else rc = tn == 4 ? wres_launch_fwd<4>(k, grid, st) : (tn == 2 ? wres_launch_fwd<2>(k, grid, st) : wres_launch_fwd<1>(k, grid, st));
    if (rc) return rc;
    REFIL_LAUNCH_CHECK();
    return 0;
}

}  // namespace refil

#ifdef REFIL_WR_TIMING
extern "C" int refil_debug_wres_timing(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(refil::g_wr_dbg), (size_t)n * 4 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
