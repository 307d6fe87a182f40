// Weight-resident GEMM for the forward projections  y = act(x W^T + b)  with a SHORT reduction (K <= 128):
// every nn.Linear forward of the REFIL nets (fc1, in_trans, out_trans, fc2, GRU input gates, fc3;
// reference: entity_rnn_agent.py:38-58, attention.py:46,65, flex_qmix.py:41-50).
//
// Why a second GEMM kernel. With K <= 128 a tiled GEMM writes an output tile after only 3-4 K tiles, so its
// main loop never reaches steady state: per K tile it pays a global->register->LDS hop for BOTH operands and
// two workgroup barriers, and the 8 waves of a tile stall on them together (gemm.hip reaches 75-85 TFLOP/s on
// these shapes, as do rocBLAS / hipBLASLt). Here the roles are split instead:
//   * W (<= 128 x 128 fp32 = 64 KB) is staged ONCE per workgroup into LDS and stays there; workgroups are
//     persistent (one per CU) and loop over row tiles, so the staging is amortised over ~10 tiles per wave;
//   * x never touches LDS: in the v_mfma_f32_32x32x2_f32 A-operand layout lane l owns row l%32, so each lane
//     streams its own row with 16-byte loads straight into the registers the MFMA reads. The k order of an
//     MFMA is a free permutation as long as A and B agree: lane (row, half) takes k = 8c + 4*half + {0..3}
//     of chunk c, and reads W[col][8c + 4*half ..+3] with one conflict-free ds_read_b128;
//   * waves are independent: no __syncthreads in the main loop. The next tile's rows are fetched (double
//     buffered in registers) while the current tile multiplies, so HBM latency hides behind ~16k MFMA cycles;
//   * the 32x32 accumulator tiles are transposed through a wave-private LDS slab into 16-byte row-contiguous
//     stores with bias / ReLU / row mask fused (same epilogue contract as gemm.hip EPI 0).
#include "common.h"
#include "kernels.h"
#include "profile.h"

namespace refil {

struct WresK {
    const float* A; const float* W; float* C; const float* bias; const uint8_t* rowmask;
    int M, N, K, lda, ldw, ldc;
    long sA, sW, sC, sBias;
    RowMap amap, cmap;
    int rowmask_mod, relu, ntiles;
};

constexpr int WR_WAVES = 8;
constexpr int WR_SLAB_P = 36;      // floats per row of the 32 x 32 epilogue slab

// select without giving the compiler a reason to put the (always legal, clamped-address) load in a branch:
// a branch around a load makes the s_waitcnt analysis conservative (vmcnt(0) everywhere = no prefetch at all)
__device__ inline float4 keep_if(bool in, float4 v) {
    v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f;
    return v;
}

// TN: 32-column tiles per workgroup; NC: 8-wide k chunks (compile time so that the main loop is straight-line code
// and the compiler can count outstanding loads and stores: K <= 8 NC, zero padded; M % 32 == 0 and N % (32 TN) == 0,
// so there is not a single predicated memory operation in the loop); RMASK: row mask
template <int TN, int NC, bool RMASK>
__global__ __launch_bounds__(64 * WR_WAVES, 2) void gemm_wres_kernel(WresK p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane31 = lane & 31, hf = lane >> 5;
    const int bz = blockIdx.z, n0 = blockIdx.y * 32 * TN;
    constexpr int KP = 8 * NC + 4;                     // LDS pitch of a W row: KP/4 odd -> conflict-free b128 reads
    const float* __restrict__ A = p.A + bz * p.sA;
    const float* __restrict__ W = p.W + bz * p.sW;
    float* __restrict__ C = p.C + bz * p.sC;
    float* Ws = lds;
    float* slab = lds + 32 * TN * KP + wave * 32 * WR_SLAB_P;

    // ---- stage the W slice [n0, n0 + 32 TN) x [0, 8 NC) once (zero padded); loads batched 8 deep ----
    {
        constexpr int KQ = 2 * NC, TOTAL = 32 * TN * KQ, NT = 64 * WR_WAVES, U = 8;
        for (int base = 0; base < TOTAL; base += NT * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT + tid;
                const int n = idx / KQ, k = (idx - n * KQ) * 4;
                const bool in = idx < TOTAL && (n0 + n < p.N) && (k < p.K);       // K % 4 == 0: all-in or all-out
                v[u] = keep_if(in, *reinterpret_cast<const float4*>(W + (long)(in ? n0 + n : 0) * p.ldw + (in ? k : 0)));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT + tid;
                const int n = idx / KQ, k = (idx - n * KQ) * 4;
                if (idx < TOTAL) *reinterpret_cast<float4*>(Ws + n * KP + k) = v[u];
            }
        }
    }
    __syncthreads();

    // per-lane constants of the epilogue: this lane stores columns n0 + 32 j + 4 (lane % 8) .. +3 of rows 8 ps + lane / 8
    const int c4 = lane & 7, rsub = lane >> 3;
    float4 bias4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + 32 * j + 4 * c4;
        bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            const float* bp = p.bias + bz * p.sBias;
            bias4[j].x = col < p.N ? bp[col] : 0.f;
            bias4[j].y = col + 1 < p.N ? bp[col + 1] : 0.f;
            bias4[j].z = col + 2 < p.N ? bp[col + 2] : 0.f;
            bias4[j].w = col + 3 < p.N ? bp[col + 3] : 0.f;
        }
    }
    const float* wb = Ws + lane31 * KP + 4 * hf;       // + 32 j KP + 8 c
    const bool relu = p.relu != 0;

    // persistent wave loop. a[c] holds chunk c of the CURRENT tile; as soon as chunk c has been issued to the MFMAs
    // its registers are refilled with the same chunk of the wave's NEXT tile (one tile of prefetch distance, no
    // second buffer). The sched_barriers pin that interleaving: left alone the scheduler sinks all 16 loads below
    // the MFMAs, right in front of the stores, and the loop head then waits for loads AND stores.
    const int stride = gridDim.x * WR_WAVES;
    int tile = blockIdx.x * WR_WAVES + wave;
    const int last = p.ntiles - 1;
    // k >= K (padding of the last chunk): the matching W columns in LDS are zero, so x only has to be FINITE there.
    // No select on the loaded value (it would make the compiler wait for every prefetch at the loop head): the
    // address is clamped to the row's last 16 bytes instead and those x values are multiplied by zeros.
    const int kmax = p.K - 4 - 4 * hf;
    float4 a[NC];
    {
        const float* src = A + p.amap(min(tile, last) * 32 + lane31) * (long)p.lda + 4 * hf;
#pragma unroll
        for (int c = 0; c < NC; ++c) a[c] = *reinterpret_cast<const float4*>(src + min(8 * c, kmax));
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(a[c].x), "+v"(a[c].y), "+v"(a[c].z), "+v"(a[c].w));   // (see the loop tail)
    }
    while (tile < p.ntiles) {
        const int next = tile + stride;
        // (a wave's last prefetch re-reads its own last tile: always a legal address, never consumed)
        const float* nsrc = A + p.amap(min(next, last) * 32 + lane31) * (long)p.lda + 4 * hf;
        // epilogue addressing / row mask of THIS tile, requested before the MFMAs so that nothing in the epilogue
        // waits on memory
        const int m0 = tile * 32;
        long coff[4];
        uint8_t dead[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int m = m0 + ps * 8 + rsub;
            coff[ps] = p.cmap(m) * (long)p.ldc;
            dead[ps] = 0;
            if (RMASK) dead[ps] = p.rowmask[m % p.rowmask_mod];
        }
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float4 b = *reinterpret_cast<const float4*>(wb + 32 * j * KP + 8 * c);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].x, b.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].y, b.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].z, b.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c].w, b.w, acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a[c] = *reinterpret_cast<const float4*>(nsrc + min(8 * c, kmax));
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: one 32 x 32 tile at a time through the wave-private slab (same-wave LDS ops execute in order);
        // the transposed rows stay in registers (they replace the accumulators) until all of them are ready
        float4 outv[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hf) * WR_SLAB_P + lane31] = acc[j][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                float4 v = *reinterpret_cast<const float4*>(slab + (ps * 8 + rsub) * WR_SLAB_P + 4 * c4);
                v.x += bias4[j].x; v.y += bias4[j].y; v.z += bias4[j].z; v.w += bias4[j].w;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (RMASK) v = keep_if(dead[ps] == 0, v);
                outv[j][ps] = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // gfx9 counts loads and stores in ONE counter and the compiler must assume they complete out of order, so
        // any wait on a load while stores are in flight is a wait on the stores too. Hence: collect the prefetched
        // rows HERE (they had the whole transposition phase to arrive), and only then issue the stores -- they
        // drain during the next tile's MFMAs and nothing waits on them.
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(a[c].x), "+v"(a[c].y), "+v"(a[c].z), "+v"(a[c].w));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                *reinterpret_cast<float4*>(C + coff[ps] + n0 + 32 * j + 4 * c4) = outv[j][ps];
        tile = next;
    }
}

static inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool gemm_wres_eligible(const refil_gemm_desc& d) {
    if (d.flags & ~REFIL_GEMM_RELU) return false;                     // plain x W^T (+bias, +relu, +row mask) only
    if (d.splits != 1 || d.K > 128 || d.K < 8 || (d.K % 4) != 0) return false;
    if ((d.lda % 4) || (d.ldb % 4) || (d.sA % 4) || (d.sB % 4) || !al16(d.A) || !al16(d.B)) return false;
    if (d.M < 2048 || (d.M % 32) != 0 || (d.N % 32) != 0) return false;    // whole 32 x 32 tiles only; tiny calls: tiled kernel
    if (!al16(d.C) || (d.ldc % 4) || (d.sC % 4)) return false;
    return true;
}

template <int TN, int NC, bool RMASK>
static int wres_launch_i(const WresK& k, dim3 grid, hipStream_t st) {
    constexpr size_t smem = ((size_t)32 * TN * (8 * NC + 4) + (size_t)WR_WAVES * 32 * WR_SLAB_P) * sizeof(float);
    static bool raised = false;                        // raise the dynamic-LDS cap of this instantiation once
    if (smem > 64 * 1024 && !raised) {
        REFIL_HIP(hipFuncSetAttribute((const void*)gemm_wres_kernel<TN, NC, RMASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        raised = true;
    }
    hipLaunchKernelGGL((gemm_wres_kernel<TN, NC, RMASK>), grid, dim3(64 * WR_WAVES), smem, st, k);
    return 0;
}
template <int TN, int NC>
static int wres_launch_t(const WresK& k, dim3 grid, hipStream_t st) {
    return k.rowmask ? wres_launch_i<TN, NC, true>(k, grid, st) : wres_launch_i<TN, NC, false>(k, grid, st);
}
template <int TN>
static int wres_launch_n(const WresK& k, dim3 grid, hipStream_t st) {
    const int nc = cdiv(k.K, 8);
    if (nc <= 4) return wres_launch_t<TN, 4>(k, grid, st);
    if (nc <= 8) return wres_launch_t<TN, 8>(k, grid, st);
    if (nc <= 11) return wres_launch_t<TN, 11>(k, grid, st);      // K = 84: the fc1 layers at the SC2 shape law
    return wres_launch_t<TN, 16>(k, grid, st);
}

int gemm_wres_launch(const refil_gemm_desc& d, hipStream_t st) {
    WresK k;
    k.A = d.A; k.W = d.B; k.C = d.C; k.bias = d.bias; k.rowmask = d.rowmask;
    k.M = d.M; k.N = d.N; k.K = d.K; k.lda = d.lda; k.ldw = d.ldb; k.ldc = d.ldc;
    k.sA = d.sA; k.sW = d.sB; k.sC = d.sC; k.sBias = d.sBias;
    auto mk = [](const refil_rowmap& m) { return m.grp ? RowMap{m.grp, m.gstride, m.off} : RowMap{1 << 30, 0, 0}; };
    k.amap = mk(d.a_map); k.cmap = mk(d.c_map);
    k.rowmask_mod = d.rowmask_mod; k.relu = (d.flags & REFIL_GEMM_RELU) ? 1 : 0;
    k.ntiles = cdiv(d.M, 32);
    const int tn = (d.N % 128 == 0) ? 4 : ((d.N % 64 == 0) ? 2 : 1);
    const int gy = cdiv(d.N, 32 * tn), gz = d.batch;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); n = 256; }
        return n;
    }();
    const int gx = min(cdiv(k.ntiles, WR_WAVES), max(1, n_cu / (gy * gz)));
    dim3 grid(gx, gy, gz);
    static const char* names[3] = {"gemm_wres_kernel<1>", "gemm_wres_kernel<2>", "gemm_wres_kernel<4>"};
    ProfScope prof(names[tn == 1 ? 0 : (tn == 2 ? 1 : 2)], 2.0 * d.M * d.N * d.K * d.batch,
                   4.0 * d.batch * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N), st);
    int rc;
    if (tn == 4) rc = wres_launch_n<4>(k, grid, st);
    else if (tn == 2) rc = wres_launch_n<2>(k, grid, st);
    else rc = wres_launch_n<1>(k, grid, st);
    if (rc) return rc;
    REFIL_LAUNCH_CHECK();
    return 0;
}

}  // namespace refil
