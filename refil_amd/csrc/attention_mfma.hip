// Matrix-core (v_mfma_f32_16x16x4_f32) masked entity attention core, forward + backward.
// Same math as attention.hip (reference: src/modules/layers/attention.py:48-64); that VALU kernel stays
// as the generic fallback for tile shapes not instantiated here.
//
// One WAVE owns one job = (row (b,t), net, head) at a time; the 4 waves of a workgroup take the jobs of the same row.
// Q / K / V (and dO) head slices live in wave-private LDS (pitch 16 NCT + 2 floats: the strided b32 fragment reads are
// conflict-free) or go straight from global memory into the MFMA operand registers (forward: K, Q), everything else lives
// in registers. Key trick: compute the TRANSPOSED logits  S^T[key,agent] = K Q^T.  In the MFMA D layout a lane then holds
// 4 consecutive keys of ONE agent, so
//   * the softmax over keys is lane-local + 2 shuffles (lanes ^16, ^32),
//   * the normalised weights are already the B operand (k = key) of  O^T = V^T P^T  -- no LDS round trip, no transposition,
//   * the result lane holds 4 consecutive channels of one agent -> 16-byte stores.
// The MFMA k order is a free permutation as long as A and B agree; every product here maps virtual k (step s, lane group q)
// to the index the D registers of the previous product hold. Backward needs products contracted over agents as well
// (dV, dK): it computes the logits in the other orientation S[agent,key] and gets dS^T through a wave-private LDS transpose.
// The launches are persistent and software-pipelined (below): what bounds them is the matrix-core issue of the ~200 small
// MFMAs per job and the VALU softmax beside it (PMC: 0.34 / 0.21 MFMA-busy, 3.0 / 3.9 TB/s backward / forward at cfg-T).
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"
#include "profile.h"
#include "bufops.h"
#include "../../include/refil_hip.h"

namespace refil {

// One launch serves up to ATTN_MAX_NETS attention blocks that share the rows and the masks (the four hypernets of a
// mixer): the mask words of a row are built once, the per-(net, head) jobs of a row are spread over the 4 waves, and a
// wave fetches the operands of its next job while it works on the current one.
constexpr int ATTN_MAX_NETS = 8;
struct AttnNet {
    const float* Q; const float* K; const float* V; float* O; const float* dO;
    float* dQ; float* dK; float* dV;
    int nvar;          // mask variants of this net: variants 0 .. nvar-1 of the launch (a single-variant net uses variant 0)
    int sum_agents;    // forward (nvar = 1): O[r][:] = sum over agents of the attention output (row r of a [R, w] matrix)
    int bcast_do;      // backward: dO is one row per r ([R, w]) shared by all agents of the row
};
struct AttnM {
    AttnNet net[ATTN_MAX_NETS]; int nnets;
    int ldq, ldkv, ldo; long sO;
    int R, T1, ne, na, heads, hd, nvar; int var[3];
    const uint8_t* obs_mask; long om_sB, om_sT;
    const uint8_t* ent_mask; const uint8_t* ent_mask0; const uint8_t* group_bits;
    const uint8_t* gt_mask; long gt_sB, gt_sT;
    int wave_floats;   // LDS floats per wave region
    int mask_floats;   // LDS floats of the mask byte region (the mask words follow it)
    // hypernets in 'vector' / 'scalar' mode only use the SUM of their per-agent outputs (flex_qmix.py:51-56), and every
    // layer between the attention core and that sum is linear, so the sum can be taken right here (AttnNet::sum_agents)
    float* nact;       // forward: nact[r] = number of active agents of row r (weight of the bias terms downstream) or NULL
    int zero_dead;     // forward: write zeros for inactive agents (the layer's post_mask, attention.py:66-67, applied early)
    // row skipping (refil_attn_desc: t_last / kv_dead / q_dead)
    const int* t_last; const uint8_t* kv_dead; const uint8_t* q_dead;
    // mask words built ahead of the launch (refil_attn_desc.mask_words / row_bits, attn_mask_words_launch): the kernels
    // then start with their operand loads instead of a mask phase (byte loads, two barriers, one ballot per agent and variant)
    const unsigned long long* mwords;    // [R][mw_nvar][16 NAT] (a launch uses the first nvar variants of a row)
    int mw_nvar;
    const unsigned long long* rbits;     // [R][3]: dead K/V rows, dead Q rows, inactive entities of the step (bit j)
    unsigned long long* mwords_out; unsigned long long* rbits_out;      // attn_mask_words_kernel outputs
};

struct MaskLds { const uint8_t *emt, *em0, *gb, *om, *gt, *kd, *qd; };

// bytes of the LDS mask region filled by load_masks: emt, em0, gb, kd, qd (ne each), om, gt (na * ne each)
static inline size_t mask_region_bytes(int ne, int na) { return (5 * (size_t)ne + 2 * (size_t)na * ne + 15) & ~(size_t)15; }

__device__ inline bool premask_m(int code, const MaskLds& s, int ne, int i, int j) {
    const bool in0 = s.em0[i] | s.em0[j];
    const bool same = !in0 && (s.gb[i] == s.gb[j]);
    switch (code) {
        case REFIL_MASK_OBS: return s.om[i * ne + j];
        case REFIL_MASK_OBS_WITHIN: return !same || s.om[i * ne + j];
        case REFIL_MASK_OBS_INTERACT: return same || s.om[i * ne + j];
        case REFIL_MASK_ENTITY: return s.emt[i] | s.emt[j];
        case REFIL_MASK_WITHIN: return !same;
        case REFIL_MASK_INTERACT: return same || in0;
        case REFIL_MASK_OBS_GTW: return s.gt[i * ne + j] || s.om[i * ne + j];
        case REFIL_MASK_OBS_GTI: return !s.gt[i * ne + j] || s.om[i * ne + j];
        case REFIL_MASK_GTW: return s.gt[i * ne + j] || in0;
        case REFIL_MASK_GTI: return !s.gt[i * ne + j] || in0;
        case REFIL_MASK_OBS_RGTW: return !same || s.gt[i * ne + j] || s.om[i * ne + j];
        case REFIL_MASK_OBS_RGTI: return (same && !s.gt[i * ne + j]) || s.om[i * ne + j];
        case REFIL_MASK_RGTW: return !same || s.gt[i * ne + j];
        default: return (same && !s.gt[i * ne + j]) || in0;   // REFIL_MASK_RGTI
    }
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// D[rowtile x coltile] = sum_c X[rowbase + (l&15)][c] * Y[colbase + (l&15)][c]   (both operands [rows][hd] in LDS)
__device__ inline f32x4 dot_tile(const float* X, int rowbase, const float* Y, int colbase, int hd, int pd, int l15, int q) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* xp = X + (rowbase + l15) * pd + q;
    const float* yp = Y + (colbase + l15) * pd + q;
#pragma unroll
    for (int s = 0; s < (hd >> 2); ++s) acc = MFMA16(xp[4 * s], yp[4 * s], acc);
    return acc;
}

// all-reduce over the 16 lanes of a DPP row (lanes sharing l>>4) with 4 DPP-modified VALU ops instead of 4
// ds_bpermute round trips: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (quad 0 <-> quad 1 of each
// half; every lane of a quad already holds the quad's value), row_mirror (half 0 <-> half 1)
template <int CTRL>
__device__ inline float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float group16_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    return v;
}
__device__ inline float group16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}
__device__ inline float cross4_sum(float v) {    // sum over the 4 lanes sharing l&15
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ inline float cross4_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// Mask words: mw[v * NA_PAD + agent] is a 64-bit word whose bit `key` is set when logit (agent, key) of mask
// variant v is masked; padded agents (>= na) are all ones and padded keys (>= ne) are set. Built once per
// row by the whole workgroup (one ballot per (variant, agent)), shared by the 4 heads and both orientations.
__device__ inline void build_mask_words(const AttnM& p, const MaskLds& m, unsigned long long* mw, int na_pad, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int idx = wave; idx < p.nvar * na_pad; idx += 4) {
        const int v = idx / na_pad, i = idx % na_pad;
        bool masked = true;
        if (i < p.na && lane < p.ne) masked = premask_m(p.var[v], m, p.ne, i, lane);
        const unsigned long long w = __ballot(masked);
        if (lane == 0) mw[idx] = w;
    }
}

// masked softmax of transposed logits: st[jt][reg] = S^T[key 16jt+4q+reg][agent] (already scaled), in place -> P^T
template <int NJT>
__device__ inline void softmax_T(f32x4 (&st)[NJT], unsigned long long w, int q) {
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const bool masked = (w >> (16 * jt + 4 * q + reg)) & 1ull;
            const float v = masked ? -INFINITY : st[jt][reg];
            st[jt][reg] = v;
            mx = fmaxf(mx, v);
        }
    mx = cross4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float e = st[jt][reg] == -INFINITY ? 0.f : __expf(st[jt][reg] - mx);
            st[jt][reg] = e;
            sum += e;
        }
    sum = cross4_sum(sum);
    const float inv = sum > 0.f ? __builtin_amdgcn_rcpf(sum) : 0.f;     // fully masked row -> 0 (attention.py:60 NaN -> 0)
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) st[jt][reg] *= inv;
}

// masked softmax of non-transposed logits: sn[jt][reg] = S[agent 16at+4q+reg][key 16jt + l15] -> P
template <int NJT>
__device__ inline void softmax_N(f32x4 (&sn)[NJT], const unsigned long long (&w)[4], int l15) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const bool masked = (w[reg] >> (16 * jt + l15)) & 1ull;
            const float v = masked ? -INFINITY : sn[jt][reg];
            sn[jt][reg] = v;
            mx = fmaxf(mx, v);
        }
        mx = group16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const float e = sn[jt][reg] == -INFINITY ? 0.f : __expf(sn[jt][reg] - mx);
            sn[jt][reg] = e;
            sum += e;
        }
        sum = group16_sum(sum);
        const float inv = sum > 0.f ? __builtin_amdgcn_rcpf(sum) : 0.f;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) sn[jt][reg] *= inv;
    }
}

__device__ inline void load_masks(const AttnM& p, uint8_t* base, MaskLds& m, int r, int tid, int nthreads, bool need_obs) {
    uint8_t* emt = base; uint8_t* em0 = base + p.ne; uint8_t* gb = base + 2 * p.ne; uint8_t* kd = base + 3 * p.ne;
    uint8_t* qd = base + 4 * p.ne; uint8_t* om = base + 5 * p.ne;
    uint8_t* gt = om + p.na * p.ne;
    const int b = r / p.T1, t = r % p.T1;
    for (int j = tid; j < p.ne; j += nthreads) {
        emt[j] = p.ent_mask ? p.ent_mask[(long)r * p.ne + j] : 0;
        em0[j] = p.ent_mask0 ? p.ent_mask0[(long)b * p.ne + j] : 0;
        gb[j] = p.group_bits ? p.group_bits[(long)b * p.ne + j] : 0;
        kd[j] = p.kv_dead ? p.kv_dead[(long)r * p.ne + j] : 0;
        if (j < p.na) qd[j] = p.q_dead ? p.q_dead[(long)r * p.na + j] : 0;
    }
    if (need_obs) {
        const uint8_t* src = p.obs_mask + b * p.om_sB + t * p.om_sT;
        for (int idx = tid; idx < p.na * p.ne; idx += nthreads) om[idx] = src[idx];
    }
    if (p.gt_mask) {
        const uint8_t* src = p.gt_mask + b * p.gt_sB + t * p.gt_sT;
        for (int idx = tid; idx < p.na * p.ne; idx += nthreads) gt[idx] = src[idx];
    }
    m.emt = emt; m.em0 = em0; m.gb = gb; m.om = om; m.gt = gt;
    m.kd = p.kv_dead ? kd : nullptr; m.qd = p.q_dead ? qd : nullptr;
}

// steps after an episode's last contributing step (refil_attn_desc.t_last) are skipped: their outputs stay untouched
__device__ inline bool row_skipped(const AttnM& p, int r) { return p.t_last && (r % p.T1) > p.t_last[r / p.T1]; }

__device__ inline bool uses_obs_m(const AttnM& p) {
    bool u = false;
    for (int v = 0; v < p.nvar; ++v)
        u |= mask_uses_obs(p.var[v]);
    return u;
}

static inline int tiles16(int n) { return (n + 15) / 16; }

// The mask state of a row: the mask words (LDS or precomputed in global memory) and three row words
struct RowMasks { const unsigned long long* mw; unsigned long long kdw, qdw, emtw; };

// in-kernel mask phase (no precomputed words): bytes -> LDS -> one ballot per (variant, agent); two barriers
__device__ inline RowMasks mask_phase(const AttnM& p, float* smem, int na_pad, int r, int tid) {
    MaskLds m;
    load_masks(p, reinterpret_cast<uint8_t*>(smem + 4 * p.wave_floats), m, r, tid, 256, uses_obs_m(p));
    unsigned long long* mw = reinterpret_cast<unsigned long long*>(smem + 4 * p.wave_floats + p.mask_floats);
    __syncthreads();
    build_mask_words(p, m, mw, na_pad, tid);
    const int lane = tid & 63;
    RowMasks rm;
    rm.mw = mw;
    rm.kdw = __ballot(m.kd && lane < p.ne && m.kd[lane < p.ne ? lane : 0]);
    rm.qdw = __ballot(m.qd && lane < p.na && m.qd[lane < p.na ? lane : 0]);
    rm.emtw = __ballot(lane < p.ne && m.emt[lane < p.ne ? lane : 0]);
    __syncthreads();
    return rm;
}
__device__ inline RowMasks mask_words_of(const AttnM& p, int na_pad, int r) {
    RowMasks rm;
    rm.mw = p.mwords + (long)r * p.mw_nvar * na_pad;
    rm.kdw = p.rbits[3 * (long)r]; rm.qdw = p.rbits[3 * (long)r + 1]; rm.emtw = p.rbits[3 * (long)r + 2];
    return rm;
}

// Builds the mask words of every row ONCE for all the attention launches of a step that share them (forward, backward,
// live and target nets): mwords_out [R][nvar][na_pad], rbits_out [R][3].
__global__ __launch_bounds__(256) void attn_mask_words_kernel(AttnM p, int na_pad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (row_skipped(p, r)) return;
    const RowMasks rm = mask_phase(p, smem, na_pad, r, tid);
    for (int i = tid; i < p.nvar * na_pad; i += 256) p.mwords_out[(long)r * p.nvar * na_pad + i] = rm.mw[i];
    if (tid == 0) { p.rbits_out[3 * (long)r] = rm.kdw; p.rbits_out[3 * (long)r + 1] = rm.qdw; p.rbits_out[3 * (long)r + 2] = rm.emtw; }
}

// ================================================================================================
// PERSISTENT workgroups with a software-pipelined operand fetch (attn_fwd_pipe / attn_bwd_pipe).
//
// The first generation of these kernels (one workgroup per row, rounds 1-2) ran at 0.24-0.33 of the HBM roof and 0.30 of the
// matrix cores: every wave walked load -> wait -> LDS -> compute -> store once per job, its operand fetch was a chain of
// dependent round trips (net pointers looked up with per-lane loads, one conditional load per operand tile, each behind its
// own s_waitcnt), and a workgroup lived for ONE row. Here
//   * a launch is (CUs x resident workgroups) workgroups; each walks the LIVE rows (b,t <= t_last[b]) it owns -- the
//     live-row ordinals are mapped to rows through a prefix sum of the episodes' live steps built in LDS once per
//     workgroup -- and each of its 4 waves walks its own stream of (row, net, head) jobs;
//   * the operands of job k+1 (Q, K, V, every variant's dO, the mask words) are fetched into REGISTERS while job k is
//     computed from wave-private LDS; the row words of the row after next are fetched one stage earlier still;
//   * every global access is a buffer instruction on a wave-uniform resource: rows nobody computed (dead K / V / Q rows,
//     padded tiles, the variants a net does not have, the jobs past the end) get an out-of-range offset -- the hardware
//     returns zeros / drops the store, moves no data and needs no branch, so the fetch is straight-line code whose
//     s_waitcnt bookkeeping the compiler can count exactly;
//   * nothing is loaded inside the compute phase (vmcnt is in order: a load waited for there would wait for the
//     prefetch issued before it).
// Same math as above (reference: src/modules/layers/attention.py:48-64).
// ================================================================================================
// ROWS x (4 C4) floats of a row-major matrix on their way global -> registers -> wave-private LDS (pitch pd)
template <int ROWS, int C4>
struct Tile {
    static constexpr int N = ROWS * C4 / 64;
    float4 v[N];
    // rows >= `rows`, columns >= hd and rows whose `dead` bit is set are not fetched: they enter as zeros
    __device__ inline void load(rsrc_t rs, int rows, int ld, int col0, int hd, int lane, unsigned long long dead) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = lane + 64 * i, r = idx / C4, c4 = idx % C4;
            const bool ok = r < rows && 4 * c4 < hd && !((dead >> r) & 1ull);
            v[i] = buf_ld4(rs, ok ? (r * ld + col0 + 4 * c4) * 4 : BUF_OOB);
        }
    }
    __device__ inline void store(float* dst, int pd, int lane) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = lane + 64 * i, r = idx / C4, c4 = idx % C4;
            float2* d = reinterpret_cast<float2*>(dst + r * pd + c4 * 4);
            d[0] = make_float2(v[i].x, v[i].y);
            d[1] = make_float2(v[i].z, v[i].w);
        }
    }
};

// The rows a persistent workgroup owns: live row ordinals blockIdx.x, blockIdx.x + G, ... (a row (b,t) is live when
// t <= t_last[b]); ordinal -> row through a prefix sum of the episodes' live steps, built once per workgroup in LDS
// scratch (the wave regions, not yet in use) and resolved into the workgroup's own row table rows[slot].
struct Walk { int nslots; const int* rows; };

__device__ inline Walk walk_setup(const AttnM& p, int* pref, int* rows, int tid) {
    const int G = gridDim.x, bid = blockIdx.x, nB = p.R / p.T1;
    Walk w;
    w.rows = rows;
    int nlive = p.R;
    if (p.t_last) {
        if (tid < 64) {
            int carry = 0;
            if (tid == 0) pref[0] = 0;
            for (int base = 0; base < nB; base += 64) {
                const int j = base + tid;
                int c = 0;
                if (j < nB) { c = p.t_last[j] + 1; c = c < 0 ? 0 : (c > p.T1 ? p.T1 : c); }
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(c, d, 64); if (tid >= d) c += o; }
                if (j < nB) pref[j + 1] = carry + c;
                carry += __shfl(c, 63, 64);
            }
        }
        __syncthreads();
        nlive = pref[nB];
    }
    w.nslots = bid < nlive ? (nlive - bid + G - 1) / G : 0;
    for (int s = tid; s < w.nslots; s += 256) {
        const int i = bid + s * G;
        int r = i;
        if (p.t_last) {                       // b = number of j in [1, nB] with pref[j] <= i
            int lo = 0, hi = nB;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (pref[mid + 1] <= i) lo = mid + 1; else hi = mid; }
            r = lo * p.T1 + (i - pref[lo]);
        }
        rows[s] = r;
    }
    __syncthreads();                          // (the row table is complete; the prefix scratch may be overwritten)
    return w;
}

struct RowInfo { int r; unsigned long long kdw, qdw, emtw; };      // wave-uniform

// the mask state of a row straight from the mask bytes (launches without precomputed words: the acting path, op-level calls)
struct RowBytes { unsigned long long emtw, em0w, gbw; };
__device__ inline RowBytes row_bytes(const AttnM& p, int r, int lane) {
    const int b = r / p.T1;
    const bool in = lane < p.ne;
    const bool e = in && p.ent_mask && p.ent_mask[(long)r * p.ne + lane];
    const bool e0 = in && p.ent_mask0 && p.ent_mask0[(long)b * p.ne + lane];
    const bool g = in && p.group_bits && p.group_bits[(long)b * p.ne + lane];
    RowBytes x; x.emtw = __ballot(e); x.em0w = __ballot(e0); x.gbw = __ballot(g);
    return x;
}
// bit j of the result: logit (agent i, key j) is masked under variant `code`; padded agents / keys are set (lane = key)
__device__ inline unsigned long long mask_word(const AttnM& p, int code, int r, int i, int lane, const RowBytes& rb) {
    bool masked = true;
    if (i < p.na && lane < p.ne) {
        const int b = r / p.T1, t = r % p.T1;
        const bool in0 = ((rb.em0w >> i) | (rb.em0w >> lane)) & 1ull;
        const bool same = !in0 && !(((rb.gbw >> i) ^ (rb.gbw >> lane)) & 1ull);
        const bool om = mask_uses_obs(code) && p.obs_mask[b * p.om_sB + t * p.om_sT + i * p.ne + lane];
        const bool gt = mask_uses_gt(code) && p.gt_mask[b * p.gt_sB + t * p.gt_sT + i * p.ne + lane];
        switch (code) {
            case REFIL_MASK_OBS: masked = om; break;
            case REFIL_MASK_OBS_WITHIN: masked = !same || om; break;
            case REFIL_MASK_OBS_INTERACT: masked = same || om; break;
            case REFIL_MASK_ENTITY: masked = ((rb.emtw >> i) | (rb.emtw >> lane)) & 1ull; break;
            case REFIL_MASK_WITHIN: masked = !same; break;
            case REFIL_MASK_INTERACT: masked = same || in0; break;
            case REFIL_MASK_OBS_GTW: masked = gt || om; break;
            case REFIL_MASK_OBS_GTI: masked = !gt || om; break;
            case REFIL_MASK_GTW: masked = gt || in0; break;
            case REFIL_MASK_GTI: masked = !gt || in0; break;
            case REFIL_MASK_OBS_RGTW: masked = !same || gt || om; break;
            case REFIL_MASK_OBS_RGTI: masked = (same && !gt) || om; break;
            case REFIL_MASK_RGTW: masked = !same || gt; break;
            default: masked = (same && !gt) || in0; break;       // REFIL_MASK_RGTI
        }
    }
    return __ballot(masked);
}

// The row words of a row are fetched one job ahead of the row's first operand fetch and stay in a VECTOR register
// until then (lane l holds word l % 3: a scalar destination would have to be waited for where the load is issued).
struct RowNext { int r; unsigned long long w; };
__device__ inline unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
template <bool PRE>
__device__ inline RowNext row_next(const AttnM& p, const Walk& w, int slot, int lane) {
    RowNext x;
    x.r = __builtin_amdgcn_readfirstlane(w.rows[slot]);
    x.w = PRE ? p.rbits[3 * (long)x.r + lane % 3] : 0ull;
    return x;
}
template <bool PRE>
__device__ inline RowInfo row_take(const AttnM& p, const RowNext& nx, int lane) {
    RowInfo x;
    x.r = nx.r;
    if (PRE) {
        x.kdw = readlane64(nx.w, 0); x.qdw = readlane64(nx.w, 1); x.emtw = readlane64(nx.w, 2);
    } else {
        x.kdw = __ballot(p.kv_dead && lane < p.ne && p.kv_dead[(long)x.r * p.ne + (lane < p.ne ? lane : 0)]);
        x.qdw = __ballot(p.q_dead && lane < p.na && p.q_dead[(long)x.r * p.na + (lane < p.na ? lane : 0)]);
        x.emtw = __ballot(p.ent_mask && lane < p.ne && p.ent_mask[(long)x.r * p.ne + (lane < p.ne ? lane : 0)]);
    }
    return x;
}

// LDS floats per wave / workgroup of the two kernels (launcher and kernels agree through these)
template <int NJT, int NAT, int NCT> struct PipeShape {
    static constexpr int KP = NJT * 16, AP = NAT * 16, CP = NCT * 16, C4 = CP / 4, PD = CP + 2, TP = KP + 4;
    static constexpr int FWD_WF = KP * PD;
    static constexpr int BWD_WF = (AP + 2 * KP + 3 * AP) * PD + 16 * TP + 3 * AP * 2;
};

template <int NJT, int NAT, int NCT, bool PRE>
__global__ __launch_bounds__(256) void attn_fwd_pipe(AttnM p) {
    using S = PipeShape<NJT, NAT, NCT>;
    constexpr int KP = S::KP, AP = S::AP, C4 = S::C4, pd = S::PD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    float* Vs = smem + wave * S::FWD_WF;
    const Walk W = walk_setup(p, reinterpret_cast<int*>(smem), reinterpret_cast<int*>(smem + 4 * S::FWD_WF), tid);
    if (W.nslots == 0) return;
    const int hd = p.hd;
    const int njobs = p.nnets * p.heads, jpw = (njobs + 3) >> 2;
    const int total = W.nslots * jpw;
    const float inv_scale = 1.0f / sqrtf((float)hd);
    const unsigned long long na_bits = (p.na >= 64) ? ~0ull : ((1ull << p.na) - 1ull);

    float4 kf[NJT][NCT], qf[NAT][NCT];          // operands of S^T = K Q^T in MFMA layout straight from global memory
    Tile<KP, C4> tv;
    unsigned long long nw[3][NAT];              // PRE: mask words of (variant, this lane's agent of tile at) in flight
    auto fetch = [&](const RowInfo& ri, int job, bool valid) {
        const int net = job / p.heads, head = job - net * p.heads;
        const AttnNet& n = p.net[valid ? net : 0];
        const int col0 = head * hd;
        const rsrc_t rk = mk_rsrc(n.K + (long)ri.r * p.ne * p.ldkv, valid ? (long)p.ne * p.ldkv * 4 : 0);
        const rsrc_t rq = mk_rsrc(n.Q + (long)ri.r * p.na * p.ldq, valid ? (long)p.na * p.ldq * 4 : 0);
        const rsrc_t rv = mk_rsrc(n.V + (long)ri.r * p.ne * p.ldkv, valid ? (long)p.ne * p.ldkv * 4 : 0);
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int key = 16 * jt + l15;
            const bool ok = key < p.ne && !((ri.kdw >> key) & 1ull);
#pragma unroll
            for (int u = 0; u < NCT; ++u) {
                const int c = 4 * NCT * q + 4 * u;
                kf[jt][u] = buf_ld4(rk, ok && c < hd ? (key * p.ldkv + col0 + c) * 4 : BUF_OOB);
            }
        }
#pragma unroll
        for (int at = 0; at < NAT; ++at) {
            const int ag = 16 * at + l15;
            const bool ok = ag < p.na && !((ri.qdw >> ag) & 1ull);
#pragma unroll
            for (int u = 0; u < NCT; ++u) {
                const int c = 4 * NCT * q + 4 * u;
                qf[at][u] = buf_ld4(rq, ok && c < hd ? (ag * p.ldq + col0 + c) * 4 : BUF_OOB);
            }
        }
        tv.load(rv, p.ne, p.ldkv, col0, hd, lane, ri.kdw);
        if (PRE) {
            const rsrc_t rw = mk_rsrc(p.mwords + (long)ri.r * p.mw_nvar * AP, valid ? (long)p.mw_nvar * AP * 8 : 0);
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int at = 0; at < NAT; ++at) nw[v][at] = buf_ld_u64(rw, v < p.nvar ? (v * AP + 16 * at + l15) * 8 : BUF_OOB);
        }
    };

    // fetch cursor (one job ahead of the compute cursor) and the row after its row
    int f_slot = 0, f_jj = 0;
    RowNext nx = row_next<PRE>(p, W, 0, lane);
    RowInfo frow = row_take<PRE>(p, nx, lane);
    nx = row_next<PRE>(p, W, min(1, W.nslots - 1), lane);
    fetch(frow, wave, wave < njobs);
    {   // as many (dropped) stores behind the first fetch as every job issues behind the fetch of its successor: the operand
        // wait at the top of the loop is then the same count on both ways into the loop
        const rsrc_t none = mk_rsrc(p.net[0].O, 0);
#pragma unroll
        for (int i = 0; i < NAT * 3 * NCT + NCT + 1; ++i) buf_st4(none, BUF_OOB - 16 * i, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    for (int k = 0; k < total; ++k) {
        const RowInfo crow = frow;
        const int job = wave + 4 * f_jj;
        const bool cvalid = job < njobs;
        const int net = job / p.heads, head = job - net * p.heads;
        const AttnNet& n = p.net[cvalid ? net : 0];
        const int r = crow.r;
        // ---- operands of job k: S^T tiles from the operand registers, V and the words to their compute-phase homes
        f32x4 stt[NAT][NJT];
#pragma unroll
        for (int at = 0; at < NAT; ++at)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NCT; ++u) {
                    const float4 kv = kf[jt][u], qv = qf[at][u];
                    acc = MFMA16(kv.x, qv.x, acc);
                    acc = MFMA16(kv.y, qv.y, acc);
                    acc = MFMA16(kv.z, qv.z, acc);
                    acc = MFMA16(kv.w, qv.w, acc);
                }
                stt[at][jt] = acc;
            }
        tv.store(Vs, pd, lane);
        unsigned long long cw[3][NAT];
        if (PRE) {
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int at = 0; at < NAT; ++at) cw[v][at] = nw[v][at];
        } else {
            const RowBytes rb = row_bytes(p, r, lane);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
#pragma unroll
                for (int at = 0; at < NAT; ++at) cw[v][at] = ~0ull;
                if (v < p.nvar)
                    for (int i = 0; i < p.na; ++i) {
                        const unsigned long long w = mask_word(p, p.var[v], r, i, lane, rb);
#pragma unroll
                        for (int at = 0; at < NAT; ++at)
                            if (i == 16 * at + l15) cw[v][at] = w;
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- advance the fetch cursor; the operands of job k+1 are in flight during the compute phase below
        const bool adv = f_jj + 1 == jpw;
        f_jj = adv ? 0 : f_jj + 1;
        f_slot += adv;
        if (adv) frow = row_take<PRE>(p, nx, lane);
        nx = row_next<PRE>(p, W, min(f_slot + 1, W.nslots - 1), lane);    // (every job: no branch around a load in flight)
        fetch(frow, wave + 4 * f_jj, k + 1 < total && wave + 4 * f_jj < njobs);
        // (nact[r] by lane 0 of the row's first job; a buffer store like the others: no divergent branch, a fixed store count)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)__popcll(~crow.emtw & na_bits)),
                                              mk_rsrc(p.nact + r, p.nact && job == 0 ? 4 : 0), lane == 0 ? 0 : BUF_OOB, 0, 0);
        // ---- compute. Every job issues the SAME number of stores (the ones it does not have are out of range): the
        // s_waitcnt for the operands of the next job can then be counted past them instead of waiting for them
        f32x4 osum[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) osum[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int at = 0; at < NAT; ++at) {
            const int agent = 16 * at + l15;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                f32x4 o[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool on = cvalid && v < n.nvar;
                if (on) {
                    f32x4 pt[NJT];
#pragma unroll
                    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) pt[jt][reg] = stt[at][jt][reg] * inv_scale;   // attention.py:54
                    softmax_T<NJT>(pt, cw[v][at], q);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        // O^T[c][agent] = sum_key V[key][c] P^T[key][agent]; virtual k (jt,reg | q) <-> key 16jt+4q+reg
#pragma unroll
                        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg)
                                o[ct] = MFMA16(Vs[(16 * jt + 4 * q + reg) * pd + 16 * ct + l15], pt[jt][reg], o[ct]);
                        if (n.sum_agents) {      // padded / inactive agents have P = 0, hence o = 0: plain sum over the 16 lanes
#pragma unroll
                            for (int e = 0; e < 4; ++e) osum[ct][e] += group16_sum(o[ct][e]);
                        } else if (p.zero_dead && ((crow.emtw >> agent) & 1ull)) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                const rsrc_t ro = mk_rsrc(n.O + v * p.sO + (long)r * p.na * p.ldo, on && !n.sum_agents ? (long)p.na * p.ldo * 4 : 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int c = 16 * ct + 4 * q;
                    buf_st4(ro, agent < p.na && c < hd ? (agent * p.ldo + head * hd + c) * 4 : BUF_OOB, o[ct]);
                }
            }
        }
        {
            const rsrc_t ro = mk_rsrc(n.O + (long)r * p.ldo, cvalid && n.sum_agents ? (long)p.ldo * 4 : 0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int c = 16 * ct + 4 * q;
                buf_st4(ro, l15 == 0 && c < hd ? (head * hd + c) * 4 : BUF_OOB, osum[ct]);
            }
        }
        __builtin_amdgcn_wave_barrier();                   // (the next job's V tile overwrites the wave's LDS region)
    }
}

#ifdef REFIL_ATTN_TIMING
// debug build only (tools/probes/attn_sched.py): when every workgroup of the LAST attn_bwd_pipe launch started and ended (100 MHz wall clock)
__device__ unsigned long long g_attn_dbg[4096 * 2];
#endif

template <int NJT, int NAT, int NCT, bool PRE>
__global__ __launch_bounds__(256, (NJT <= 2 && NAT == 1) ? 2 : 1) void attn_bwd_pipe(AttnM p) {
#ifdef REFIL_ATTN_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 4096) { g_attn_dbg[2 * blockIdx.x] = wall_clock64(); g_attn_dbg[2 * blockIdx.x + 1] = 0; }
#endif
    using S = PipeShape<NJT, NAT, NCT>;
    constexpr int KP = S::KP, AP = S::AP, C4 = S::C4, pd = S::PD, TP = S::TP;
    constexpr int NW = (3 * AP + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    float* Qs = smem + wave * S::BWD_WF;
    float* Ks = Qs + AP * pd;
    float* Vs = Ks + KP * pd;
    float* Ds = Vs + KP * pd;                 // dO of every variant: [3][AP] rows
    float* Ts = Ds + 3 * AP * pd;             // dS transposition tile (16 agents x keys)
    unsigned long long* MW = reinterpret_cast<unsigned long long*>(Ts + 16 * TP);      // mask words [3][AP]
    const Walk W = walk_setup(p, reinterpret_cast<int*>(smem), reinterpret_cast<int*>(smem + 4 * S::BWD_WF), tid);
    if (W.nslots == 0) return;
    const int hd = p.hd;
    const int njobs = p.nnets * p.heads, jpw = (njobs + 3) >> 2;
    const int total = W.nslots * jpw;
    const float inv_scale = 1.0f / sqrtf((float)hd);

    Tile<AP, C4> tq, td[3];
    Tile<KP, C4> tk, tv;
    unsigned long long tw[NW];
    auto fetch = [&](const RowInfo& ri, int job, bool valid) {
        const int net = job / p.heads, head = job - net * p.heads;
        const AttnNet& n = p.net[valid ? net : 0];
        const int col0 = head * hd;
        const long qrow = (long)ri.r * p.na, krow = (long)ri.r * p.ne;
        tq.load(mk_rsrc(n.Q + qrow * p.ldq, valid ? (long)p.na * p.ldq * 4 : 0), p.na, p.ldq, col0, hd, lane, ri.qdw);
        tk.load(mk_rsrc(n.K + krow * p.ldkv, valid ? (long)p.ne * p.ldkv * 4 : 0), p.ne, p.ldkv, col0, hd, lane, ri.kdw);
        tv.load(mk_rsrc(n.V + krow * p.ldkv, valid ? (long)p.ne * p.ldkv * 4 : 0), p.ne, p.ldkv, col0, hd, lane, ri.kdw);
        // (bcast_do: one dO row per (b,t), shared by its agents -- leading dimension 0 re-reads the same row)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const bool on = valid && v < n.nvar, bc = n.bcast_do != 0;
            td[v].load(mk_rsrc(n.dO + v * p.sO + (bc ? (long)ri.r : qrow) * p.ldo, on ? (long)(bc ? 1 : p.na) * p.ldo * 4 : 0), p.na, bc ? 0 : p.ldo,
                       col0, hd, lane, bc ? 0ull : ri.qdw);
        }
        if (PRE) {
            const rsrc_t rw = mk_rsrc(p.mwords + (long)ri.r * p.mw_nvar * AP, valid ? (long)p.mw_nvar * AP * 8 : 0);
#pragma unroll
            for (int i = 0; i < NW; ++i) tw[i] = buf_ld_u64(rw, lane + 64 * i < p.nvar * AP ? (lane + 64 * i) * 8 : BUF_OOB);
        }
    };

    int f_slot = 0, f_jj = 0;
    RowNext nx = row_next<PRE>(p, W, 0, lane);
    RowInfo frow = row_take<PRE>(p, nx, lane);
    nx = row_next<PRE>(p, W, min(1, W.nslots - 1), lane);
    fetch(frow, wave, wave < njobs);
    {   // as many (dropped) stores behind the first fetch as every job issues behind the fetch of its successor: the operand
        // wait at the top of the loop is then the same count on both ways into the loop
        const rsrc_t none = mk_rsrc(p.net[0].dQ, 0);
#pragma unroll
        for (int i = 0; i < NAT * NCT + 2 * NCT * NJT; ++i) buf_st4(none, BUF_OOB - 16 * i, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    for (int k = 0; k < total; ++k) {
        const RowInfo crow = frow;
        const int job = wave + 4 * f_jj;
        const bool cvalid = job < njobs;
        const int net = job / p.heads, head = job - net * p.heads;
        const AttnNet& n = p.net[cvalid ? net : 0];
        const int r = crow.r, nvar = cvalid ? n.nvar : 0;
        // ---- operands of job k: registers -> wave-private LDS
        tq.store(Qs, pd, lane); tk.store(Ks, pd, lane); tv.store(Vs, pd, lane);
#pragma unroll
        for (int v = 0; v < 3; ++v)
            if (v < nvar) td[v].store(Ds + v * AP * pd, pd, lane);
        if (PRE) {
#pragma unroll
            for (int i = 0; i < NW; ++i)
                if (lane + 64 * i < 3 * AP) MW[lane + 64 * i] = tw[i];
        } else if (cvalid) {
            const RowBytes rb = row_bytes(p, r, lane);
            for (int idx = 0; idx < nvar * AP; ++idx) {
                const unsigned long long w = mask_word(p, p.var[idx / AP], r, idx % AP, lane, rb);
                if (lane == 0) MW[idx] = w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- advance the fetch cursor; the operands of job k+1 are in flight during the compute phase below
        const bool adv = f_jj + 1 == jpw;
        f_jj = adv ? 0 : f_jj + 1;
        f_slot += adv;
        if (adv) frow = row_take<PRE>(p, nx, lane);
        nx = row_next<PRE>(p, W, min(f_slot + 1, W.nslots - 1), lane);    // (every job: no branch around a load in flight)
        fetch(frow, wave + 4 * f_jj, k + 1 < total && wave + 4 * f_jj < njobs);
        // ---- compute (a job past the end has no variants and its stores are dropped: every job issues the same number of
        // stores, so the operand wait of the next job can be counted past them)
        f32x4 dKt[NCT][NJT], dVt[NCT][NJT];     // [c 16ct+4q+reg][key 16jt+l15]
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                dKt[ct][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
                dVt[ct][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        const rsrc_t rdq = mk_rsrc(n.dQ + (long)r * p.na * p.ldq, cvalid ? (long)p.na * p.ldq * 4 : 0);
#pragma unroll
        for (int at = 0; at < NAT; ++at) {
            const int agentT = 16 * at + l15;        // agent of this lane in the transposed orientation (dQ^T columns)
            const int agentN0 = 16 * at + 4 * q;     // first agent of this lane in the normal orientation
            f32x4 sn0[NJT];
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) sn0[jt] = dot_tile(Qs, 16 * at, Ks, 16 * jt, 16 * NCT, pd, l15, q);   // S[agent][key]
            f32x4 dQt[NCT];                           // [c][agent 16at+l15]
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) dQt[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int v = 0; v < nvar; ++v) {
                const float* Dv = Ds + (v * AP + 16 * at) * pd;
                f32x4 pn[NJT];
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) pn[jt][reg] = sn0[jt][reg] * inv_scale;
                const unsigned long long* mv = MW + v * AP + agentN0;
                const unsigned long long wn[4] = {mv[0], mv[1], mv[2], mv[3]};
                softmax_N<NJT>(pn, wn, l15);
                // dP[agent][key] = dO V^T ; dS = P (dP - sum_key P dP) / scale
                f32x4 dsn[NJT];
                float rdN[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) {
                    dsn[jt] = dot_tile(Dv, 0, Vs, 16 * jt, 16 * NCT, pd, l15, q);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) rdN[reg] += pn[jt][reg] * dsn[jt][reg];
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) rdN[reg] = group16_sum(rdN[reg]);
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        dsn[jt][reg] = pn[jt][reg] * (dsn[jt][reg] - rdN[reg]) * inv_scale;
                        Ts[(4 * q + reg) * TP + 16 * jt + l15] = dsn[jt][reg];
                    }
                // the transposed copy dS^T[key 16jt+4q+reg][agent l15] (B operand of the contraction over keys)
                // comes back through wave-private LDS (same-wave LDS operations execute in order)
                __builtin_amdgcn_wave_barrier();
                f32x4 dst[NJT];
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt) dst[jt] = *reinterpret_cast<const f32x4*>(Ts + l15 * TP + 16 * jt + 4 * q);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            // contraction over agents: virtual k (reg | q) <-> agent 16at+4q+reg (rows of Dv / Qs)
                            dVt[ct][jt] = MFMA16(Dv[(4 * q + reg) * pd + 16 * ct + l15], pn[jt][reg], dVt[ct][jt]);
                            dKt[ct][jt] = MFMA16(Qs[(16 * at + 4 * q + reg) * pd + 16 * ct + l15], dsn[jt][reg], dKt[ct][jt]);
                            // contraction over keys: virtual k (jt,reg | q) <-> key 16jt+4q+reg
                            dQt[ct] = MFMA16(Ks[(16 * jt + 4 * q + reg) * pd + 16 * ct + l15], dst[jt][reg], dQt[ct]);
                        }
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int c = 16 * ct + 4 * q;
                const bool ok = agentT < p.na && c < hd && !((crow.qdw >> agentT) & 1ull);       // (rows of skipped agents: nobody reads them)
                buf_st4(rdq, ok ? (agentT * p.ldq + head * hd + c) * 4 : BUF_OOB, dQt[ct]);
            }
        }
        const rsrc_t rdk = mk_rsrc(n.dK + (long)r * p.ne * p.ldkv, cvalid ? (long)p.ne * p.ldkv * 4 : 0);
        const rsrc_t rdv = mk_rsrc(n.dV + (long)r * p.ne * p.ldkv, cvalid ? (long)p.ne * p.ldkv * 4 : 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                const int key = 16 * jt + l15, c = 16 * ct + 4 * q;
                const bool ok = key < p.ne && c < hd && !((crow.kdw >> key) & 1ull);      // (dead K / V rows: their gradients are exact zeros nobody reads)
                const int off = ok ? (key * p.ldkv + head * hd + c) * 4 : BUF_OOB;
                buf_st4(rdk, off, dKt[ct][jt]);
                buf_st4(rdv, off, dVt[ct][jt]);
            }
        __builtin_amdgcn_wave_barrier();                   // (the next job's operands overwrite the wave's LDS tiles)
    }
#ifdef REFIL_ATTN_TIMING
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_attn_dbg[2 * blockIdx.x + 1] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------
// EntityPoolingLayer core (attention.py:114-123): masked mean / max over the entities, same mask variants.
// One workgroup per row (b,t); the row's in_trans outputs E[ne][w] sit in LDS; the mask words are the
// attention kernels' (one ballot per (variant, agent)).
// ------------------------------------------------------------------------------------------------
struct PoolK {
    AttnM a;
    int mode;      // 1 mean, 2 max
    int w;         // channels = heads * hd
    int na_pad;
};

__device__ inline void pool_stage(const PoolK& k, float* E, int r, int tid) {
    const AttnM& p = k.a;
    const int w4 = k.w >> 2;
    for (int idx = tid; idx < p.ne * w4; idx += 256) {
        const int j = idx / w4, c4 = idx - j * w4;
        *reinterpret_cast<float4*>(E + j * k.w + 4 * c4) = *reinterpret_cast<const float4*>(p.net[0].K + ((long)r * p.ne + j) * p.ldkv + 4 * c4);
    }
}

__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolK k) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttnM& p = k.a;
    const int tid = threadIdx.x, r = blockIdx.x;
    float* E = smem;
    uint8_t* mb = reinterpret_cast<uint8_t*>(smem + p.ne * k.w);
    unsigned long long* mw = reinterpret_cast<unsigned long long*>(mb + 4 * p.mask_floats);
    MaskLds m;
    load_masks(p, mb, m, r, tid, 256, uses_obs_m(p));
    pool_stage(k, E, r, tid);
    __syncthreads();
    build_mask_words(p, m, mw, k.na_pad, tid);
    __syncthreads();
    const float inv_ne = 1.0f / (float)p.ne;
    for (int idx = tid; idx < p.nvar * p.na * k.w; idx += 256) {
        const int c = idx % k.w, vi = idx / k.w, i = vi % p.na, v = vi / p.na;
        const unsigned long long word = mw[v * k.na_pad + i];
        float acc = k.mode == 2 ? -INFINITY : 0.f;
        for (int j = 0; j < p.ne; ++j) {
            const float x = ((word >> j) & 1ull) ? 0.f : E[j * k.w + c];
            acc = k.mode == 2 ? fmaxf(acc, x) : acc + x;
        }
        p.net[0].O[v * p.sO + ((long)r * p.na + i) * p.ldo + c] = k.mode == 2 ? acc : acc * inv_ne;
    }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolK k) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttnM& p = k.a;
    const int tid = threadIdx.x, r = blockIdx.x;
    float* E = smem;                                   // [ne][w] forward values (max mode)
    float* G = smem + p.ne * k.w;                      // [ne][w] gradient accumulator
    uint8_t* mb = reinterpret_cast<uint8_t*>(G + p.ne * k.w);
    unsigned long long* mw = reinterpret_cast<unsigned long long*>(mb + 4 * p.mask_floats);
    MaskLds m;
    load_masks(p, mb, m, r, tid, 256, uses_obs_m(p));
    if (k.mode == 2) pool_stage(k, E, r, tid);
    for (int idx = tid; idx < p.ne * k.w; idx += 256) G[idx] = 0.f;
    __syncthreads();
    build_mask_words(p, m, mw, k.na_pad, tid);
    __syncthreads();
    const float inv_ne = 1.0f / (float)p.ne;
    // a thread owns channel(s) c: every (variant, agent) of that channel is routed by the same thread, so the LDS
    // accumulator column needs no atomics
    for (int c = tid; c < k.w; c += 256) {
        for (int v = 0; v < p.nvar; ++v)
            for (int i = 0; i < p.na; ++i) {
                const unsigned long long word = mw[v * k.na_pad + i];
                const float g = p.net[0].dO[v * p.sO + ((long)r * p.na + i) * p.ldo + c];
                if (k.mode == 2) {
                    float best = -INFINITY; int jb = 0;
                    for (int j = 0; j < p.ne; ++j) {
                        const float x = ((word >> j) & 1ull) ? 0.f : E[j * k.w + c];
                        if (x > best) { best = x; jb = j; }       // strict: the FIRST maximum wins (torch.max)
                    }
                    if (!((word >> jb) & 1ull)) G[jb * k.w + c] += g;      // masked_fill blocks the gradient
                } else {
                    const float gs = g * inv_ne;
                    for (int j = 0; j < p.ne; ++j)
                        if (!((word >> j) & 1ull)) G[j * k.w + c] += gs;
                }
            }
    }
    __syncthreads();
    const int w4 = k.w >> 2;
    for (int idx = tid; idx < p.ne * w4; idx += 256) {
        const int j = idx / w4, c4 = idx - j * w4;
        *reinterpret_cast<float4*>(p.net[0].dK + ((long)r * p.ne + j) * p.ldkv + 4 * c4) = *reinterpret_cast<const float4*>(G + j * k.w + 4 * c4);
    }
}

int pool_launch(const refil_attn_desc& d, int mode, bool bwd, hipStream_t st) {
    REFIL_CHECK(mode == 1 || mode == 2, "refil_pool: mode must be 1 (mean) or 2 (max)");
    REFIL_CHECK(d.K && (bwd ? (d.dO && d.dK) : (d.O != nullptr)), "refil_pool: null pointer");
    REFIL_CHECK(d.ne >= 1 && d.ne <= 64 && d.na >= 1 && d.na <= d.ne, "refil_pool: n_entities must be <= 64");
    REFIL_CHECK(d.nvar >= 1 && d.nvar <= 3 && d.R > 0 && d.T1 > 0, "refil_pool: bad nvar / R / T1");
    const int w = d.heads * d.hd;
    REFIL_CHECK(w % 4 == 0 && d.ldkv % 4 == 0, "refil_pool: width and ldkv must be multiples of 4");
    for (int v = 0; v < d.nvar; ++v) {
        REFIL_CHECK(d.var[v] >= 0 && d.var[v] < REFIL_MASK_COUNT, "refil_pool: bad mask code %d", d.var[v]);
        REFIL_CHECK(!mask_uses_obs(d.var[v]) || d.obs_mask, "refil_pool: obs_mask required by a mask variant");
        REFIL_CHECK(!mask_uses_gt(d.var[v]) || d.gt_mask, "refil_pool: gt_mask required by a mask variant");
        REFIL_CHECK(!mask_uses_groups(d.var[v]) || (d.ent_mask0 && d.group_bits), "refil_pool: ent_mask0/group_bits required by a mask variant");
        REFIL_CHECK(!mask_uses_inactive0(d.var[v]) || d.ent_mask0, "refil_pool: ent_mask0 required by a mask variant");
        REFIL_CHECK(d.var[v] != REFIL_MASK_ENTITY || d.ent_mask, "refil_pool: ent_mask required by a mask variant");
    }
    PoolK k;
    AttnM& a = k.a;
    memset(&a, 0, sizeof(a));
    a.net[0].K = d.K; a.net[0].O = d.O; a.net[0].dO = d.dO; a.net[0].dK = d.dK; a.nnets = 1; a.ldkv = d.ldkv; a.ldo = d.ldo; a.sO = d.sO;
    a.R = d.R; a.T1 = d.T1; a.ne = d.ne; a.na = d.na; a.heads = d.heads; a.hd = d.hd; a.nvar = d.nvar;
    for (int v = 0; v < 3; ++v) a.var[v] = d.var[v];
    a.obs_mask = d.obs_mask; a.om_sB = d.om_sB; a.om_sT = d.om_sT;
    a.ent_mask = d.ent_mask; a.ent_mask0 = d.ent_mask0; a.group_bits = d.group_bits;
    a.gt_mask = d.gt_mask; a.gt_sB = d.gt_sB; a.gt_sT = d.gt_sT;
    a.mask_floats = (int)(mask_region_bytes(d.ne, d.na) / 4);
    k.mode = mode; k.w = w; k.na_pad = d.na;
    const size_t smem = ((size_t)(bwd ? 2 : 1) * d.ne * w + a.mask_floats) * sizeof(float) + (size_t)3 * d.na * 8 + 16;
    REFIL_CHECK(smem <= 160 * 1024, "refil_pool: row tile does not fit LDS");
    ProfScope prof(bwd ? "pool_bwd_kernel" : "pool_fwd_kernel", 0.0, 4.0 * d.R * w * (d.ne + (double)d.nvar * d.na), st);
    if (bwd) {
        static bool raised = false;
        if (smem > 64 * 1024 && !raised) { REFIL_HIP(hipFuncSetAttribute((const void*)pool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); raised = true; }
        hipLaunchKernelGGL(pool_bwd_kernel, dim3(d.R), dim3(256), smem, st, k);
    } else {
        static bool raised = false;
        if (smem > 64 * 1024 && !raised) { REFIL_HIP(hipFuncSetAttribute((const void*)pool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); raised = true; }
        hipLaunchKernelGGL(pool_fwd_kernel, dim3(d.R), dim3(256), smem, st, k);
    }
    REFIL_LAUNCH_CHECK();
    return 0;
}


static int device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// persistent launch: as many workgroups as the device keeps resident (occupancy query cached per instantiation and LDS size)
template <int NJT, int NAT, int NCT, bool PRE>
static int launch_pipe_x(const AttnM& k, bool bwd, hipStream_t st) {
    using S = PipeShape<NJT, NAT, NCT>;
    const size_t waves = (size_t)4 * (bwd ? S::BWD_WF : S::FWD_WF) * 4;
    if (((size_t)(k.R / k.T1) + 1) * 4 > waves) return -1;         // (the live-step prefix sums use the wave regions as scratch)
    void (*kern)(AttnM) = bwd ? attn_bwd_pipe<NJT, NAT, NCT, PRE> : attn_fwd_pipe<NJT, NAT, NCT, PRE>;
    static bool raised[2] = {false, false};
    static int occ[2] = {0, 0};
    if (!raised[bwd]) {
        REFIL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised[bwd] = true;
    }
    if (!occ[bwd]) {                            // resident workgroups per CU (with room for a row table of a few hundred slots)
        int n = 0;
        REFIL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 256, waves + 1024));
        occ[bwd] = n > 0 ? n : 1;
    }
    static const int occ_cap = [] { const char* e = getenv("REFIL_ATTN_OCC"); return e ? atoi(e) : 0; }();
    const int per_cu = occ_cap > 0 && occ_cap < occ[bwd] ? occ_cap : occ[bwd];
    const long resident = (long)device_cus() * per_cu;
    const int grid = (int)(resident < k.R ? resident : k.R);
    const size_t smem = waves + ((size_t)(k.R + grid - 1) / grid + 1) * 4;      // + the workgroup's row table
    if (smem > 160 * 1024) return -1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, k);
    REFIL_LAUNCH_CHECK();
    return 0;
}
template <int NJT, int NAT, int NCT>
static int launch_pipe(const AttnM& k, bool bwd, hipStream_t st) {
    return k.mwords ? launch_pipe_x<NJT, NAT, NCT, true>(k, bwd, st) : launch_pipe_x<NJT, NAT, NCT, false>(k, bwd, st);
}

bool attn_mfma_supported(int ne, int na, int hd) {
    const int j = tiles16(ne), a = tiles16(na), c = tiles16(hd);
    return hd % 4 == 0 && na <= ne && ((j <= 2 && a == 1 && c <= 2) || (j >= 2 && j <= 4 && a <= 2 && c == 2));
}

int attn_mfma_launch(const refil_attn_desc& d, bool bwd, hipStream_t st) { return attn_mfma_launch_ex(d, bwd, st, 0, nullptr, 0, 0); }

int attn_mfma_launch_ex(const refil_attn_desc& d, bool bwd, hipStream_t st, int sum_agents, float* nact, int bcast_do, int zero_dead) {
    AttnNetOpts o{sum_agents, bcast_do};
    return attn_mfma_launch_multi(&d, &o, 1, bwd, st, nact, zero_dead);
}

// mask words of all rows for the variants of `d` (attn_mask_words_kernel): mwords [R][nvar][16 ceil(na/16)], rbits [R][3]
int attn_mask_words_launch(const refil_attn_desc& d, unsigned long long* mwords, unsigned long long* rbits, hipStream_t st) {
    REFIL_CHECK(mwords && rbits && d.nvar >= 1 && d.nvar <= 3, "refil_attn_mask_words: bad arguments");
    AttnM k;
    memset(&k, 0, sizeof(k));
    k.R = d.R; k.T1 = d.T1; k.ne = d.ne; k.na = d.na; k.heads = d.heads; k.hd = d.hd; k.nvar = d.nvar;
    for (int v = 0; v < 3; ++v) k.var[v] = d.var[v];
    k.obs_mask = d.obs_mask; k.om_sB = d.om_sB; k.om_sT = d.om_sT;
    k.ent_mask = d.ent_mask; k.ent_mask0 = d.ent_mask0; k.group_bits = d.group_bits;
    k.gt_mask = d.gt_mask; k.gt_sB = d.gt_sB; k.gt_sT = d.gt_sT;
    k.t_last = d.t_last; k.kv_dead = d.kv_dead; k.q_dead = d.q_dead;
    k.mwords_out = mwords; k.rbits_out = rbits;
    k.wave_floats = 0;
    k.mask_floats = (int)(mask_region_bytes(d.ne, d.na) / 4);
    const int na_pad = tiles16(d.na) * 16;
    const size_t smem = (size_t)k.mask_floats * 4 + (size_t)3 * na_pad * 8;
    ProfScope prof("attn_mask_words_kernel", 0.0, 0.0, st);
    hipLaunchKernelGGL(attn_mask_words_kernel, dim3(d.R), dim3(256), smem, st, k, na_pad);
    REFIL_LAUNCH_CHECK();
    return 0;
}

// n attention blocks in one launch (same rows, masks, widths and leading dimensions; descs[0] carries the mask variants,
// the others are single-variant nets under its variant 0). Returns -1 when the tile shape is not instantiated (caller
// falls back to the VALU kernel).
int attn_mfma_launch_multi(const refil_attn_desc* descs, const AttnNetOpts* opts, int n, bool bwd, hipStream_t st, float* nact, int zero_dead) {
    REFIL_CHECK(descs && opts && n >= 1 && n <= ATTN_MAX_NETS, "refil_attn: 1..%d nets per launch", ATTN_MAX_NETS);
    const refil_attn_desc& d = descs[0];
    const int njt = tiles16(d.ne), nat = tiles16(d.na), nct = tiles16(d.hd);
    AttnM k;
    memset(&k, 0, sizeof(k));
    k.nnets = n;
    for (int i = 0; i < n; ++i) {
        const refil_attn_desc& e = descs[i];
        REFIL_CHECK(e.R == d.R && e.T1 == d.T1 && e.ne == d.ne && e.na == d.na && e.heads == d.heads && e.hd == d.hd && e.ldq == d.ldq &&
                    e.ldkv == d.ldkv && e.ldo == d.ldo && e.sO == d.sO && e.obs_mask == d.obs_mask && e.ent_mask == d.ent_mask &&
                    e.ent_mask0 == d.ent_mask0 && e.group_bits == d.group_bits && e.gt_mask == d.gt_mask && e.t_last == d.t_last &&
                    e.kv_dead == d.kv_dead && e.q_dead == d.q_dead && e.mask_words == d.mask_words && e.row_bits == d.row_bits,
                    "refil_attn: nets of one launch must share rows, masks and strides");
        REFIL_CHECK(i == 0 || (e.nvar == 1 && e.var[0] == d.var[0]), "refil_attn: nets after the first are single-variant under variant 0");
        REFIL_CHECK(!opts[i].sum_agents || (!bwd && e.nvar == 1), "refil_attn: the agent-sum output is a forward, single-variant option");
        AttnNet& t = k.net[i];
        t.Q = e.Q; t.K = e.K; t.V = e.V; t.O = e.O; t.dO = e.dO; t.dQ = e.dQ; t.dK = e.dK; t.dV = e.dV;
        t.nvar = e.nvar; t.sum_agents = opts[i].sum_agents; t.bcast_do = opts[i].bcast_do;
    }
    k.ldq = d.ldq; k.ldkv = d.ldkv; k.ldo = d.ldo; k.sO = d.sO;
    k.R = d.R; k.T1 = d.T1; k.ne = d.ne; k.na = d.na; k.heads = d.heads; k.hd = d.hd; k.nvar = d.nvar;
    for (int v = 0; v < 3; ++v) k.var[v] = d.var[v];
    k.obs_mask = d.obs_mask; k.om_sB = d.om_sB; k.om_sT = d.om_sT;
    k.ent_mask = d.ent_mask; k.ent_mask0 = d.ent_mask0; k.group_bits = d.group_bits;
    k.gt_mask = d.gt_mask; k.gt_sB = d.gt_sB; k.gt_sT = d.gt_sT;
    k.nact = nact; k.zero_dead = zero_dead;
    k.t_last = d.t_last; k.kv_dead = d.kv_dead; k.q_dead = d.q_dead;
    k.mwords = reinterpret_cast<const unsigned long long*>(d.mask_words); k.rbits = reinterpret_cast<const unsigned long long*>(d.row_bits);
    k.mw_nvar = d.mask_words_nvar > 0 ? d.mask_words_nvar : d.nvar;
    REFIL_CHECK(!d.mask_words || (d.row_bits && k.mw_nvar >= d.nvar), "refil_attn: mask_words needs row_bits and >= nvar variants per row");
    REFIL_CHECK(!zero_dead || d.ent_mask || d.mask_words, "refil_attn: zeroing inactive agents needs ent_mask");
    REFIL_CHECK(d.T1 > 0 && d.R % d.T1 == 0, "refil_attn: R must be a multiple of T1");
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const double unit = (double)d.R * d.heads * d.na * d.ne * d.hd;
        flops += unit * (bwd ? 2.0 + 8.0 * descs[i].nvar : 2.0 + 2.0 * descs[i].nvar);
        bytes += 4.0 * d.R * d.heads * d.hd * (bwd ? d.na * (2.0 + descs[i].nvar) + 4.0 * d.ne : d.na * (1.0 + descs[i].nvar) + 2.0 * d.ne);
    }
    if (!attn_mfma_supported(d.ne, d.na, d.hd)) return -1;
    ProfScope prof(bwd ? "attn_bwd_mfma" : "attn_fwd_mfma", flops, bytes, st);
#define CASE(J, A, C) if (njt == J && nat == A && nct == C) return launch_pipe<J, A, C>(k, bwd, st)
    CASE(1, 1, 1); CASE(1, 1, 2); CASE(2, 1, 1); CASE(2, 1, 2); CASE(2, 2, 2); CASE(3, 1, 2); CASE(3, 2, 2); CASE(4, 1, 2); CASE(4, 2, 2);
#undef CASE
    return -1;
}

}  // namespace refil

#ifdef REFIL_ATTN_TIMING
extern "C" int refil_debug_attn_timing(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(refil::g_attn_dbg), (size_t)n * 2 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
