// Masked multi-head entity attention core for gfx950 (forward + hand-written backward).
//
// Restates EntityAttentionLayer.forward between in_trans and out_trans (reference:
// src/modules/layers/attention.py:48-64): queries = first na entities, keys/values = all ne,
// logits / sqrt(hd), pre-mask -> -inf, softmax over entities, fully-masked rows -> 0 (the
// reference's NaN->0 fill), weights @ V.
//
// MI355X design:
//  * one workgroup (2 waves) per (row = (b,t), head): the whole problem of a head -- Q [na,hd],
//    K,V [ne,hd], logits [na,ne] -- lives in LDS (<= ~30 KB), so K/V are read from HBM exactly once
//    for ALL mask variants; R*heads workgroups (>= 10k at the bench shapes) fill the 256 CUs.
//  * the 3 REFIL mask variants (real / within-group / between-group) are NOT materialised as
//    [3B,T,ne,ne] tensors like the reference does (entity_rnn_agent.py:97-121); each lane derives
//    its mask bit from 2 bits per entity (random group, inactive) + the obs_mask byte.
//  * softmax rows are ne <= 64 wide: one sub-wave group of NEP = pow2(ne) lanes per row, max/sum by
//    __shfl_xor inside the group.
//  * LDS rows are padded to hd+4 floats: ds_read_b128 of 16 different rows is then conflict-free
//    (36*j mod 64 distinct for j mod 16).
//  * backward recomputes the softmax from Q,K (no [G,R,heads,na,ne] weights tensor is stored) and
//    accumulates dQ/dK/dV over the variants in LDS before one coalesced store.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "profile.h"
#include "../../include/refil_hip.h"

namespace refil {

constexpr int ANT = 128;  // threads per attention workgroup

struct AttnK {
    const float* Q; const float* K; const float* V; float* O; const float* dO;
    float* dQ; float* dK; float* dV;
    int ldq, ldkv, ldo; long sO;
    int R, T1, ne, na, heads, hd, nvar; int var[3];
    const uint8_t* obs_mask; long om_sB, om_sT;
    const uint8_t* ent_mask; const uint8_t* ent_mask0; const uint8_t* group_bits;
    const uint8_t* gt_mask; long gt_sB, gt_sT;
    int nep;  // pow2 >= ne
};

struct AttnSmem {
    float *Qs, *Ks, *Vs, *S, *P, *dS, *dOs, *dQs, *dKs, *dVs;
    uint8_t *emt, *em0, *gb, *om, *gt;
};

__host__ __device__ inline int attn_pitch(int hd) { return hd + 4; }

__host__ __device__ inline size_t attn_smem_bytes(int ne, int na, int hd, bool bwd) {
    const int pd = attn_pitch(hd), ps = ne + 1;
    size_t f = (size_t)(na + 2 * ne) * pd + 2 * (size_t)na * ps;
    if (bwd) f += (size_t)na * ps + (size_t)na * pd + (size_t)(na + 2 * ne) * pd;
    size_t bytes = f * 4 + 3 * (size_t)ne + 2 * (size_t)na * ne;
    return (bytes + 15) & ~(size_t)15;
}

__device__ inline AttnSmem carve(float* base, int ne, int na, int hd, bool bwd) {
    const int pd = attn_pitch(hd), ps = ne + 1;
    AttnSmem s;
    float* p = base;
    s.Qs = p; p += na * pd;
    s.Ks = p; p += ne * pd;
    s.Vs = p; p += ne * pd;
    s.dS = s.dOs = s.dQs = s.dKs = s.dVs = nullptr;
    if (bwd) {   // all hd-pitched (16-byte aligned) arrays first; dQs,dKs,dVs contiguous (zeroed together)
        s.dOs = p; p += na * pd;
        s.dQs = p; p += na * pd;
        s.dKs = p; p += ne * pd;
        s.dVs = p; p += ne * pd;
    }
    s.S = p; p += na * ps;
    s.P = p; p += na * ps;
    if (bwd) { s.dS = p; p += na * ps; }
    uint8_t* b = reinterpret_cast<uint8_t*>(p);
    s.emt = b; b += ne;
    s.em0 = b; b += ne;
    s.gb = b; b += ne;
    s.om = b; b += na * ne;
    s.gt = b;
    return s;
}

// true = pair (agent i, entity j) is masked out before the softmax
__device__ inline bool premask(int code, const AttnSmem& s, int ne, int i, int j) {
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

// load a [rows, hd] slice (column offset col0) of a row-major matrix into LDS rows of pitch hd+4
__device__ inline void load_tile(float* dst, const float* src, long row0, int rows, int ld, int col0, int hd, int tid) {
    const int pd = attn_pitch(hd), c4n = hd >> 2;
    for (int idx = tid; idx < rows * c4n; idx += ANT) {
        const int r = idx / c4n, c4 = idx % c4n;
        const float4 v = *reinterpret_cast<const float4*>(src + (row0 + r) * (long)ld + col0 + c4 * 4);
        *reinterpret_cast<float4*>(dst + r * pd + c4 * 4) = v;
    }
}

__device__ inline float dot_rows(const float* a, const float* b, int hd) {
    float s = 0.f;
    for (int c = 0; c < hd; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
    }
    return s;
}

__device__ inline void load_common(const AttnK& p, const AttnSmem& s, int r, int head, int tid, bool need_obs) {
    const int b = r / p.T1, t = r % p.T1;
    load_tile(s.Qs, p.Q, (long)r * p.na, p.na, p.ldq, head * p.hd, p.hd, tid);
    load_tile(s.Ks, p.K, (long)r * p.ne, p.ne, p.ldkv, head * p.hd, p.hd, tid);
    load_tile(s.Vs, p.V, (long)r * p.ne, p.ne, p.ldkv, head * p.hd, p.hd, tid);
    for (int j = tid; j < p.ne; j += ANT) {
        s.emt[j] = p.ent_mask ? p.ent_mask[(long)r * p.ne + j] : 0;
        s.em0[j] = p.ent_mask0 ? p.ent_mask0[(long)b * p.ne + j] : 0;
        s.gb[j] = p.group_bits ? p.group_bits[(long)b * p.ne + j] : 0;
    }
    if (need_obs) {
        const uint8_t* om = p.obs_mask + b * p.om_sB + t * p.om_sT;
        for (int idx = tid; idx < p.na * p.ne; idx += ANT) s.om[idx] = om[idx];
    }
    if (p.gt_mask) {
        const uint8_t* gt = p.gt_mask + b * p.gt_sB + t * p.gt_sT;
        for (int idx = tid; idx < p.na * p.ne; idx += ANT) s.gt[idx] = gt[idx];
    }
}

__device__ inline void compute_logits(const AttnK& p, const AttnSmem& s, int tid) {
    const int pd = attn_pitch(p.hd), ps = p.ne + 1;
    const float scale = sqrtf((float)p.hd);   // attention.py:18-19,54: logits / sqrt(head_dim)
    for (int idx = tid; idx < p.na * p.ne; idx += ANT) {
        const int i = idx / p.ne, j = idx % p.ne;
        s.S[i * ps + j] = dot_rows(s.Qs + i * pd, s.Ks + j * pd, p.hd) / scale;
    }
}

__device__ inline bool uses_obs(const AttnK& p) {
    bool u = false;
    for (int v = 0; v < p.nvar; ++v)
        u |= mask_uses_obs(p.var[v]);
    return u;
}

__global__ __launch_bounds__(ANT) void attn_fwd_kernel(AttnK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int r = blockIdx.x, head = blockIdx.y;
    const AttnSmem s = carve(smem, p.ne, p.na, p.hd, false);
    const int pd = attn_pitch(p.hd), ps = p.ne + 1, nep = p.nep;
    load_common(p, s, r, head, tid, uses_obs(p));
    __syncthreads();
    compute_logits(p, s, tid);
    __syncthreads();
    const int rows_per_pass = ANT / nep;
    const int c4n = p.hd >> 2;
    for (int v = 0; v < p.nvar; ++v) {
        const int code = p.var[v];
        for (int row0 = 0; row0 < p.na; row0 += rows_per_pass) {
            const int i = row0 + tid / nep, j = tid % nep;
            const bool valid = (i < p.na) && (j < p.ne);
            float sv = -INFINITY;
            if (valid && !premask(code, s, p.ne, i, j)) sv = s.S[i * ps + j];
            float m = sv;
            for (int off = nep >> 1; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, nep));
            const float e = (sv == -INFINITY) ? 0.f : expf(sv - m);
            float sum = e;
            for (int off = nep >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, nep);
            if (valid) s.P[i * ps + j] = sum > 0.f ? e / sum : 0.f;
        }
        __syncthreads();
        float* O = p.O + v * p.sO;
        for (int idx = tid; idx < p.na * c4n; idx += ANT) {
            const int i = idx / c4n, c4 = idx % c4n;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = 0; j < p.ne; ++j) {
                const float w = s.P[i * ps + j];
                const float4 vv = *reinterpret_cast<const float4*>(s.Vs + j * pd + c4 * 4);
                acc.x = fmaf(w, vv.x, acc.x); acc.y = fmaf(w, vv.y, acc.y);
                acc.z = fmaf(w, vv.z, acc.z); acc.w = fmaf(w, vv.w, acc.w);
            }
            *reinterpret_cast<float4*>(O + ((long)r * p.na + i) * p.ldo + head * p.hd + c4 * 4) = acc;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(ANT) void attn_bwd_kernel(AttnK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int r = blockIdx.x, head = blockIdx.y;
    const AttnSmem s = carve(smem, p.ne, p.na, p.hd, true);
    const int pd = attn_pitch(p.hd), ps = p.ne + 1, nep = p.nep;
    const float scale = sqrtf((float)p.hd);
    load_common(p, s, r, head, tid, uses_obs(p));
    for (int idx = tid; idx < (p.na + 2 * p.ne) * pd; idx += ANT) s.dQs[idx] = 0.f;   // dQs,dKs,dVs contiguous
    __syncthreads();
    compute_logits(p, s, tid);
    __syncthreads();
    const int rows_per_pass = ANT / nep;
    const int c4n = p.hd >> 2;
    for (int v = 0; v < p.nvar; ++v) {
        const int code = p.var[v];
        load_tile(s.dOs, p.dO + v * p.sO, (long)r * p.na, p.na, p.ldo, head * p.hd, p.hd, tid);
        __syncthreads();
        for (int row0 = 0; row0 < p.na; row0 += rows_per_pass) {
            const int i = row0 + tid / nep, j = tid % nep;
            const bool valid = (i < p.na) && (j < p.ne);
            float sv = -INFINITY;
            if (valid && !premask(code, s, p.ne, i, j)) sv = s.S[i * ps + j];
            float m = sv;
            for (int off = nep >> 1; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, nep));
            const float e = (sv == -INFINITY) ? 0.f : expf(sv - m);
            float sum = e;
            for (int off = nep >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, nep);
            const float pw = sum > 0.f ? e / sum : 0.f;
            // dP_ij = dO_i . V_j ; dS = P (dP - sum_j P dP)   (softmax backward; masked entries have P = 0)
            float dp = 0.f;
            if (valid && pw != 0.f) dp = dot_rows(s.dOs + i * pd, s.Vs + j * pd, p.hd);
            float rd = pw * dp;
            for (int off = nep >> 1; off > 0; off >>= 1) rd += __shfl_xor(rd, off, nep);
            if (valid) {
                s.P[i * ps + j] = pw;
                s.dS[i * ps + j] = pw * (dp - rd) / scale;
            }
        }
        __syncthreads();
        // dV_j += sum_i P_ij dO_i ; dK_j += sum_i dS_ij Q_i
        for (int idx = tid; idx < p.ne * c4n; idx += ANT) {
            const int j = idx / c4n, c4 = idx % c4n;
            float4 av = *reinterpret_cast<float4*>(s.dVs + j * pd + c4 * 4);
            float4 ak = *reinterpret_cast<float4*>(s.dKs + j * pd + c4 * 4);
            for (int i = 0; i < p.na; ++i) {
                const float w = s.P[i * ps + j], g = s.dS[i * ps + j];
                const float4 d = *reinterpret_cast<const float4*>(s.dOs + i * pd + c4 * 4);
                const float4 q = *reinterpret_cast<const float4*>(s.Qs + i * pd + c4 * 4);
                av.x = fmaf(w, d.x, av.x); av.y = fmaf(w, d.y, av.y); av.z = fmaf(w, d.z, av.z); av.w = fmaf(w, d.w, av.w);
                ak.x = fmaf(g, q.x, ak.x); ak.y = fmaf(g, q.y, ak.y); ak.z = fmaf(g, q.z, ak.z); ak.w = fmaf(g, q.w, ak.w);
            }
            *reinterpret_cast<float4*>(s.dVs + j * pd + c4 * 4) = av;
            *reinterpret_cast<float4*>(s.dKs + j * pd + c4 * 4) = ak;
        }
        // dQ_i += sum_j dS_ij K_j
        for (int idx = tid; idx < p.na * c4n; idx += ANT) {
            const int i = idx / c4n, c4 = idx % c4n;
            float4 aq = *reinterpret_cast<float4*>(s.dQs + i * pd + c4 * 4);
            for (int j = 0; j < p.ne; ++j) {
                const float g = s.dS[i * ps + j];
                const float4 k = *reinterpret_cast<const float4*>(s.Ks + j * pd + c4 * 4);
                aq.x = fmaf(g, k.x, aq.x); aq.y = fmaf(g, k.y, aq.y); aq.z = fmaf(g, k.z, aq.z); aq.w = fmaf(g, k.w, aq.w);
            }
            *reinterpret_cast<float4*>(s.dQs + i * pd + c4 * 4) = aq;
        }
        __syncthreads();
    }
    for (int idx = tid; idx < p.na * c4n; idx += ANT) {
        const int i = idx / c4n, c4 = idx % c4n;
        *reinterpret_cast<float4*>(p.dQ + ((long)r * p.na + i) * p.ldq + head * p.hd + c4 * 4) =
            *reinterpret_cast<const float4*>(s.dQs + i * pd + c4 * 4);
    }
    for (int idx = tid; idx < p.ne * c4n; idx += ANT) {
        const int j = idx / c4n, c4 = idx % c4n;
        const long off = ((long)r * p.ne + j) * p.ldkv + head * p.hd + c4 * 4;
        *reinterpret_cast<float4*>(p.dK + off) = *reinterpret_cast<const float4*>(s.dKs + j * pd + c4 * 4);
        *reinterpret_cast<float4*>(p.dV + off) = *reinterpret_cast<const float4*>(s.dVs + j * pd + c4 * 4);
    }
}

static int fill(const refil_attn_desc& d, AttnK& k, bool bwd) {
    REFIL_CHECK(d.Q && d.K && d.V, "refil_attn: null Q/K/V");
    REFIL_CHECK(d.ne >= 1 && d.ne <= 64 && d.na >= 1 && d.na <= d.ne, "refil_attn: need 1 <= na <= ne <= 64 (ne=%d na=%d)", d.ne, d.na);
    REFIL_CHECK(d.hd >= 4 && d.hd % 4 == 0 && d.hd <= 128, "refil_attn: head dim %d must be a multiple of 4 in [4,128]", d.hd);
    REFIL_CHECK(d.nvar >= 1 && d.nvar <= 3, "refil_attn: nvar must be 1..3");
    REFIL_CHECK(d.ldq % 4 == 0 && d.ldkv % 4 == 0 && d.ldo % 4 == 0, "refil_attn: leading dims must be multiples of 4");
    REFIL_CHECK(d.R > 0 && d.T1 > 0 && d.heads > 0, "refil_attn: bad R/T1/heads");
    bool need_obs = false, need_grp = false, need_emt = false, need_gt = false, need_e0 = false;
    for (int v = 0; v < d.nvar; ++v) {
        REFIL_CHECK(d.var[v] >= 0 && d.var[v] < REFIL_MASK_COUNT, "refil_attn: bad mask code %d", d.var[v]);
        need_gt |= mask_uses_gt(d.var[v]);
        need_obs |= mask_uses_obs(d.var[v]);
        need_grp |= mask_uses_groups(d.var[v]);
        need_e0 |= mask_uses_inactive0(d.var[v]);
        need_emt |= d.var[v] == REFIL_MASK_ENTITY;
    }
    REFIL_CHECK(!need_gt || d.gt_mask, "refil_attn: gt_mask required by a ground-truth-factor mask variant");
    REFIL_CHECK(!need_e0 || d.ent_mask0, "refil_attn: ent_mask0 required by a mask variant");
    REFIL_CHECK(!need_obs || d.obs_mask, "refil_attn: obs_mask required by a mask variant");
    REFIL_CHECK(!need_grp || (d.ent_mask0 && d.group_bits), "refil_attn: ent_mask0/group_bits required by a mask variant");
    REFIL_CHECK(!need_emt || d.ent_mask, "refil_attn: ent_mask required by a mask variant");
    if (bwd) REFIL_CHECK(d.dO && d.dQ && d.dK && d.dV, "refil_attn_backward: null gradient pointer");
    else REFIL_CHECK(d.O, "refil_attn_forward: null output");
    k.Q = d.Q; k.K = d.K; k.V = d.V; k.O = d.O; k.dO = d.dO; k.dQ = d.dQ; k.dK = d.dK; k.dV = d.dV;
    k.ldq = d.ldq; k.ldkv = d.ldkv; k.ldo = d.ldo; k.sO = d.sO;
    k.R = d.R; k.T1 = d.T1; k.ne = d.ne; k.na = d.na; k.heads = d.heads; k.hd = d.hd; k.nvar = d.nvar;
    for (int v = 0; v < 3; ++v) k.var[v] = d.var[v];
    k.obs_mask = d.obs_mask; k.om_sB = d.om_sB; k.om_sT = d.om_sT;
    k.ent_mask = d.ent_mask; k.ent_mask0 = d.ent_mask0; k.group_bits = d.group_bits;
    k.gt_mask = d.gt_mask; k.gt_sB = d.gt_sB; k.gt_sT = d.gt_sT;
    int nep = 1;
    while (nep < d.ne) nep <<= 1;
    k.nep = nep;
    return 0;
}

int attn_mfma_launch(const refil_attn_desc& d, bool bwd, hipStream_t st);   // attention_mfma.hip; -1 = shape not instantiated

static bool force_valu() {
    const char* e = getenv("REFIL_ATTN_VALU");      // debugging / A-B switch: run the generic VALU kernels
    return e && e[0] == '1';
}

int attn_forward_launch(const refil_attn_desc& d, hipStream_t st) {
    AttnK k;
    if (int e = fill(d, k, false)) return e;
    if (!force_valu()) {
        const int rc = attn_mfma_launch(d, false, st);
        if (rc >= 0) return rc;
    }
    REFIL_CHECK(!d.mask_words, "refil_attn_forward: precomputed mask words need a shape the MFMA kernels take");
    const size_t smem = attn_smem_bytes(d.ne, d.na, d.hd, false);
    const double unit = (double)d.R * d.heads * d.na * d.ne * d.hd;
    ProfScope prof("attn_fwd_kernel", unit * (2.0 + 2.0 * d.nvar),
                   4.0 * d.R * d.heads * d.hd * (d.na * (1.0 + d.nvar) + 2.0 * d.ne), st);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(d.R, d.heads), dim3(ANT), smem, st, k);
    REFIL_LAUNCH_CHECK();
    return 0;
}

int attn_backward_launch(const refil_attn_desc& d, hipStream_t st) {
    AttnK k;
    if (int e = fill(d, k, true)) return e;
    if (!force_valu()) {
        const int rc = attn_mfma_launch(d, true, st);
        if (rc >= 0) return rc;
    }
    REFIL_CHECK(!d.mask_words, "refil_attn_backward: precomputed mask words need a shape the MFMA kernels take");
    const size_t smem = attn_smem_bytes(d.ne, d.na, d.hd, true);
    REFIL_CHECK(smem <= 160 * 1024, "refil_attn_backward: LDS need %zu B exceeds 160 KiB", smem);
    if (smem > 64 * 1024) {
        REFIL_HIP(hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    const double unit = (double)d.R * d.heads * d.na * d.ne * d.hd;
    ProfScope prof("attn_bwd_kernel", unit * (2.0 + 8.0 * d.nvar),
                   4.0 * d.R * d.heads * d.hd * (d.na * (2.0 + d.nvar) + 4.0 * d.ne), st);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(d.R, d.heads), dim3(ANT), smem, st, k);
    REFIL_LAUNCH_CHECK();
    return 0;
}

}  // namespace refil

extern "C" int refil_attn_mask_words(const refil_attn_desc* desc, void* mask_words, void* row_bits, void* stream) {
    REFIL_CHECK(desc, "refil_attn_mask_words: null desc");
    return refil::attn_mask_words_launch(*desc, static_cast<unsigned long long*>(mask_words), static_cast<unsigned long long*>(row_bits),
                                         (hipStream_t)stream);
}

extern "C" int refil_attn_forward(const refil_attn_desc* desc, void* stream) {
    REFIL_CHECK(desc, "refil_attn_forward: null desc");
    return refil::attn_forward_launch(*desc, (hipStream_t)stream);
}
extern "C" int refil_attn_backward(const refil_attn_desc* desc, void* stream) {
    REFIL_CHECK(desc, "refil_attn_backward: null desc");
    return refil::attn_backward_launch(*desc, (hipStream_t)stream);
}

extern "C" int refil_pool_forward(const refil_attn_desc* desc, int32_t mode, void* stream) {
    REFIL_CHECK(desc, "refil_pool_forward: null desc");
    return refil::pool_launch(*desc, mode, false, (hipStream_t)stream);
}
extern "C" int refil_pool_backward(const refil_attn_desc* desc, int32_t mode, void* stream) {
    REFIL_CHECK(desc, "refil_pool_backward: null desc");
    return refil::pool_launch(*desc, mode, true, (hipStream_t)stream);
}
