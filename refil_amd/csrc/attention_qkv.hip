// EntityAttentionLayer forward from the layer's INPUT: in_trans and the masked multi-head attention core in ONE launch
// (reference: src/modules/layers/attention.py:46-64 -- `query, key, value = in_trans(x).chunk(3)`, logits, masked softmax, .V,
// heads merged; the target networks of src/learners/q_learner.py:111-113,154 are never differentiated, so for them Q / K / V
// exist in registers only).
//
// Why. As separate launches the projection writes Q / K / V to HBM and the attention core reads them back: at the north-star
// shape 250 MB + 45 MB written and 186 MB read per hypernet set and step -- a fifth of the step's HBM traffic -- for values that
// are consumed once, by the wave that could have produced them.
//
// How. A workgroup owns ONE (net, head): its 3 hd rows of in_trans.weight (Wq_h, Wk_h, Wv_h), split once into three bf16 planes
// (the exact 3-way split of split.h: six bf16 matrix-pipe products with fp32 accumulate = the fp32 product), stay in LDS; its
// waves walk the live (b,t) rows. The orientation of each product is chosen so that the accumulators ARE the operands of the
// fp32 16x16x4 core (attention_mfma.hip), lane for lane:
//     K^T[c][key]   = Wk_h x^T   (A = W fragment, B = x rows)  -> lane (key, c = 16ct+4q+reg): the A operand of S^T = K Q^T
//     Q^T[c][agent] = Wq_h x^T   (same x registers: the agents are the first entities)  -> the B operand of S^T = K Q^T
//     V[key][c]     = x Wv_h^T   (A = the SAME x registers, B = W fragment) -> lane (c, key = 16jt+4q+reg): the A operand of O^T = V^T P^T
// (v_mfma_f32_16x16x32_bf16: A and B fragments have the same lane layout -- index = lane % 16, 8 consecutive reduction indices
// per lane group -- so one split of an x row serves as A and as B). Nothing but the W planes touches LDS; the x rows stream
// global -> registers one job ahead; dead rows (refil_attn_desc.kv_dead / t_last) are not fetched. For the live nets the three
// projections are also stored once, in refil_attn_desc's Q / K / V layouts, for refil_attn_backward.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "bufops.h"
#include "common.h"
#include "kernels.h"
#include "profile.h"
#include "split.h"
#include "../../include/refil_hip.h"

namespace refil {

#ifdef REFIL_QKV_TIMING
// debug build only (REFIL_EXTRA_FLAGS=-DREFIL_QKV_TIMING; tools/probes/qkv_timing.py): per-wave cycle sums (shader clock) of the job phases
__device__ unsigned long long g_qkv_dbg[4096 * 8];
#define QKV_TICK(acc_) { const unsigned long long t1_ = __builtin_readcyclecounter(); acc_ += t1_ - t0_; t0_ = t1_; }
#else
#define QKV_TICK(acc_)
#endif

constexpr int QKV_MAX_NETS = 8;
// waves per workgroup of an instantiation: a third key tile or a second agent tile needs more than the 256 registers of two waves per SIMD
constexpr int qkv_waves(int njt, int nat) { return njt + nat >= 4 ? 4 : 8; }
struct QkvNet {
    const float* X;      // layer input of this net: entity row (r ne + j) at X + (r ne + j) ldx
    const float* W;      // in_trans.weight [3w][w], rows [0,w) -> Q, [w,2w) -> K, [2w,3w) -> V
    float* O;
    float* Qo; float* Ko; float* Vo;     // optional stores (layouts of refil_attn_desc Q / K / V)
    int nvar, sum_agents;
};
struct QkvM {
    QkvNet net[QKV_MAX_NETS]; int nnets;
    int ldx, w, ldq, ldkv, ldo; long sO;
    int R, T1, ne, na, heads, hd, nvar;
    const int* t_last;
    const unsigned long long* mwords; int mw_nvar; const unsigned long long* rbits;
    float* nact; int zero_dead;
    int nslices, ngroups, xcd_groups;    // workgroup -> (slice = (net, head), row group); xcd_groups > 0: row groups per XCD (XCD-aware map)
    int stagger;                         // how the two waves of a SIMD are kept out of phase: 0 nothing, 1 static priority, 2 a delayed start (REFIL_QKV_STAGGER)
};

#define MFMA16F(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ inline float q_cross4_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ inline float q_cross4_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }
template <int CTRL>
__device__ inline float q_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float q_group16_sum(float v) { v += q_dpp<0xB1>(v); v += q_dpp<0x4E>(v); v += q_dpp<0x141>(v); v += q_dpp<0x140>(v); return v; }

// masked softmax of transposed logits (attention_mfma.hip: softmax_T): st[jt][reg] = S^T[key 16jt+4q+reg][agent], in place -> P^T
template <int NJT>
__device__ inline void qkv_softmax_T(f32x4 (&st)[NJT], unsigned long long w, int q) {
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
    mx = q_cross4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float e = st[jt][reg] == -INFINITY ? 0.f : __expf(st[jt][reg] - mx);
            st[jt][reg] = e;
            sum += e;
        }
    sum = q_cross4_sum(sum);
    const float inv = sum > 0.f ? __builtin_amdgcn_rcpf(sum) : 0.f;     // fully masked row -> 0 (attention.py:60 NaN -> 0)
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) st[jt][reg] *= inv;
}

__device__ inline unsigned long long q_readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

struct QRow { int r; unsigned long long kdw, qdw, emtw; };

// 8 fp32 values (two float4 of one x row) -> the three bf16x8 pieces hi / mid / lo
// (residuals as single v_sub_f32: the SLP-packed v_pk_add_f32 costs matrix-pipe cycles beside an MFMA, split.h: wr_sub)
__device__ inline void split2(float x, float y, unsigned& h, unsigned& m, unsigned& l) {
    h = wr_pk(x, y);
    x = wr_sub(x, __uint_as_float(h << 16)); y = wr_sub(y, __uint_as_float(h & 0xFFFF0000u));
    m = wr_pk(x, y);
    x = wr_sub(x, __uint_as_float(m << 16)); y = wr_sub(y, __uint_as_float(m & 0xFFFF0000u));
    l = wr_pk(x, y);
}
__device__ inline void split8(const float4& a0, const float4& a1, wr_bf16x8 (&o)[3]) {
    unsigned h[4], m[4], l[4];
    split2(a0.x, a0.y, h[0], m[0], l[0]); split2(a0.z, a0.w, h[1], m[1], l[1]);
    split2(a1.x, a1.y, h[2], m[2], l[2]); split2(a1.z, a1.w, h[3], m[3], l[3]);
    o[0] = __builtin_bit_cast(wr_bf16x8, wr_u32x4{h[0], h[1], h[2], h[3]});
    o[1] = __builtin_bit_cast(wr_bf16x8, wr_u32x4{m[0], m[1], m[2], m[3]});
    o[2] = __builtin_bit_cast(wr_bf16x8, wr_u32x4{l[0], l[1], l[2], l[3]});
}

// LDS bytes of an instantiation: the W planes [3 planes][3 matrices (q, k, v)][NCT][NKS][1 KB: lane (c = lane % 16, k group = lane / 16) -> 8 bf16]
constexpr size_t qkv_plane_bytes(int nct, int nks) { return (size_t)3 * nct * nks * 1024; }

// NJT: 16-entity key tiles (ne <= 16 NJT, up to 3), NCT: 16-channel tiles of a head (hd = 16 NCT), NKS: 32-index steps of the reduction
// (w = 32 NKS), NAT: 16-agent query tiles (n_agents <= 16 NAT <= 16 NJT; the agents are the first entities, so the queries of agent tile
// `at` are projected from the x registers of key tile `at`). Three key tiles hold 96 registers of x rows in flight alone: the forms with a
// third key tile or a second agent tile run four waves per workgroup (one per SIMD, 512 registers), the others eight.
//
// Key compaction (NJT > 1). Only the entities that are alive as keys (row_bits: ~kv_dead) take part: they are packed, in entity order,
// into the first `cnt` tile positions -- the agents, being the first entities, land in tile 0 -- and the dead ones behind them, so a row
// with <= 16 live entities costs ONE key tile of K^T / V / S^T / P V work instead of two (SC2-law batches: 40-55 % of the rows). The
// permutation is built per row with one ds_permute (position of every entity: rank among the live / the dead ones) and read back with
// ds_bpermute: ek = the entity at position 16 jt + lane % 16 (its x row, its K row, as agent: its mask word and its Q / O row),
// ev = the entities at positions 16 jt + 4 (lane / 16) + reg (their mask bits, their V rows). Attention is invariant under a permutation of
// the keys; per agent it is a relabelling of the output rows.
// STORE: some net of the launch keeps its Q / K / V (a launch of target nets only carries no store code at all).
template <int NJT, int NCT, int NKS, bool STORE, int NAT>
__global__ __launch_bounds__(64 * qkv_waves(NJT, NAT), 1) void attn_qkv_fwd(QkvM p) {
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    static_assert(NAT >= 1 && NAT <= NJT && NJT <= 3, "agent tiles are key tiles");
    constexpr int QKV_WAVES = qkv_waves(NJT, NAT);
    constexpr int NAP = 16 * NAT;                // agents per row of the mask-word table (refil_attn_mask_words pads to 16)
    constexpr size_t PSZ = qkv_plane_bytes(NCT, NKS);
    constexpr bool COMPACT = NJT > 1;
    // (<= 32 entities: the low half of the 64-bit mask word is all a lane needs)
    using mw_t = std::conditional_t<(NJT > 2), unsigned long long, unsigned>;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
#ifdef REFIL_QKV_TIMING
    const unsigned long long t_kernel0 = __builtin_readcyclecounter();
#endif
    int slice, grp;
    {
        const int b = blockIdx.x;
        if (p.xcd_groups > 0) { const int xcd = b & 7, j = b >> 3; slice = j % p.nslices; grp = xcd * p.xcd_groups + j / p.nslices; }
        else { slice = b % p.nslices; grp = b / p.nslices; }
    }
    const int net = slice / p.heads, head = slice - net * p.heads;
    const QkvNet& n = p.net[net];
    const int hd = 16 * NCT, w = 32 * NKS;
    char* planes = smem_q;
    int* pref = reinterpret_cast<int*>(smem_q + 3 * PSZ);

    // ---- stage this (net, head)'s three weight slices once: split into bf16 planes in fragment order (loads batched: one memory
    // latency for the whole slice instead of one per 16 bytes and thread) ----
    {
        constexpr int NTH = 64 * QKV_WAVES, W4 = 8 * NKS, TOTAL = 3 * 16 * NCT * W4, NIT = (TOTAL + NTH - 1) / NTH;
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NTH + tid, ic = idx < TOTAL ? idx : 0;
            const int k4 = ic % W4, row = ic / W4, m = row / hd, c = row - m * hd;
            v[it] = *reinterpret_cast<const float4*>(n.W + ((long)m * w + head * hd + c) * w + 4 * k4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NTH + tid;
            if (idx < TOTAL) {
                const int k4 = idx % W4, row = idx / W4, m = row / hd, c = row - m * hd;
                unsigned h0, m0, l0, h1, m1, l1;
                wr_split(v[it].x, v[it].y, h0, m0, l0); wr_split(v[it].z, v[it].w, h1, m1, l1);
                const int k = 4 * k4, s = k >> 5, qq = (k >> 3) & 3, e = k & 7;
                char* dst = planes + ((size_t)((m * NCT + (c >> 4)) * NKS + s) << 10) + qq * 256 + (c & 15) * 16 + e * 2;
                *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(dst + PSZ) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(dst + 2 * PSZ) = make_uint2(l0, l1);
            }
        }
    }
    // ---- live rows: prefix sum of the episodes' live steps (t <= t_last[b]) ----
    const int nB = p.R / p.T1;
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
    }
    __syncthreads();
    if (p.t_last) nlive = pref[nB];
    const int wstride = p.ngroups * QKV_WAVES;
    const int ord0 = grp * QKV_WAVES + wave;
    const int njobs = ord0 < nlive ? (nlive - ord0 + wstride - 1) / wstride : 0;
    // the rows of this workgroup's jobs, resolved once (live ordinal -> (b,t) through the prefix sums): rows[wave][job]
    const int maxjobs = (p.R + wstride - 1) / wstride;
    int* rows = pref + nB + 2;
    for (int i = tid; i < QKV_WAVES * maxjobs; i += 64 * QKV_WAVES) {
        const int wv = i / maxjobs, jb = i - wv * maxjobs;
        int o = grp * QKV_WAVES + wv + jb * wstride;
        o = o < nlive ? o : (nlive > 0 ? nlive - 1 : 0);
        int rr = o;
        if (p.t_last) {
            int lo = 0, hi = nB;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (pref[mid + 1] <= o) lo = mid + 1; else hi = mid; }
            rr = lo * p.T1 + (o - pref[lo]);
        }
        rows[i] = rr;
    }
    __syncthreads();
    if (njobs == 0) return;
    const int* myrows = rows + wave * maxjobs;
    auto row_of = [&](int k) -> int { return __builtin_amdgcn_readfirstlane(myrows[k < njobs ? k : njobs - 1]); };
    const float inv_scale = 1.0f / sqrtf((float)hd);
    const unsigned long long na_bits = (p.na >= 64) ? ~0ull : ((1ull << p.na) - 1ull);
    const unsigned long long ne_bits = (p.ne >= 64) ? ~0ull : ((1ull << p.ne) - 1ull);
    const int col0 = head * hd;

    // the key permutation of a row (see above); identity without compaction
    struct QMap { int ek[NJT]; unsigned evp[NJT]; int cnt; };
    auto map_of = [&](unsigned long long kdw) -> QMap {
        QMap m;
        if (!COMPACT) {
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) {
                m.ek[jt] = 16 * jt + l15;
                const unsigned e0 = 16 * jt + 4 * q;
                m.evp[jt] = e0 | ((e0 + 1) << 8) | ((e0 + 2) << 16) | ((e0 + 3) << 24);
            }
            m.cnt = 16 * NJT;
            return m;
        }
        const unsigned long long L = ~kdw & ne_bits;
        m.cnt = __popcll(L);
        const unsigned long long below = (1ull << lane) - 1ull;
        const bool live = (L >> lane) & 1ull;
        const int pos = live ? __popcll(L & below) : m.cnt + __popcll(~L & below);
        const int T = __builtin_amdgcn_ds_permute(pos << 2, lane);          // lane `pos` <- this lane's entity
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            m.ek[jt] = __builtin_amdgcn_ds_bpermute((16 * jt + l15) << 2, T);
            unsigned pk = 0;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) pk |= (unsigned)__builtin_amdgcn_ds_bpermute((16 * jt + 4 * q + reg) << 2, T) << (8 * reg);
            m.evp[jt] = pk;
        }
        return m;
    };

    // operands in flight for the NEXT job: the x rows of every key tile and reduction step (refilled in place, step by step, as the
    // current job's projection consumes them: a whole projection + core of flight time), the mask words of this lane's agent
    float4 xr[NKS][NJT][2];
    mw_t nw[3][NAT];
    auto fetch_x = [&](auto s_, const QRow& ri, const QMap& fm, bool valid) {
        constexpr int s = decltype(s_)::value;
        const rsrc_t rx = mk_rsrc(n.X + (long)ri.r * p.ne * p.ldx, valid ? ((long)(p.ne - 1) * p.ldx + w) * 4 : 0);
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int key = fm.ek[jt];
            // (compacted: the live entities are exactly the first cnt tile positions)
            const bool ok = COMPACT ? 16 * jt + l15 < fm.cnt : (key < p.ne && !((ri.kdw >> key) & 1ull));
            const int off = (key * p.ldx + 32 * s + 8 * q) * 4;
            xr[s][jt][0] = buf_ld4(rx, ok ? off : BUF_OOB);
            xr[s][jt][1] = buf_ld4(rx, ok ? off + 16 : BUF_OOB);
        }
    };
    auto fetch_w = [&](const QRow& ri, const QMap& fm, bool valid) {
        const rsrc_t rw = mk_rsrc(p.mwords + (long)ri.r * p.mw_nvar * NAP, valid ? (long)p.mw_nvar * NAP * 8 : 0);
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int at = 0; at < NAT; ++at) {
                const int off = v < p.nvar && fm.ek[at] < NAP ? (v * NAP + fm.ek[at]) * 8 : BUF_OOB;
                if constexpr (NJT > 2) nw[v][at] = buf_ld_u64(rw, off);
                else nw[v][at] = __builtin_amdgcn_raw_buffer_load_b32(rw, off, 0, 0);
            }
    };
    // the row words of a row travel one job ahead of its operand fetch, in a vector register (lane l holds word l % 3)
    auto words_of = [&](int r) -> unsigned long long { return p.rbits[3 * (long)r + lane % 3]; };
    auto take = [&](int r, unsigned long long wv) -> QRow {
        QRow x; x.r = r; x.kdw = q_readlane64(wv, 0); x.qdw = q_readlane64(wv, 1); x.emtw = q_readlane64(wv, 2);
        // an agent that is alive as a query needs its x row (Q^T comes from the same registers as K^T / V) whatever row_bits says about
        // it as a key: it is fetched, projected and -- unless its mask bits exclude it -- attended to like a live key
        // (refil_hip.h, refil_attn_qkv_forward; the learner's row lists never produce such a row)
        x.kdw &= x.qdw | ~na_bits;
        return x;
    };
    int r_next = row_of(0);
    unsigned long long w_next = words_of(r_next);
    QRow frow = take(r_next, w_next);
    QMap fmap = map_of(frow.kdw);
    r_next = row_of(1); w_next = words_of(r_next);
    static_for<NKS>([&](auto s_) { fetch_x(s_, frow, fmap, true); });
    fetch_w(frow, fmap, true);
    // The two waves of a SIMD start their jobs in lockstep. Keeping them out of phase -- a static priority for the second-dispatched wave
    // or a delayed start -- was measured and changes nothing (the matrix pipe is ~83 % busy either way: profiles/r05_qkv_timing.txt,
    // r05_qkv_stagger.txt), and a raised priority takes issue slots from the OTHER chain's kernels on the same SIMD: off by default
    if (wave >= QKV_WAVES / 2) {
        if (p.stagger == 1) __builtin_amdgcn_s_setprio(1);
        else if (p.stagger >= 2) { for (int i = 0; i < 24 * (p.stagger - 1); ++i) __builtin_amdgcn_s_sleep(2); }       // (~128 cycles per s_sleep 2)
    }

#ifdef REFIL_QKV_TIMING
    unsigned long long t_top = 0, t_proj = 0, t_st = 0, t_core = 0, t_pro = 0, t0_ = __builtin_readcyclecounter();
    const unsigned long long t_begin = t0_;
#endif
    // memory operations a job issues BEHIND the fetch of its successor (all buffer stores, the missing ones out of range): the prologue
    // issues as many dropped stores behind the first fetch, so that the operand wait at the top of the loop is the same count on both
    // ways into it (vmcnt retires in order and the compiler takes the smaller count of the incoming paths: attention_mfma.hip)
    constexpr int NSTORE = 1 + (STORE ? NCT * NJT + NCT * NAT + 4 * NJT * NCT : 0) + 6 * NCT * NAT + NCT;
    {
        const rsrc_t none = mk_rsrc(n.O, 0);
#pragma unroll
        for (int i = 0; i < NSTORE; ++i) buf_st4(none, BUF_OOB - 16 * i, f32x4{0.f, 0.f, 0.f, 0.f});
    }

    for (int k = 0; k < njobs; ++k) {
        const QRow crow = frow;
        const QMap cmap = fmap;
        const int r = crow.r;
        // this lane's agent of agent tile `at`: the entity at tile position lane % 16 of key tile `at`, when it is an agent that is alive
        // as a query (the live agents, being the first entities, sit in the first positions: all of them within the first NAT tiles)
        bool agent_ok[NAT];
        mw_t cw[3][NAT];
#pragma unroll
        for (int at = 0; at < NAT; ++at) {
            const int agent = cmap.ek[at];
            agent_ok[at] = 16 * at + l15 < cmap.cnt && agent < p.na && !((crow.qdw >> agent) & 1ull);
#pragma unroll
            for (int v = 0; v < 3; ++v) cw[v][at] = agent_ok[at] ? nw[v][at] : ~(mw_t)0;
        }
        // the row after this one: its words arrived a job ago; its key permutation is built beside the projection below
        frow = take(r_next, w_next);
        fmap = map_of(frow.kdw);
        r_next = row_of(k + 2); w_next = words_of(r_next);
        const bool nvalid = k + 1 < njobs;
        auto job = [&](auto nt_) {
            constexpr int NT = decltype(nt_)::value;        // key tiles this row needs (1 .. NJT)
            constexpr int NA = NAT < NT ? NAT : NT;         // agent tiles this row needs (its live agents sit within its live positions)
            // ---- in_trans of this row's entities for this head: K^T, V, Q^T tiles on the bf16 pipe (fp32 accumulate) ----
            f32x4 Kt[NCT][NT], Vv[NT][NCT], Qt[NCT][NA];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                for (int at = 0; at < NA; ++at) Qt[ct][at] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) { Kt[ct][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; Vv[jt][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            // product order: smallest first -- (lo, hi), (hi, lo), (mid, mid), (mid, hi), (hi, mid), (hi, hi); planes 0 / 1 / 2 = hi / mid / lo
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
            QKV_TICK(t_top)
            static_for<NKS>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                wr_bf16x8 xs[NT][3];
#pragma unroll
                for (int jt = 0; jt < NT; ++jt) split8(xr[s][jt][0], xr[s][jt][1], xs[jt]);
                fetch_x(s_, frow, fmap, nvalid);            // (this step's x registers are free: the next row's come in)
                const char* pb = planes + ((size_t)s << 10) + lane * 16;
                // (one matrix's fragments at a time: 12 registers per channel tile instead of 36 -- the 256-register budget of two waves per SIMD)
                {
                    wr_bf16x8 wk[NCT][3];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wk[ct][pl] = *reinterpret_cast<const wr_bf16x8*>(pb + pl * PSZ + ((size_t)((1 * NCT + ct) * NKS) << 10));
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                            for (int jt = 0; jt < NT; ++jt) Kt[ct][jt] = MFMA16B(wk[ct][PA[pr]], xs[jt][PB[pr]], Kt[ct][jt]);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    wr_bf16x8 wq[NCT][3];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wq[ct][pl] = *reinterpret_cast<const wr_bf16x8*>(pb + pl * PSZ + ((size_t)((0 * NCT + ct) * NKS) << 10));
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                            for (int at = 0; at < NA; ++at) Qt[ct][at] = MFMA16B(wq[ct][PA[pr]], xs[at][PB[pr]], Qt[ct][at]);
                }
                __builtin_amdgcn_sched_barrier(0);
                wr_bf16x8 wv[NCT][3];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        wv[ct][pl] = *reinterpret_cast<const wr_bf16x8*>(pb + pl * PSZ + ((size_t)((2 * NCT + ct) * NKS) << 10));
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) Vv[jt][ct] = MFMA16B(xs[jt][PB[pr]], wv[ct][PA[pr]], Vv[jt][ct]);
                __builtin_amdgcn_sched_barrier(0);
            });
            fetch_w(frow, fmap, nvalid);
            QKV_TICK(t_proj)
            // queries of inactive agents and of tile positions that hold no agent enter the core as zeros (refil_attn_desc.q_dead)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int at = 0; at < NA; ++at)
                    if (!agent_ok[at]) Qt[ct][at] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (nact[r] by lane 0 of the row's first slice)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)__popcll(~crow.emtw & na_bits)),
                                                  mk_rsrc(p.nact + r, p.nact && slice == 0 ? 4 : 0), lane == 0 ? 0 : BUF_OOB, 0, 0);
            // ---- the live nets keep their projections for the backward (layouts of refil_attn_desc Q / K / V) ----
            if constexpr (STORE) {
                const rsrc_t rk = mk_rsrc(n.Ko + (long)r * p.ne * p.ldkv, n.Ko ? (long)p.ne * p.ldkv * 4 : 0);
                const rsrc_t rv = mk_rsrc(n.Vo + (long)r * p.ne * p.ldkv, n.Vo ? (long)p.ne * p.ldkv * 4 : 0);
                const rsrc_t rq = mk_rsrc(n.Qo + (long)r * p.na * p.ldq, n.Qo ? (long)p.na * p.ldq * 4 : 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int c = col0 + 16 * ct + 4 * q;
#pragma unroll
                    for (int jt = 0; jt < NJT; ++jt) {
                        const int key = cmap.ek[jt];
                        const bool ok = jt < NT && (COMPACT ? 16 * jt + l15 < cmap.cnt : (key < p.ne && !((crow.kdw >> key) & 1ull)));
                        buf_st4(rk, ok ? (key * p.ldkv + c) * 4 : BUF_OOB, Kt[ct][jt < NT ? jt : 0]);
                    }
#pragma unroll
                    for (int at = 0; at < NAT; ++at)
                        buf_st4(rq, at < NA && agent_ok[at] ? (cmap.ek[at] * p.ldq + c) * 4 : BUF_OOB, Qt[ct][at < NA ? at : 0]);
                    __builtin_amdgcn_sched_barrier(0);      // (address arithmetic stays beside its store: the compiler otherwise computes all of them first)
                }
#pragma unroll
                for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int key = (cmap.evp[jt] >> (8 * reg)) & 0xff;
                            const bool ok = jt < NT && (COMPACT ? 16 * jt + 4 * q + reg < cmap.cnt : (key < p.ne && !((crow.kdw >> key) & 1ull)));
                            // (__float_as_uint of a copy: __builtin_bit_cast of ONE element of a vector reads element 0 in this clang)
                            const float vv = Vv[jt < NT ? jt : 0][ct][reg];
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv), rv,
                                                                  ok ? (key * p.ldkv + col0 + 16 * ct + l15) * 4 : BUF_OOB, 0, 0);
                            if (reg == 3) __builtin_amdgcn_sched_barrier(0);
                        }
            }
            QKV_TICK(t_st)
            // ---- attention core (attention_mfma.hip: attn_fwd_pipe), operands straight from the accumulators ----
            // dropped: the entities a short row leaves outside its NT key tiles -- all of them dead as keys. A dead K / V row is a ZERO row,
            // not an absent one (refil_attn_desc.kv_dead): where a mask leaves such a key visible it still counts in the softmax -- logit 0,
            // value 0 --, which is added in closed form below. The tiles hold the cnt live entities and the FIRST 16 NT - cnt dead ones (in
            // entity order): dropped = the dead-key word without its 16 NT - cnt lowest bits (scalar code; nothing is carried from row to
            // row for it). The learner's row lists mark a key dead only when every mask hides it: nothing is ever added there.
            mw_t dropped = 0;
            if (COMPACT && NT < NJT) {
                unsigned long long dk = crow.kdw & ne_bits;
                for (int i = 16 * NT - cmap.cnt; i > 0 && dk; --i) dk &= dk - 1;
                dropped = (mw_t)dk;
            }
            f32x4 stt[NT][NA];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                for (int at = 0; at < NA; ++at) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) acc = MFMA16F(Kt[ct][jt][reg], Qt[ct][at][reg], acc);
                    stt[jt][at] = acc;
                }
            f32x4 osum[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) osum[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const bool on = v < n.nvar;
                // rows of the agents that are alive as queries; the others (inactive agents) read as exact zeros: the layer's post-mask
                // (attention.py:66-67) for the callers that ask for it (zero_dead), a defined value for the rest (refil_attn_desc: discarded)
                const rsrc_t ro = mk_rsrc(n.O + v * p.sO + (long)r * p.na * p.ldo, on && !n.sum_agents ? (long)p.na * p.ldo * 4 : 0);
#pragma unroll
                for (int at = 0; at < NAT; ++at) {
                    constexpr int ZA = NA - 1;      // (index clamp for the agent tiles a short row does not have: no code is generated for them)
                    const int ax = at < NA ? at : ZA;
                    f32x4 o[NCT];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (on && at < NA) {
                        // masked softmax over the keys of this lane's agent: bit (entity at the tile position) of the agent's mask word
                        f32x4 pt[NT];
                        float mx = -INFINITY;
#pragma unroll
                        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) {
                                const bool masked = (cw[v][ax] >> ((cmap.evp[jt] >> (8 * reg)) & (NJT > 2 ? 63 : 31))) & (mw_t)1;
                                const float x = masked ? -INFINITY : stt[jt][ax][reg] * inv_scale;      // attention.py:54-57
                                pt[jt][reg] = x;
                                mx = fmaxf(mx, x);
                            }
                        mx = q_cross4_max(mx);
                        // visible zero rows outside the tiles: logit 0 each
                        const int nd = NJT > 2 ? __popcll(~cw[v][ax] & dropped) : __popc((unsigned)(~cw[v][ax] & dropped));
                        if (nd) mx = fmaxf(mx, 0.f);
                        float sum = 0.f;
#pragma unroll
                        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) {
                                const float e = pt[jt][reg] == -INFINITY ? 0.f : __expf(pt[jt][reg] - mx);
                                pt[jt][reg] = e;
                                sum += e;
                            }
                        sum = q_cross4_sum(sum);
                        if (nd) sum += (float)nd * __expf(0.f - mx);
                        const float inv = sum > 0.f ? __builtin_amdgcn_rcpf(sum) : 0.f;     // fully masked row -> 0 (attention.py:60 NaN -> 0)
#pragma unroll
                        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) pt[jt][reg] *= inv;
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                                for (int reg = 0; reg < 4; ++reg) o[ct] = MFMA16F(Vv[jt][ct][reg], pt[jt][reg], o[ct]);
                            if (n.sum_agents) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) osum[ct][e] += q_group16_sum(o[ct][e]);
                            }
                        }
                    }
                    const int idx = 16 * at + l15;
                    const bool zrow = idx < p.na && ((crow.qdw >> idx) & 1ull);
                    const bool orow = at < NA && agent_ok[ax];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        buf_st4(ro, orow ? (cmap.ek[ax] * p.ldo + col0 + 16 * ct + 4 * q) * 4 : BUF_OOB, o[ct]);
                        buf_st4(ro, zrow ? (idx * p.ldo + col0 + 16 * ct + 4 * q) * 4 : BUF_OOB, f32x4{0.f, 0.f, 0.f, 0.f});
                    }
                }
            }
            {
                const rsrc_t ro = mk_rsrc(n.O + (long)r * p.ldo, n.sum_agents ? (long)p.ldo * 4 : 0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) buf_st4(ro, l15 == 0 ? (col0 + 16 * ct + 4 * q) * 4 : BUF_OOB, osum[ct]);
            }
            QKV_TICK(t_core)
        };
        if (COMPACT && cmap.cnt <= 16) job(std::integral_constant<int, 1>{});
        else if (NJT > 2 && cmap.cnt <= 32) job(std::integral_constant<int, (NJT > 2 ? 2 : NJT)>{});
        else job(std::integral_constant<int, NJT>{});
    }
#ifdef REFIL_QKV_TIMING
    if (lane == 0) {
        const int wi = blockIdx.x * QKV_WAVES + wave;
        if (wi < 4096) {
            unsigned long long* o = g_qkv_dbg + 8 * wi;
            o[0] = t_top; o[1] = t_proj; o[2] = t_st; o[3] = t_core; o[4] = njobs; o[5] = __builtin_readcyclecounter() - t_begin; o[6] = t_begin - t_kernel0; o[7] = 0;
        }
    }
#endif
}

constexpr int QKV_MAX_DEV = 16;
static int qkv_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= QKV_MAX_DEV) { (void)hipGetLastError(); dev = 0; }
    return dev;
}
// compute units of the CURRENT device (a process may drive several): asked once per device
static int qkv_device_cus() {
    static std::atomic<int> cus[QKV_MAX_DEV];
    const int dev = qkv_device();
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// LDS a launch may ask for: 160 KB (one workgroup per CU). refil_set_tuning("qkv_lds_budget", bytes) lowers it -- the tests' way to reach
// the learner's fall-back to the separate projection / attention launches without a 350 k-row batch
static std::atomic<long> g_qkv_lds_budget{160 * 1024};
void attn_qkv_set_lds_budget(long bytes) { g_qkv_lds_budget.store(bytes > 0 && bytes < 160 * 1024 ? bytes : 160 * 1024, std::memory_order_relaxed); }

static bool qkv_wide_instantiated(int nct, int nks) { return (nct == 2 && nks == 4) || (nct == 1 && nks == 2); }
bool attn_qkv_supported(int ne, int na, int heads, int hd) {
    const int w = heads * hd;
    if (!(ne >= 1 && ne <= 48 && na >= 1 && na <= 32 && na <= ne && (hd == 16 || hd == 32) && (w == 64 || w == 128))) return false;
    const int njt = (ne + 15) / 16, nat = (na + 15) / 16;
    if (nat > njt) return false;
    // (<= 32 entities / <= 16 agents: every head dim / width combination; the wide forms: head dim 32 at width 128, 16 at 64)
    return (njt <= 2 && nat == 1) || qkv_wide_instantiated(hd / 16, w / 32);
}
// more than two key tiles or a second agent tile: instantiated and tested, but never timed on a GPU -- the learner takes these only when asked
bool attn_qkv_wide(int ne, int na) { return ne > 32 || na > 16; }

// workgroup grid of a launch over `nslices` (net, head) slices and R rows, and its LDS bytes: the W planes, the live-step prefix sums
// [B + 2], the workgroup's row table [waves][jobs per wave]
static size_t qkv_grid_lds(int nslices, long R, int T1, size_t plane_bytes, int waves, int& xcd_groups, int& ngroups) {
    const int cus = qkv_device_cus();
    if (cus % 8 == 0 && (cus / 8) % nslices == 0) { xcd_groups = cus / 8 / nslices; ngroups = 8 * xcd_groups; }
    else { xcd_groups = 0; ngroups = cus / nslices > 0 ? cus / nslices : 1; }
    const long live_groups = (R + waves - 1) / waves;
    if (!xcd_groups && ngroups > live_groups) ngroups = (int)live_groups;
    const long wstride = (long)ngroups * waves, maxjobs = (R + wstride - 1) / wstride;
    return plane_bytes + ((size_t)(R / T1) + 2 + waves * maxjobs) * 4;
}
// does a launch of `nnets` blocks over R rows fit the LDS (the row table grows with R / workgroups per slice)? The learner asks before it
// decides for the fused launch; beyond it the separate projection / attention launches run. Same sizing function, same device, same
// budget as the launcher: the two cannot disagree.
bool attn_qkv_fits(int ne, int na, int heads, int hd, long R, int T1, int nnets) {
    int xg = 0, ng = 0;
    const size_t planes = 3 * qkv_plane_bytes(hd / 16, heads * hd / 32);
    return T1 > 0 && nnets >= 1 && nnets <= QKV_MAX_NETS &&
           qkv_grid_lds(nnets * heads, R, T1, planes, qkv_waves((ne + 15) / 16, (na + 15) / 16), xg, ng) <= (size_t)g_qkv_lds_budget.load(std::memory_order_relaxed);
}

template <int NJT, int NCT, int NKS, int NAT>
static int qkv_launch_x(QkvM& k, hipStream_t st) {
    bool store = false;
    for (int i = 0; i < k.nnets; ++i) store |= k.net[i].Ko != nullptr || k.net[i].Vo != nullptr || k.net[i].Qo != nullptr;
    void (*kern)(QkvM) = store ? attn_qkv_fwd<NJT, NCT, NKS, true, NAT> : attn_qkv_fwd<NJT, NCT, NKS, false, NAT>;
    // (per device: the attribute belongs to the device's copy of the function; a repeated call is harmless, a missed one is not)
    static std::atomic<bool> raised[QKV_MAX_DEV][2];
    const int dev = qkv_device();
    if (!raised[dev][store].load(std::memory_order_relaxed)) {
        REFIL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised[dev][store].store(true, std::memory_order_relaxed);
    }
    // one workgroup per CU; workgroup -> (slice, row group). XCD-aware: the heads (and nets) that read the same x rows sit on one
    // XCD (workgroups are dealt round-robin to the 8 XCDs by block id), so a row comes from HBM once and from that XCD's L2 afterwards
    k.nslices = k.nnets * k.heads;
    static const int stagger_env = [] { const char* e = getenv("REFIL_QKV_STAGGER"); return e ? atoi(e) : 0; }();
    k.stagger = stagger_env;
    const size_t lds = qkv_grid_lds(k.nslices, k.R, k.T1, 3 * qkv_plane_bytes(NCT, NKS), qkv_waves(NJT, NAT), k.xcd_groups, k.ngroups);
    if (lds > (size_t)g_qkv_lds_budget.load(std::memory_order_relaxed)) return -1;
    hipLaunchKernelGGL(kern, dim3(k.nslices * k.ngroups), dim3(64 * qkv_waves(NJT, NAT)), lds, st, k);
    REFIL_LAUNCH_CHECK();
    return 0;
}

// n attention blocks in one launch (same rows, masks, widths and leading dimensions: attn_mfma_launch_multi's rules); src[i] = the
// layer input / in_trans weight / optional projection stores of block i. Needs precomputed mask words. -1: shape not instantiated.
int attn_qkv_launch_multi(const refil_attn_desc* descs, const AttnNetOpts* opts, const AttnQkvSrc* src, int n, int ldx, hipStream_t st,
                          float* nact, int zero_dead) {
    REFIL_CHECK(descs && opts && src && n >= 1 && n <= QKV_MAX_NETS, "refil_attn_qkv: 1..%d nets per launch", QKV_MAX_NETS);
    const refil_attn_desc& d = descs[0];
    if (!attn_qkv_supported(d.ne, d.na, d.heads, d.hd)) return -1;
    REFIL_CHECK(d.mask_words && d.row_bits, "refil_attn_qkv: precomputed mask words required (refil_attn_mask_words)");
    REFIL_CHECK(d.T1 > 0 && d.R % d.T1 == 0, "refil_attn_qkv: R must be a multiple of T1");
    const int w = d.heads * d.hd;
    REFIL_CHECK(ldx % 4 == 0 && d.ldq % 4 == 0 && d.ldkv % 4 == 0 && d.ldo % 4 == 0, "refil_attn_qkv: leading dimensions must be multiples of 4");
    QkvM k;
    memset(&k, 0, sizeof(k));
    k.nnets = n;
    for (int i = 0; i < n; ++i) {
        const refil_attn_desc& e = descs[i];
        REFIL_CHECK(e.R == d.R && e.T1 == d.T1 && e.ne == d.ne && e.na == d.na && e.heads == d.heads && e.hd == d.hd && e.ldq == d.ldq &&
                    e.ldkv == d.ldkv && e.ldo == d.ldo && e.sO == d.sO && e.t_last == d.t_last && e.mask_words == d.mask_words &&
                    e.row_bits == d.row_bits, "refil_attn_qkv: nets of one launch must share rows, masks and strides");
        REFIL_CHECK(i == 0 || e.nvar == 1, "refil_attn_qkv: nets after the first are single-variant under variant 0");
        REFIL_CHECK(!opts[i].sum_agents || e.nvar == 1, "refil_attn_qkv: the agent-sum output is a single-variant option");
        REFIL_CHECK(src[i].X && src[i].W && e.O, "refil_attn_qkv: null X / W / O");
        REFIL_CHECK((reinterpret_cast<uintptr_t>(src[i].X) & 15) == 0 && (reinterpret_cast<uintptr_t>(src[i].W) & 15) == 0, "refil_attn_qkv: X / W must be 16-byte aligned");
        QkvNet& t = k.net[i];
        t.X = src[i].X; t.W = src[i].W; t.O = e.O; t.Qo = src[i].Qo; t.Ko = src[i].Ko; t.Vo = src[i].Vo;
        t.nvar = e.nvar; t.sum_agents = opts[i].sum_agents;
    }
    k.ldx = ldx; k.w = w; k.ldq = d.ldq; k.ldkv = d.ldkv; k.ldo = d.ldo; k.sO = d.sO;
    k.R = d.R; k.T1 = d.T1; k.ne = d.ne; k.na = d.na; k.heads = d.heads; k.hd = d.hd; k.nvar = d.nvar;
    k.t_last = d.t_last;
    k.mwords = reinterpret_cast<const unsigned long long*>(d.mask_words); k.rbits = reinterpret_cast<const unsigned long long*>(d.row_bits);
    k.mw_nvar = d.mask_words_nvar > 0 ? d.mask_words_nvar : d.nvar;
    REFIL_CHECK(k.mw_nvar >= d.nvar, "refil_attn_qkv: mask_words holds fewer variants than nvar");
    k.nact = nact; k.zero_dead = zero_dead;
    // algorithmic work of the launch (all rows; the callers scale by the live fraction): in_trans on the bf16 pipe (x 6), the core on the
    // fp32 matrix instruction; bytes: the layer input read once, the outputs (and the kept projections) written once
    double flops = 0.0, bytes = 0.0, fsplit = 0.0;
    for (int i = 0; i < n; ++i) {
        const double proj = (double)d.R * 2.0 * (2.0 * d.ne + d.na) * w * w;
        fsplit += proj;
        flops += proj + (double)d.R * (2.0 + 2.0 * descs[i].nvar) * d.heads * d.na * d.ne * d.hd;
        bytes += 4.0 * d.R * ((double)d.ne * w + (double)descs[i].nvar * d.na * w + (src[i].Ko ? (2.0 * d.ne + d.na) * w : 0.0));
    }
    ProfScope prof("attn_qkv_fwd", flops, bytes, st, nullptr, 0.0, fsplit);
    const int njt = (d.ne + 15) / 16, nct = d.hd / 16, nks = w / 32, nat = (d.na + 15) / 16;
#define CASE(J, C, S, A) if (njt == J && nct == C && nks == S && nat == A) return qkv_launch_x<J, C, S, A>(k, st)
    CASE(2, 2, 4, 1); CASE(1, 2, 4, 1); CASE(2, 1, 2, 1); CASE(1, 1, 2, 1); CASE(2, 2, 2, 1); CASE(1, 2, 2, 1); CASE(2, 1, 4, 1); CASE(1, 1, 4, 1);
    // > 32 entities / > 16 agents (BASELINE configs[4]: 48 entities, 24 agents)
    CASE(3, 2, 4, 2); CASE(3, 2, 4, 1); CASE(2, 2, 4, 2); CASE(3, 1, 2, 2); CASE(3, 1, 2, 1); CASE(2, 1, 2, 2);
#undef CASE
    return -1;
}

}  // namespace refil

using namespace refil;

#ifdef REFIL_QKV_TIMING
extern "C" int refil_debug_qkv_timing(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(refil::g_qkv_dbg), (size_t)n * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int refil_attn_qkv_forward(const refil_attn_qkv_desc* q, void* stream) {
    REFIL_CHECK(q && q->X && q->W_in, "refil_attn_qkv_forward: null desc / X / W_in");
    const refil_attn_desc& d = q->attn;
    REFIL_CHECK(d.nvar >= 1 && d.nvar <= 3, "refil_attn_qkv_forward: nvar must be 1..3");
    AttnNetOpts o{0, 0};
    AttnQkvSrc s{q->X, q->W_in, q->q_out, q->k_out, q->v_out};
    const int rc = attn_qkv_launch_multi(&d, &o, &s, 1, q->ldx, (hipStream_t)stream, nullptr, 0);
    REFIL_CHECK(rc >= 0, "refil_attn_qkv_forward: shape not instantiated (n_entities <= 48, n_agents <= 32, head dim 16 / 32, width 64 / 128; more than "
                         "32 entities or 16 agents: head dim 32 at width 128 or 16 at 64) or the launch's row table exceeds the LDS");
    return rc;
}
