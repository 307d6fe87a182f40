// The REFIL learner step as an explicit forward + backward schedule of the gfx950 kernels
// (no autograd graph). Restates QLearner.train up to loss.backward()
// (reference: src/learners/q_learner.py:66-176) and the module forwards below it:
//   EntityMAC._build_inputs            src/controllers/entity_controller.py:11-30
//   ImagineEntityAttentionRNNAgent     src/modules/agents/entity_rnn_agent.py:31-64,87-126
//   EntityAttentionLayer               src/modules/layers/attention.py:24-79
//   AttentionHyperNet / FlexQMixer     src/modules/mixers/flex_qmix.py:40-57,79-121
//
// Work the reference repeats is done once: fc1/K/V are shared by the three "imagine" copies (the
// reference triples the batch, entity_rnn_agent.py:119-124), the entity||last-action tensor is
// built once for MAC and mixer (entity_controller.py:13-27 and q_learner.py:50-60 build it twice),
// the four hypernet fc1 layers are ONE [E -> 4*hyp] GEMM, and hyper_b_1 / hyper_w_final / V are
// evaluated once for the real and the imagined mix (q_learner.py:134-150 evaluates them twice).
//
// Activations live in a caller-provided workspace arena (sized for 288 GB HBM parts: nothing is
// recomputed except the attention softmax).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <utility>

#include "kernels.h"

namespace refil {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static inline long rup(long x, long m) { return (x + m - 1) / m * m; }

static int check_dims(const refil_dims& d) {
    REFIL_CHECK(d.B > 0 && d.T1 > 0, "refil: B and T1 must be > 0 (B=%d T1=%d)", d.B, d.T1);
    REFIL_CHECK(d.ne >= 1 && d.ne <= 64 && d.na >= 1 && d.na <= d.ne, "refil: need 1 <= n_agents <= n_entities <= 64");
    REFIL_CHECK(d.ed > 0 && d.A > 0, "refil: entity_shape and n_actions must be > 0");
    REFIL_CHECK(d.heads > 0 && d.d % d.heads == 0 && d.hyp % d.heads == 0, "refil: embed dims must be divisible by attn_n_heads");
    REFIL_CHECK((d.d / d.heads) % 4 == 0 && (d.hyp / d.heads) % 4 == 0, "refil: head dim must be a multiple of 4");
    REFIL_CHECK(d.agent_ff || d.H == 32 || d.H == 64 || d.H == 128, "refil: rnn_hidden_dim must be 32, 64 or 128 (got %d)", d.H);
    REFIL_CHECK(!d.mixer_lin || 2 * d.na <= 64, "refil: LinearFlexQMixer supports n_agents <= 32");
    REFIL_CHECK(d.M >= 1 && d.M <= 64, "refil: mixing_embed_dim must be in [1,64]");
    REFIL_CHECK(d.pooling >= 0 && d.pooling <= 2, "refil: pooling must be 0 (attention), 1 (mean) or 2 (max)");
    REFIL_CHECK(!d.mixer_none || !d.imagine, "refil: mixer=None cannot train an imagine agent (caq_imagine [B,T,2*n_agents] vs targets [B,T,n_agents], q_learner.py:96,169)");
    REFIL_CHECK(!d.mixer_none || (!d.mixer_lin && !d.mixer_vdn), "refil: mixer_none excludes mixer_lin / mixer_vdn");
    return 0;
}

static inline int in_dim(const refil_dims& d) { return d.ed + (d.entity_last_action ? d.A : 0); }

// ------------------------------------------------------------------------------------------------
// flat parameter layout
// ------------------------------------------------------------------------------------------------
static void param_layout(const refil_dims& d, refil_param_layout& L) {
    const long E = in_dim(d), dd = d.d, H = d.H, A = d.A, h = d.hyp, M = d.M;
    long o = 0;
    auto take = [&](long n) { long r = o; o = rup(o + n, 4); return r; };
    L.ag_fc1_w = take(dd * E); L.ag_fc1_b = take(dd);
    L.ag_in_w = take(3 * dd * dd);
    L.ag_out_w = take(dd * dd); L.ag_out_b = take(dd);
    if (d.agent_ff) {      // EntityAttentionFFAgent: fc2 maps attn_embed_dim -> n_actions, no GRU / fc3
        L.ag_fc2_w = take(A * dd); L.ag_fc2_b = take(A);
        L.ag_w_ih = L.ag_w_hh = L.ag_b_ih = L.ag_b_hh = L.ag_fc3_w = L.ag_fc3_b = o;
    } else {
        L.ag_fc2_w = take(H * dd); L.ag_fc2_b = take(H);
        L.ag_w_ih = take(3 * H * H); L.ag_w_hh = take(3 * H * H);
        L.ag_b_ih = take(3 * H); L.ag_b_hh = take(3 * H);
        L.ag_fc3_w = take(A * H); L.ag_fc3_b = take(A);
    }
    L.agent_total = o;
    const long nn = (d.mixer_vdn || d.mixer_none) ? 0 : (d.mixer_lin ? 2 : 4);       // hypernets: none (VDN), (hyper_w_1, V) or (hyper_w_1, hyper_w_final, hyper_b_1, V)
    L.mix_fc1_w_stride = h * E; L.mix_fc1_w = take(nn * h * E);
    L.mix_fc1_b_stride = h; L.mix_fc1_b = take(nn * h);
    L.mix_in_w_stride = 3 * h * h; L.mix_in_w = take(nn * 3 * h * h);
    L.mix_out_w_stride = h * h; L.mix_out_w = take(nn * h * h);
    L.mix_out_b_stride = h; L.mix_out_b = take(nn * h);
    L.mix_fc2_w_stride = M * h; L.mix_fc2_w = take(nn * M * h);
    L.mix_fc2_b_stride = M; L.mix_fc2_b = take(nn * M);
    L.total = o;
}

// ------------------------------------------------------------------------------------------------
// workspace arena
// ------------------------------------------------------------------------------------------------
struct Arena {
    char* base; size_t cap; size_t off; bool overflow;
    template <typename T> T* take(long n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += (size_t)n * sizeof(T);
        if (base && off > cap) overflow = true;
        return p;
    }
};

constexpr long PARTIAL_FLOATS = 40L << 20;   // split-K scratch (160 MiB)
// partials of the step's deferred split reductions: sized from the parameter count (measured 53-64 x the parameter floats across the
// BASELINE configs: 28-119 MiB; twice that + a floor; a launch that does not fit reduces at once, gemm_launch_dw)
static long dpool_floats(const refil_dims& d) { refil_param_layout L; param_layout(d, L); return 128L * L.total + (4L << 20); }

struct AgentBufs {     // one entity-attention recurrent agent evaluation (G mask variants)
    float *x1, *kv, *q, *ao, *x2, *x3, *gi, *hsx, *sr, *sz, *sn, *sg, *qv;
    float *wc, *bc, *bd;   // composed out_trans o fc2 of the recurrent agent: W_2 W_o [H,d], W_2 b_o + b_2, W_2 b_o (Ctx::compose_agent)
};
struct HyperBufs {     // the four attention hypernets (NV variant evaluations in total)
    float *x1, *kv, *q, *ao, *x2, *x3;
    float *wc, *bc;    // composed out_trans o fc2 maps [nets][M][h], [nets][M] (Ctx::presum)
};
struct Work {
    float* xe; uint8_t *emc, *amask, *em0;
    AgentBufs la, ta;
    HyperBufs lh, th;
    float *chosen, *tmax, *q_tot, *q_tot_im, *tq_tot, *gc_real, *gc_im, *targets, *ingroup;
    float* row_stats;  // [B*T][8] per-row terms of the stat sums (fused mixing kernel)
    float* actf;       // [R*na] 1 / 0 for active / inactive agents
    float* nact;       // [R] active agents per (b,t): weight of the bias terms of the agent-summed hypernet tails
    // backward
    float *dx3h, *dchosen, *dx2h, *daoh, *dqh, *dkvh, *dx1h;
    float *gwc, *gbc;  // gradients of the composed maps [nets][M][h], [nets][M]
    float *gwca, *gbca, *gbact;   // agent: dL/dW_c [H,d], colsum(dx3) over all rows [H], over active rows [H]
    float *dqva, *dhs, *dgi, *dgh, *dx3a, *dx2a, *daoa, *dqa, *dkva, *dx1a;
    float* partial;
    float* partial2;   // split-K scratch of the side (agent-chain) stream
    float* partial3; float* partial4;   // split-K scratch of the chains' own streams (their last weight gradients)
    float* dpool;      // partials of the weight gradients whose reductions wait for the end of the step (DeferredReduce)
    // row lists (kernels.h: ListArgs): rows that cannot influence the step are skipped
    int *t_last, *list_ea, *list_eh, *list_a, *counts, *lcnt, *loff;
    int *list_t, *list_t3, *list_h, *list_ht;      // agent-row lists of the layers behind the attention cores (kernels.h: ListArgs)
    uint8_t *kdead_a, *kdead_h, *ever;
    unsigned long long* lsync;     // [B][8] exchange area of the one-launch list kernel
    // mask words of the step, built once for all attention launches (attention_mfma.hip: attn_mask_words_kernel)
    unsigned long long *mw_a, *rb_a, *mw_h, *rb_h;
    // variant-0 words alone (observability / entity mask: no partition bits needed) for the target nets' EARLY forward
    unsigned long long *mw_at, *rb_at, *mw_ht, *rb_ht;
    // Second copy of everything the step's input-only prologue writes (input assembly, row lists, mask words): a step whose
    // batch fields are ready early (refil_batch.ready_event) builds them beside the end of the previous step, which still reads its own.
    struct Early {
        float* xe; uint8_t *emc, *amask, *em0; float* actf; float* nact;
        int *t_last, *list_ea, *list_eh, *list_a, *counts, *lcnt, *loff, *list_t, *list_t3, *list_h, *list_ht;
        uint8_t *kdead_a, *kdead_h, *ever;
        unsigned long long *lsync, *mw_a, *rb_a, *mw_h, *rb_h, *mw_at, *rb_at, *mw_ht, *rb_ht;
    } alt;
};
static void use_alt_slot(Work& w) {
    const Work::Early& e = w.alt;
    w.xe = e.xe; w.emc = e.emc; w.amask = e.amask; w.em0 = e.em0; w.actf = e.actf; w.nact = e.nact;
    w.t_last = e.t_last; w.list_ea = e.list_ea; w.list_eh = e.list_eh; w.list_a = e.list_a; w.counts = e.counts; w.lcnt = e.lcnt;
    w.loff = e.loff; w.list_t = e.list_t; w.list_t3 = e.list_t3; w.list_h = e.list_h; w.list_ht = e.list_ht;
    w.kdead_a = e.kdead_a; w.kdead_h = e.kdead_h; w.ever = e.ever;
    w.lsync = e.lsync; w.mw_a = e.mw_a; w.rb_a = e.rb_a; w.mw_h = e.mw_h; w.rb_h = e.rb_h;
    w.mw_at = e.mw_at; w.rb_at = e.rb_at; w.mw_ht = e.mw_ht; w.rb_ht = e.rb_ht;
}

struct Sizes {
    long R, NE, NA; int G, nv0, NV, E, Ep, nets;
    long NEa, NAa;     // allocated rows of the entity-row / agent-row buffers the row-list GEMMs touch: the rows past
                       // the logical end are the scratch rows the padded list entries point at
};
static Sizes sizes_of(const refil_dims& d) {
    Sizes s;
    s.R = (long)d.B * d.T1; s.NE = s.R * d.ne; s.NA = s.R * d.na;
    s.G = d.imagine ? 3 : 1; s.nv0 = s.G; s.nets = (d.mixer_vdn || d.mixer_none) ? 0 : (d.mixer_lin ? 2 : 4); s.NV = s.nets ? s.nv0 + s.nets - 1 : 0;
    s.E = in_dim(d); s.Ep = (int)rup(s.E, 4);
    s.NEa = s.NE + 8; s.NAa = s.NA + 8;
    return s;
}

static void carve_agent(Arena& a, const refil_dims& d, const Sizes& s, int G, bool save, AgentBufs& b) {
    b.x1 = a.take<float>(s.NEa * d.d);
    b.kv = a.take<float>(s.NEa * 2 * d.d);
    b.q = a.take<float>(s.NAa * d.d);
    b.ao = a.take<float>(((long)G * s.NA + 8) * d.d);            // (+8: scratch rows behind the agent-row lists' padding)
    b.x2 = a.take<float>((long)G * s.NA * d.d);
    b.x3 = a.take<float>(((long)G * s.NA + 8) * d.H);
    b.gi = a.take<float>(((long)G * s.NA + 8) * 3 * d.H);
    b.hsx = a.take<float>((long)G * d.B * (d.T1 + 1) * d.na * d.H);
    b.sr = b.sz = b.sn = b.sg = nullptr;
    if (save) {
        b.sr = a.take<float>((long)G * s.NA * d.H); b.sz = a.take<float>((long)G * s.NA * d.H);
        b.sn = a.take<float>((long)G * s.NA * d.H); b.sg = a.take<float>((long)G * s.NA * d.H);
    }
    b.qv = a.take<float>((long)G * s.NA * d.A);
    b.wc = a.take<float>((long)d.H * d.d); b.bc = a.take<float>(d.H); b.bd = a.take<float>(d.H);
}
static void carve_hyper(Arena& a, const refil_dims& d, const Sizes& s, int NV, HyperBufs& b) {
    b.x1 = a.take<float>(s.NEa * s.nets * d.hyp);
    b.kv = a.take<float>(s.nets * s.NEa * 2 * d.hyp);
    b.q = a.take<float>(s.nets * s.NAa * d.hyp);
    b.ao = a.take<float>(((long)NV * s.NA + 8) * d.hyp);
    b.x2 = a.take<float>((long)NV * s.NA * d.hyp);
    b.x3 = a.take<float>(((long)NV * s.NA + 8) * d.M);
    b.wc = a.take<float>((long)s.nets * d.M * d.hyp);
    b.bc = a.take<float>((long)s.nets * d.M);
}

enum CarveMode { CARVE_LEARNER, CARVE_AGENT_FWD, CARVE_MIXER_FWD };
constexpr int QKV_DEFAULT = 15;

static void carve(Arena& a, const refil_dims& d, Work& w, CarveMode mode) {
    const Sizes s = sizes_of(d);
    const long BT = (long)d.B * (d.T1 > 1 ? d.T1 - 1 : 1);
    memset(&w, 0, sizeof(w));
    w.xe = a.take<float>(s.NEa * s.Ep);
    w.emc = a.take<uint8_t>(s.NE); w.amask = a.take<uint8_t>(s.NA); w.em0 = a.take<uint8_t>((long)d.B * d.ne);
    w.nact = a.take<float>(s.R);
    w.actf = a.take<float>(s.NA);
    if (mode == CARVE_AGENT_FWD) { carve_agent(a, d, s, s.G, false, w.la); return; }
    if (mode == CARVE_MIXER_FWD) {
        carve_hyper(a, d, s, s.NV, w.lh);
        w.ingroup = a.take<float>(BT + d.B);
        w.chosen = a.take<float>(3 * BT * d.na);          // de-interleaved agent Qs
        return;
    }
    carve_agent(a, d, s, s.G, true, w.la);
    carve_agent(a, d, s, 1, false, w.ta);
    carve_hyper(a, d, s, s.NV, w.lh);
    carve_hyper(a, d, s, s.nets, w.th);
    w.chosen = a.take<float>((long)s.G * BT * d.na);
    w.tmax = a.take<float>(BT * d.na);
    w.q_tot = a.take<float>(BT); w.q_tot_im = a.take<float>(BT); w.tq_tot = a.take<float>(BT);
    w.gc_real = a.take<float>(BT); w.gc_im = a.take<float>(BT); w.targets = a.take<float>(BT * (d.mixer_none ? d.na : 1)); w.ingroup = a.take<float>(BT);
    w.row_stats = a.take<float>(BT * 8);
    w.dx3h = a.take<float>(((long)s.NV * s.NA + 8) * d.M);
    w.dchosen = a.take<float>((long)s.G * BT * d.na);
    w.dx2h = a.take<float>((long)s.NV * s.NA * d.hyp);
    w.daoh = a.take<float>(((long)s.NV * s.NA + 8) * d.hyp);
    w.dqh = a.take<float>(s.nets * s.NAa * d.hyp);
    w.dkvh = a.take<float>(s.nets * s.NEa * 2 * d.hyp);
    w.dx1h = a.take<float>(s.NEa * s.nets * d.hyp);
    w.gwc = a.take<float>((long)s.nets * d.M * d.hyp);
    w.gbc = a.take<float>((long)s.nets * d.M);
    w.gwca = a.take<float>((long)d.H * d.d); w.gbca = a.take<float>(d.H); w.gbact = a.take<float>(d.H);
    w.dqva = a.take<float>((long)s.G * s.NA * d.A);
    w.dhs = a.take<float>((long)s.G * s.NA * d.H);
    w.dgi = a.take<float>(((long)s.G * s.NA + 8) * 3 * d.H);
    w.dgh = a.take<float>((long)s.G * s.NA * d.H);          // n block of d(gh) only (gru.hip: the r / z blocks are dgi's)
    w.dx3a = a.take<float>(((long)s.G * s.NA + 8) * d.H);
    w.dx2a = a.take<float>((long)s.G * s.NA * d.d);
    w.daoa = a.take<float>(((long)s.G * s.NA + 8) * d.d);
    w.dqa = a.take<float>(s.NAa * d.d);
    w.dkva = a.take<float>(s.NEa * 2 * d.d);
    w.dx1a = a.take<float>(s.NEa * d.d);
    w.partial = a.take<float>(PARTIAL_FLOATS);
    w.partial2 = a.take<float>(PARTIAL_FLOATS);
    w.partial3 = a.take<float>(PARTIAL_FLOATS);
    w.partial4 = a.take<float>(PARTIAL_FLOATS);
    w.dpool = a.take<float>(dpool_floats(d));
    w.t_last = a.take<int>(d.B);
    w.list_ea = a.take<int>(s.NE + 256); w.list_eh = a.take<int>(s.NE + 256); w.list_a = a.take<int>(s.NA + 256);
    w.counts = a.take<int>(8); w.lcnt = a.take<int>(4 * s.R); w.loff = a.take<int>(4 * (s.R + 1));
    w.list_t = a.take<int>(s.NA + 256); w.list_t3 = a.take<int>((long)s.G * s.NA + 256);
    w.list_h = a.take<int>((long)s.nv0 * s.NA + 256); w.list_ht = a.take<int>(s.NA + 256);
    w.kdead_a = a.take<uint8_t>(s.NE); w.kdead_h = a.take<uint8_t>(s.NE); w.ever = a.take<uint8_t>((long)d.B * d.na);
    w.lsync = a.take<unsigned long long>(256 * 8 + (long)d.B * 8);
    {
        const long na_pad = (d.na + 15) / 16 * 16;
        w.mw_a = a.take<unsigned long long>(s.R * 3 * na_pad); w.rb_a = a.take<unsigned long long>(s.R * 3);
        w.mw_h = a.take<unsigned long long>(s.R * 3 * na_pad); w.rb_h = a.take<unsigned long long>(s.R * 3);
        w.mw_at = a.take<unsigned long long>(s.R * na_pad); w.rb_at = a.take<unsigned long long>(s.R * 3);
        w.mw_ht = a.take<unsigned long long>(s.R * na_pad); w.rb_ht = a.take<unsigned long long>(s.R * 3);
        Work::Early& e = w.alt;
        e.mw_a = a.take<unsigned long long>(s.R * 3 * na_pad); e.rb_a = a.take<unsigned long long>(s.R * 3);
        e.mw_h = a.take<unsigned long long>(s.R * 3 * na_pad); e.rb_h = a.take<unsigned long long>(s.R * 3);
        e.mw_at = a.take<unsigned long long>(s.R * na_pad); e.rb_at = a.take<unsigned long long>(s.R * 3);
        e.mw_ht = a.take<unsigned long long>(s.R * na_pad); e.rb_ht = a.take<unsigned long long>(s.R * 3);
        e.t_last = a.take<int>(d.B);
        e.xe = a.take<float>(s.NEa * s.Ep);
        e.emc = a.take<uint8_t>(s.NE); e.amask = a.take<uint8_t>(s.NA); e.em0 = a.take<uint8_t>((long)d.B * d.ne);
        e.actf = a.take<float>(s.NA); e.nact = a.take<float>(s.R);
        e.list_ea = a.take<int>(s.NE + 256); e.list_eh = a.take<int>(s.NE + 256); e.list_a = a.take<int>(s.NA + 256);
        e.counts = a.take<int>(8); e.lcnt = a.take<int>(4 * s.R); e.loff = a.take<int>(4 * (s.R + 1));
        e.list_t = a.take<int>(s.NA + 256); e.list_t3 = a.take<int>((long)s.G * s.NA + 256);
        e.list_h = a.take<int>((long)s.nv0 * s.NA + 256); e.list_ht = a.take<int>(s.NA + 256);
        e.kdead_a = a.take<uint8_t>(s.NE); e.kdead_h = a.take<uint8_t>(s.NE); e.ever = a.take<uint8_t>((long)d.B * d.na);
        e.lsync = a.take<unsigned long long>(256 * 8 + (long)d.B * 8);
    }
}

static size_t workspace_bytes(const refil_dims& d, CarveMode mode) {
    Arena a{nullptr, 0, 0, false};
    Work w;
    carve(a, d, w, mode);
    return a.off + 256;
}

// ------------------------------------------------------------------------------------------------
// GEMM helpers
// ------------------------------------------------------------------------------------------------
static refil_gemm_desc G0() {
    refil_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.batch = 1; g.splits = 1;
    return g;
}
// y[M,N] = x[M,K] W[N,K]^T (+bias)
static refil_gemm_desc linear(const float* x, int ldx, const float* W, int ldw, const float* bias, float* y, int ldy,
                              long M, int N, int K, int flags) {
    refil_gemm_desc g = G0();
    g.A = x; g.lda = ldx; g.B = W; g.ldb = ldw; g.bias = bias; g.C = y; g.ldc = ldy;
    g.M = (int)M; g.N = N; g.K = K; g.flags = flags;
    return g;
}
// dx[M,K] = dy[M,N] W[N,K]
static refil_gemm_desc linear_dx(const float* dy, int lddy, const float* W, int ldw, float* dx, int lddx, long M, int N,
                                 int K, int flags) {
    refil_gemm_desc g = G0();
    g.A = dy; g.lda = lddy; g.B = W; g.ldb = ldw; g.C = dx; g.ldc = lddx;
    g.M = (int)M; g.N = K; g.K = N; g.flags = flags | REFIL_GEMM_B_OUTC;
    return g;
}
// reduction splits of a weight-gradient launch (g.M x g.N outputs, g.K rows, g.batch nets). The count follows the kernel that will take the
// launch (4x4-tile / bf16 x 6 / streamed), and which one does depends on the row maps and the row list the caller attaches AFTER
// linear_dw(): the rule runs once on the bare descriptor and again on the final one (gemm_launch_dw) -- same rule, so a descriptor whose
// kernel did not change keeps its count, and one whose maps turned a kernel away gets the count that kernel's sweep chose
static void dw_size_splits(refil_gemm_desc& g) {
    const int N = g.M, K = g.N, batch = g.batch;
    const long R = g.K;
    g.splits = 2;
    if (gemm_dw4_enabled() && gemm_dw4_eligible(g)) {            // gemm_dw4.hip: one workgroup per CU
        g.splits = gemm_dw4_splits(g);
        while (g.splits > 2 && (long)batch * g.splits * ((long)N * K + N) > PARTIAL_FLOATS) --g.splits;
        return;
    }
    const int bn = K > 64 ? 128 : (K > 32 ? 64 : 32);
    const long tiles = (long)cdiv(N, 128) * cdiv(K, bn) * batch;
    // workgroups a streamed weight-gradient launch aims for. Swept on one box with the step's four streams running
    // (tools/sweep.sh): 512 beats 1024 and 256 -- these launches share the GPU with three other streams, and fewer splits
    // mean less partial traffic and a shorter reduction
    static const long target = [] { const char* e = getenv("REFIL_DW_TARGET"); return e ? atol(e) : 512L; }();
    long splits = (g_tuning.dw_target > 0 ? g_tuning.dw_target : target) / tiles;
    splits = min(splits, cdivl(R, N <= 64 ? 128 : 256));      // thin layers run narrow tiles: more, shorter splits
    splits = max(splits, 1L);
    while (splits > 1 && (long)batch * splits * ((long)N * K + N) > PARTIAL_FLOATS) --splits;
    g.splits = (int)splits;
}

// dW[N,K] = dy[R,N]^T x[R,K]; db[N] = colsum(dy)  (reduction over the R rows, split deterministically)
static refil_gemm_desc linear_dw(const float* dy, int lddy, const float* x, int ldx, float* dW, int lddw, float* db,
                                 long R, int N, int K, float* partial, int batch) {
    refil_gemm_desc g = G0();
    g.A = dy; g.lda = lddy; g.B = x; g.ldb = ldx; g.C = dW; g.ldc = lddw;
    g.M = N; g.N = K; g.K = (int)R;
    g.flags = REFIL_GEMM_A_OUTC | REFIL_GEMM_B_OUTC | (db ? REFIL_GEMM_COLSUM_A : 0);
    g.colsum = db; g.partial = partial; g.batch = batch;
    dw_size_splits(g);
    return g;
}

#define RUN(x) do { if (int e_ = (x)) return e_; } while (0)

// mask code of the within-group (which = 0) / between-group (which = 1) imagined copy: random split,
// ground-truth factors or their combination (dims.gt_factors), with or without the observability mask
static int group_code(const refil_dims& d, int which, bool with_obs) {
    static const int T[3][2][2] = {
        {{REFIL_MASK_WITHIN, REFIL_MASK_INTERACT}, {REFIL_MASK_OBS_WITHIN, REFIL_MASK_OBS_INTERACT}},
        {{REFIL_MASK_GTW, REFIL_MASK_GTI}, {REFIL_MASK_OBS_GTW, REFIL_MASK_OBS_GTI}},
        {{REFIL_MASK_RGTW, REFIL_MASK_RGTI}, {REFIL_MASK_OBS_RGTW, REFIL_MASK_OBS_RGTI}}};
    return T[d.gt_factors][with_obs ? 1 : 0][which];
}

// The step has two independent chains between its join points: the agent chain (3-variant live agent,
// target agent, GRUs -- latency-bound, few workgroups) and the hypernet chain (throughput-bound GEMMs).
// They run on two HIP streams (fork/join with events, graph-capturable) so that the persistent GRU's 96
// workgroups do not leave the other 160 CUs idle. REFIL_NO_OVERLAP=1 serialises everything on one stream.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // Weight-gradient streams: nothing but the optimiser consumes dW / db, so the dW GEMMs (and their split
    // reductions) of each chain are forked onto their own stream as soon as their operands exist instead of sitting
    // in the chain's launch order in front of the dX GEMMs that ARE on the critical path.
    hipStream_t g[2] = {nullptr, nullptr};
    hipEvent_t pool[128] = {};
    int next_ev = 0;
    bool ok = false;
};
constexpr int MAX_DEVICES = 16;
static thread_local SideStream g_side[MAX_DEVICES];       // one per device: a stream is bound to the device it was created on
static int side_stream(SideStream*& out) {
    int dev = 0;
    REFIL_HIP(hipGetDevice(&dev));
    REFIL_CHECK(dev >= 0 && dev < MAX_DEVICES, "refil: device ordinal %d out of range", dev);
    SideStream& sd = g_side[dev];
    if (!sd.ok) {
        int lo = 0, hi = 0;
        REFIL_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // Both chains are throughput-bound GEMM trains now, so the side stream runs at the default priority (measured:
        // 3.65 vs 3.69 ms/step with the highest priority). REFIL_SIDE_PRIO=1 restores the high-priority stream.
        // REFIL_SIDE_PRIO / REFIL_GRAD_PRIO: 1 = highest, -1 = lowest priority of the hypernet-chain / weight-gradient streams
        // relative to the caller's stream (which carries the agent chain, the longer dependency chain of the step)
        auto prio = [&](const char* name, int dflt) {
            const char* e = getenv(name);
            const int v = e ? atoi(e) : dflt;
            return v > 0 ? hi : (v < 0 ? lo : 0);
        };
        REFIL_HIP(hipStreamCreateWithPriority(&sd.s, hipStreamNonBlocking, prio("REFIL_SIDE_PRIO", 0)));
        for (auto& e : sd.ev) REFIL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& g : sd.g) REFIL_HIP(hipStreamCreateWithPriority(&g, hipStreamNonBlocking, prio("REFIL_GRAD_PRIO", 0)));
        for (auto& e : sd.pool) REFIL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sd.ok = true;
    }
    out = &sd;
    return 0;
}
// error path: whatever was enqueued on the side stream after a fork must still be ordered before the caller's next work
static void side_stream_rejoin(SideStream* sd, hipStream_t main) {
    if (!sd || !sd->ok) return;
    if (hipEventRecord(sd->ev[3], sd->s) == hipSuccess) (void)hipStreamWaitEvent(main, sd->ev[3], 0);
    for (int i = 0; i < 2; ++i)
        if (hipEventRecord(sd->pool[i], sd->g[i]) == hipSuccess) (void)hipStreamWaitEvent(main, sd->pool[i], 0);
    (void)hipGetLastError();
}
// `to` waits for everything enqueued on `from` so far (no-op when they are the same stream)
static int stream_after(SideStream* sd, hipStream_t from, hipStream_t to) {
    if (from == to || !sd) return 0;
    hipEvent_t e = sd->pool[sd->next_ev];
    sd->next_ev = (sd->next_ev + 1) % 128;
    REFIL_HIP(hipEventRecord(e, from));
    REFIL_HIP(hipStreamWaitEvent(to, e, 0));
    return 0;
}
// Data parallelism: the mixer's gradients (+ the stat sums) are complete long before the agent's BPTT is; this hook lets
// the caller start their all-reduce at that point, on the stream they complete on (refil_set_mixer_grads_hook).
static thread_local refil_grads_hook g_mixer_hook = nullptr;
static thread_local void* g_mixer_hook_user = nullptr;

static thread_local int64_t g_stat_steps = 0, g_stat_early = 0, g_stat_et_h = 0, g_stat_et_a = 0;      // refil_get_stat

static int g_overlap = -1;      // -1: follow the environment, 0/1: set by refil_set_overlap
static bool overlap_enabled() {
    if (g_overlap >= 0) return g_overlap != 0;
    const char* e = getenv("REFIL_NO_OVERLAP");
    return !(e && e[0] == '1');
}

// Parameter gradients are read by nobody before the optimiser, so the reductions of their split launches are collected here
// and run as ONE launch after the step's last join (reduce_multi_launch) instead of one small launch behind every weight
// gradient: each deferred launch keeps its partials in its own slice of Work::dpool until then. Same arithmetic, same order.
struct DeferredReduce {
    ReduceK r[64];
    hipStream_t on[64];        // the stream each launch ran on
    int n = 0;
    int mode = 2;              // 1: one launch after the step's last join; 2: one launch per stream, at the end of that stream
    float* pool = nullptr; long cap = 0, used = 0;
    const float* lo = nullptr; const float* hi = nullptr;      // the gradient buffer: only launches writing into it are deferred
};
struct Ctx {
    refil_dims d; Sizes s; refil_batch b; refil_param_layout L; Work w; hipStream_t st;
    // FlexQMixer's hyper_w_final / hyper_b_1 / V are consumed only through their MEAN over the agents
    // (flex_qmix.py:51-56) and out_trans / fc2 are linear, so these three nets are evaluated on the agent-SUM of the
    // attention output: [R, h] rows instead of [R*na, h] through out_trans, fc2 and their backward (16x fewer rows):
    //   S2 = W_o (sum_i a_i) + n_act b_o ,  S3 = W_2 S2 + n_act b_2 ,  mean_i x3_i = S3 / na .
    bool presum;
    // recurrent agent: x3 = relu(fc2(mask(out_trans(a)))) = relu(a W_c^T + b_2 + active * W_2 b_o) with a = 0 for inactive
    // agents (the attention kernel applies the post-mask): out_trans and its backward GEMMs disappear
    bool compose_agent;
    // Row lists: the entity-row projections (fc1, K/V, their dX / dW) and the query projections run over the rows that
    // can influence the loss only, attention and the recurrences stop after an episode's last contributing step.
    // Learner steps of the flagship family at sizes where every listed GEMM takes the weight-resident / streaming
    // kernels; REFIL_DENSE=1 switches it off (same results up to the summation order of the weight gradients).
    bool lists;
    bool mwords;       // the step's mask words are precomputed (learner steps on the matrix-core attention path)
    // weight-gradient stream of this chain (== st when the chains are serialised) and its split-reduction scratch
    hipStream_t gst; float* gpartial; SideStream* sd;
    // The chain's LAST weight gradients (bit 0: the query projection's, bit 1: fc1's) run on the chain's own stream, which has
    // nothing else left to do, beside the ones still queued on the weight-gradient stream; bit 2: the composed tail's
    // parameter gradients are enqueued behind the in_trans gradient instead of in front of it
    int tail_dw; float* tpartial;
    hipStream_t mwst;  // stream the step's mask words are built on (the chain waits for it right before its first attention launch)
    DeferredReduce* defer;
    // this workspace's previous call was a learner step of the same shape (the carve did not move) / the prologue slot of this
    // call (alternates on an unchanged layout) / the events behind the last readers of the two slots
    bool same_layout; int slot; hipEvent_t* slot_free; hipEvent_t pre_done;
    struct Prev* prev; // what the library remembers about this workspace (make_ctx)
    // the composed out_trans o fc2 maps of this call's nets are already in the workspace: built at the head of the chain (live agent)
    // or by an earlier call with the same target parameters (target nets: refil_batch.target_version) -- the forward skips them
    bool agent_composed, skip_compose;
    int mw_nvar;       // variants per row in w.mw_a / w.mw_h (the step's G; 1 for the target nets' early variant-0 words)
    bool target_same;  // params_target is what the previous call on this workspace saw (refil_batch.target_version)
    // in_trans + attention core as ONE launch (attention_qkv.hip) for: bit 0 the target hypernets, bit 1 the target agent (neither
    // stores Q / K / V: q_learner.py:111-113,154 never differentiates them), bit 2 the live hypernets, bit 3 the live agent (Q / K / V
    // stored once for the backward). REFIL_ATTN_QKV / refil_set_tuning("attn_qkv")
    int qkv;
};
enum { QKV_T_HYPER = 1, QKV_T_AGENT = 2, QKV_L_HYPER = 4, QKV_L_AGENT = 8 };

static int gemm_launch_dw(const Ctx& c, refil_gemm_desc& g, hipStream_t st) {
    dw_size_splits(g);              // (the descriptor is final here: row maps and row list attached)
    DeferredReduce* df = c.defer;
    if (df && g.splits > 1 && df->n < 64 && !(g.flags & REFIL_GEMM_ACCUM) && g.C >= df->lo && g.C < df->hi &&
        (!g.colsum || (g.colsum >= df->lo && g.colsum < df->hi))) {
        const long need = ((long)g.batch * g.splits * ((long)g.M * g.N + g.M) + 63) & ~63L;
        if (df->used + need <= df->cap) {
            g.partial = df->pool + df->used;
            ReduceK r;
            if (int e = gemm_launch(g, st, &r)) return e;
            if (r.splits > 1) { df->on[df->n] = st; df->r[df->n++] = r; df->used += need; }
            return 0;
        }
    }
    return gemm_launch(g, st);
}

// weight gradients (+ their reductions) leave the chain: forked behind everything enqueued on the chain so far
static int launch_dw(const Ctx& c, refil_gemm_desc g) {
    g.partial = c.gpartial;
    if (int e = stream_after(c.sd, c.st, c.gst)) return e;
    return gemm_launch_dw(c, g, c.gst);
}
static int launch_dw_tail(const Ctx& c, refil_gemm_desc g, int bit) {
    if (!(c.tail_dw & bit)) return launch_dw(c, g);
    g.partial = c.tpartial;
    return gemm_launch_dw(c, g, c.st);
}

// List lengths of an earlier step, copied back asynchronously into pinned host memory (one slot per device): a HINT
// for sizing the next steps' launch grids -- never waited for, never used for anything a result depends on.
static thread_local int* g_hint[MAX_DEVICES] = {};
static thread_local int* g_hint_dev[MAX_DEVICES] = {};      // the same memory as the device sees it
static int* row_hints();
static int* row_hints_dev() {
    int dev = 0;
    if (!row_hints() || hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (!g_hint_dev[dev]) {
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, g_hint[dev], 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        g_hint_dev[dev] = static_cast<int*>(d);
    }
    return g_hint_dev[dev];
}
static int* row_hints() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) { (void)hipGetLastError(); return nullptr; }
    if (!g_hint[dev]) {
        void* p = nullptr;
        // [0..7] row-count hints, [8] sticky error word of the one-launch row-list kernel (mixer.hip: lists_fused_kernel)
        if (hipHostMalloc(&p, 16 * sizeof(int), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        memset(p, 0, 16 * sizeof(int));
        g_hint[dev] = static_cast<int*>(p);
    }
    return g_hint[dev];
}

const int* lists_error_word_dev() {
    // (only once a learner step has created the words: the optimiser alone never allocates them)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES || !g_hint[dev]) { (void)hipGetLastError(); return nullptr; }
    const int* h = row_hints_dev();
    return h ? h + 8 : nullptr;
}

struct RowList { const int* idx; const int* cnt; int which; };      // which: index into the counts array (= the hint slot)
static refil_gemm_desc with_rows(refil_gemm_desc g, const Ctx& c, RowList l) {
    if (c.lists) {
        g.row_index = l.idx; g.row_count = l.cnt;
        const int* h = row_hints();
        g.row_count_hint = h ? h[l.which] : 0;
    }
    return g;
}
static RowList rows_ea(const Ctx& c) { return RowList{c.w.list_ea, c.w.counts + 0, 0}; }
static RowList rows_eh(const Ctx& c) { return RowList{c.w.list_eh, c.w.counts + 1, 1}; }
static RowList rows_a(const Ctx& c) { return RowList{c.w.list_a, c.w.counts + 2, 2}; }
// rows variant * NA + (b,t,i) of the recurrent tail of an agent evaluated under G mask variants (ListArgs::list_t)
static RowList rows_t(const Ctx& c, int G) { return G == 1 ? RowList{c.w.list_t, c.w.counts + 7, 7} : RowList{c.w.list_t3, c.w.counts + 4, 4}; }
// rows variant * NA + (b,t,i) of hyper_w_1's per-agent tail under nv0 mask variants (active agents of live steps)
static RowList rows_h(const Ctx& c, int nv0) { return nv0 == c.s.nv0 ? RowList{c.w.list_h, c.w.counts + 5, 5} : RowList{c.w.list_ht, c.w.counts + 6, 6}; }
static void attn_rows(const Ctx& c, refil_attn_desc& a, bool hyper) {
    if (c.lists) { a.t_last = c.w.t_last; a.kv_dead = hyper ? c.w.kdead_h : c.w.kdead_a; a.q_dead = c.w.amask; }
    if (c.mwords) { a.mask_words = hyper ? c.w.mw_h : c.w.mw_a; a.row_bits = hyper ? c.w.rb_h : c.w.rb_a; a.mask_words_nvar = c.mw_nvar; }
}

static refil_rowmap agent_rows(const Ctx& c) { return refil_rowmap{c.d.na, c.d.ne, 0}; }
static refil_rowmap hs_rows(const Ctx& c, int off) { return refil_rowmap{c.d.T1 * c.d.na, (c.d.T1 + 1) * c.d.na, off}; }

static refil_attn_desc attn_base(const Ctx& c, int w) {
    refil_attn_desc a;
    memset(&a, 0, sizeof(a));
    a.R = (int)c.s.R; a.T1 = c.d.T1; a.ne = c.d.ne; a.na = c.d.na; a.heads = c.d.heads; a.hd = w / c.d.heads;
    a.obs_mask = c.b.obs_mask; a.om_sB = c.b.om_sB; a.om_sT = c.b.om_sT;
    if (c.d.gt_obs_mask) { a.obs_mask = c.b.gt_mask; a.om_sB = c.b.gt_sB; a.om_sT = c.b.gt_sT; }   // entity_ff_agent.py:34-35
    a.gt_mask = c.b.gt_mask; a.gt_sB = c.b.gt_sB; a.gt_sT = c.b.gt_sT;
    a.ent_mask = c.w.emc; a.ent_mask0 = c.w.em0; a.group_bits = c.b.group_bits;
    a.ldq = w; a.ldkv = 2 * w; a.ldo = w;
    return a;
}

// ------------------------------------------------------------------------------------------------
// agent forward (entity_rnn_agent.py:31-64 with the G mask variants of :116-124)
// ------------------------------------------------------------------------------------------------
enum { AG_PRE = 1, AG_GRU = 2, AG_POST = 4, AG_ALL = 7, AG_NO_ENTITY = 8 };     // AG_NO_ENTITY: fc1 .. attention core already done (agent_entity_dual)

static refil_gru_desc agent_gru_desc(const Ctx& c, const float* P, const AgentBufs& b, int G, bool zero_h0 = false) {
    const refil_dims& d = c.d;
    refil_gru_desc g;
    memset(&g, 0, sizeof(g));
    g.gi = b.gi; g.hsx = b.hsx; g.w_hh = P + c.L.ag_w_hh; g.b_hh = P + c.L.ag_b_hh;
    g.save_r = b.sr; g.save_z = b.sz; g.save_n = b.sn; g.save_ghn = b.sg;
    g.NR = G * d.B * d.na; g.T1 = d.T1; g.na = d.na; g.H = d.H;
    if (c.lists) { g.t_last = c.w.t_last; g.B = d.B; }
    if (c.lists && c.compose_agent) g.ever = c.w.ever;     // (everything behind the recurrence runs on list_t: ever-active agents only)
    g.zero_h0 = zero_h0 ? 1 : 0;
    return g;
}

// The entity-side layers (fc1, K/V and Q projections, attention core) of the LIVE and the TARGET agent as two-net launches:
// same rows, same masks, different parameters and buffers (batch strides = pointer differences). Five launches instead of
// ten on the agent chain, and twice the tiles per projection launch (learner steps on row lists).
static int agent_entity_dual(const Ctx& c, const float* Pl, const float* Pt, const AgentBufs& bl, const AgentBufs& bt, int G) {
    const refil_dims& d = c.d; const Sizes& s = c.s; const refil_param_layout& L = c.L;
    const int dd = d.d;
    const long dP = Pt - Pl;
    // in_trans + attention core as one launch (attention_qkv.hip) for the live (Q / K / V stored for the backward) and / or the
    // target agent (nothing stored); a net that is not fused keeps its projection launches
    const bool fl = (c.qkv & QKV_L_AGENT) != 0, ft = (c.qkv & QKV_T_AGENT) != 0;
    {
        refil_gemm_desc g = with_rows(linear(c.w.xe, s.Ep, Pl + L.ag_fc1_w, s.E, Pl + L.ag_fc1_b, bl.x1, dd, s.NE, dd, s.E, REFIL_GEMM_RELU), c, rows_ea(c));
        g.batch = 2; g.sA = 0; g.sB = dP; g.sBias = dP; g.sC = bt.x1 - bl.x1;
        RUN(gemm_launch(g, c.st));
    }
    const int n0 = fl ? 1 : 0, n1 = ft ? 1 : 2;           // nets [n0, n1) run the separate projections
    if (n0 < n1) {
        const AgentBufs& b0 = n0 ? bt : bl; const float* P0 = n0 ? Pt : Pl;
        refil_gemm_desc g = with_rows(linear(b0.x1, dd, P0 + L.ag_in_w + (long)dd * dd, dd, nullptr, b0.kv, 2 * dd, s.NE, 2 * dd, dd, 0), c, rows_ea(c));
        g.batch = n1 - n0; g.sA = bt.x1 - bl.x1; g.sB = dP; g.sC = bt.kv - bl.kv;
        RUN(gemm_launch(g, c.st));
        refil_gemm_desc q = linear(b0.x1, dd, P0 + L.ag_in_w, dd, nullptr, b0.q, dd, s.NA, dd, dd, 0);
        q.a_map = agent_rows(c);
        q = with_rows(q, c, rows_a(c));
        q.batch = n1 - n0; q.sA = bt.x1 - bl.x1; q.sB = dP; q.sC = bt.q - bl.q;
        RUN(gemm_launch(q, c.st));
    }
    refil_attn_desc ad[2];
    AttnNetOpts ao[2] = {AttnNetOpts{0, 0}, AttnNetOpts{0, 0}};
    AttnQkvSrc qs[2];
    for (int n = 0; n < 2; ++n) {
        const AgentBufs& b = n ? bt : bl;
        refil_attn_desc a = attn_base(c, dd);
        a.Q = b.q; a.K = b.kv; a.V = b.kv + dd; a.O = b.ao; a.sO = s.NA * dd;
        a.nvar = n ? 1 : G; a.var[0] = REFIL_MASK_OBS;
        a.var[1] = group_code(d, 0, true);
        a.var[2] = group_code(d, 1, true);
        attn_rows(c, a, false);
        ad[n] = a;
        qs[n] = AttnQkvSrc{b.x1, (n ? Pt : Pl) + L.ag_in_w, n ? nullptr : b.q, n ? nullptr : b.kv, n ? nullptr : b.kv + dd};
    }
    RUN(stream_after(c.sd, c.mwst, c.st));
    if (fl || ft) {
        // (both fused: ONE launch, 2 x heads slices; else the fused net's launch and the other net's attention core)
        const int f0 = fl ? 0 : 1, nf = (fl && ft) ? 2 : 1;
        const int rcq = attn_qkv_launch_multi(ad + f0, ao + f0, qs + f0, nf, dd, c.st, nullptr, 1);
        REFIL_CHECK(rcq >= 0, "refil: fused in_trans + attention shape not instantiated");
        if (rcq) return rcq;
        if (fl && ft) return 0;
        const int rc1 = attn_mfma_launch_ex(ad[fl ? 1 : 0], false, c.st, 0, nullptr, 0, 1);
        REFIL_CHECK(rc1 >= 0, "refil: agent attention shape not instantiated");
        return rc1;
    }
    // the two attention cores: one launch up to 16 entities (launch-bound shapes: cfg2 1.4 % faster), one launch per net
    // above (a second job per wave lengthens every workgroup: cfg-T 1.6 % faster with two launches). REFIL_AGENT_DUAL=2 / 3
    // force two launches / one
    static const int dual_mode = [] { const char* e = getenv("REFIL_AGENT_DUAL"); return e ? atoi(e) : 1; }();
    const bool split_attn = dual_mode == 2 || (dual_mode != 3 && d.ne > 16);
    if (split_attn) {
        for (int n = 0; n < 2; ++n) {
            const int rc1 = attn_mfma_launch_ex(ad[n], false, c.st, 0, nullptr, 0, 1);
            REFIL_CHECK(rc1 >= 0, "refil: agent attention shape not instantiated");
            if (rc1) return rc1;
        }
        return 0;
    }
    const int rc = attn_mfma_launch_multi(ad, ao, 2, false, c.st, nullptr, 1);      // inactive agents -> exact zeros
    REFIL_CHECK(rc >= 0, "refil: agent attention shape not instantiated");
    return rc;
}

// phases: AG_PRE everything up to the GRU input gates, AG_GRU the recurrence, AG_POST fc3 (the learner runs the live and
// the target agent's recurrences in ONE launch between their PRE and POST parts)
static int agent_forward(const Ctx& c, const float* P, const AgentBufs& b, int G, const float* h0, int phases = AG_ALL, bool target = false) {
    const refil_dims& d = c.d; const Sizes& s = c.s; const refil_param_layout& L = c.L;
    const int dd = d.d, H = d.H;
    const bool fused = (c.qkv & (target ? QKV_T_AGENT : QKV_L_AGENT)) != 0;      // in_trans + attention core as one launch
    if (phases & AG_PRE) {
    if (!(phases & AG_NO_ENTITY)) {
    // x1 = relu(fc1(entities))                                       :38
    RUN(gemm_launch(with_rows(linear(c.w.xe, s.Ep, P + L.ag_fc1_w, s.E, P + L.ag_fc1_b, b.x1, dd, s.NE, dd, s.E, REFIL_GEMM_RELU), c, rows_ea(c)), c.st));
    if (d.pooling) {
        // EntityPoolingLayer: in_trans (with bias) on all entities, masked mean / max pool       attention.py:110-123
        RUN(gemm_launch(linear(b.x1, dd, P + L.ag_in_w, dd, P + L.ag_in_w + (long)dd * dd, b.kv, 2 * dd, s.NE, dd, dd, 0), c.st));
        refil_attn_desc a = attn_base(c, dd);
        a.K = b.kv; a.O = b.ao; a.sO = s.NA * dd;
        a.nvar = G; a.var[0] = REFIL_MASK_OBS;
        a.var[1] = group_code(d, 0, true);
        a.var[2] = group_code(d, 1, true);
        RUN(pool_launch(a, d.pooling, false, c.st));
    } else {
    // K,V for all entities; Q for the agents only                    attention.py:46-48
    if (!fused) {
    RUN(gemm_launch(with_rows(linear(b.x1, dd, P + L.ag_in_w + (long)dd * dd, dd, nullptr, b.kv, 2 * dd, s.NE, 2 * dd, dd, 0), c, rows_ea(c)), c.st));
    {
        refil_gemm_desc g = linear(b.x1, dd, P + L.ag_in_w, dd, nullptr, b.q, dd, s.NA, dd, dd, 0);
        g.a_map = agent_rows(c);
        RUN(gemm_launch(with_rows(g, c, rows_a(c)), c.st));
    }
    }
    {
        refil_attn_desc a = attn_base(c, dd);
        a.Q = b.q; a.K = b.kv; a.V = b.kv + dd; a.O = b.ao; a.sO = s.NA * dd;
        a.nvar = G; a.var[0] = REFIL_MASK_OBS;
        a.var[1] = group_code(d, 0, true);
        a.var[2] = group_code(d, 1, true);
        attn_rows(c, a, false);
        RUN(stream_after(c.sd, c.mwst, c.st));
        if (fused) {
            const AttnNetOpts ao{0, 0};
            const AttnQkvSrc qs{b.x1, P + L.ag_in_w, target ? nullptr : b.q, target ? nullptr : b.kv, target ? nullptr : b.kv + dd};
            const int rc = attn_qkv_launch_multi(&a, &ao, &qs, 1, dd, c.st, nullptr, c.compose_agent ? 1 : 0);
            REFIL_CHECK(rc >= 0, "refil: fused in_trans + attention shape not instantiated");
            if (rc) return rc;
        } else if (c.compose_agent) {
            const int rc = attn_mfma_launch_ex(a, false, c.st, 0, nullptr, 0, 1);      // inactive agents -> exact zeros
            REFIL_CHECK(rc >= 0, "refil: agent attention shape not instantiated");
            if (rc) return rc;
        } else RUN(attn_forward_launch(a, c.st));
    }
    }
    }   // !AG_NO_ENTITY
    if (d.agent_ff) {
        // feed-forward agent (entity_ff_agent.py:40-52): x2 = relu(out_trans(attn)) (inactive agents zeroed), q = fc2(x2)
        refil_gemm_desc g = linear(b.ao, dd, P + L.ag_out_w, dd, P + L.ag_out_b, b.x2, dd, (long)G * s.NA, dd, dd, REFIL_GEMM_RELU);
        g.rowmask = c.w.amask; g.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(g, c.st));
        refil_gemm_desc f = linear(b.x2, dd, P + L.ag_fc2_w, dd, P + L.ag_fc2_b, b.qv, d.A, (long)G * s.NA, d.A, dd, 0);
        f.rowmask = c.w.amask; f.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(f, c.st));
        return 0;
    }
    if (c.compose_agent) {
        // x3 = relu(fc2(mask(out_trans(a)))) = relu(a W_c^T + b_2 + active * (W_2 b_o)),  a = 0 for inactive agents
        ComposeArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.W2 = P + L.ag_fc2_w; ca.b2 = P + L.ag_fc2_b; ca.Wo = P + L.ag_out_w; ca.bo = P + L.ag_out_b;
        ca.Wc = b.wc; ca.bc = b.bc; ca.bd = b.bd; ca.nets = 1; ca.M = H; ca.h = dd;
        if (!c.agent_composed && !c.skip_compose) RUN(compose_forward_launch(ca, c.st));
        refil_gemm_desc g = linear(b.ao, dd, b.wc, dd, P + L.ag_fc2_b, b.x3, H, (long)G * s.NA, H, dd, REFIL_GEMM_RELU);
        g.bias2 = b.bd; g.rowscale = c.w.actf; g.rowscale_mod = (int)s.NA;
        RUN(gemm_launch(with_rows(g, c, rows_t(c, G)), c.st));
    } else {
    // x2 = out_trans(attn) with inactive agents zeroed                attention.py:65-67
    {
        refil_gemm_desc g = linear(b.ao, dd, P + L.ag_out_w, dd, P + L.ag_out_b, b.x2, dd, (long)G * s.NA, dd, dd, 0);
        g.rowmask = c.w.amask; g.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(g, c.st));
    }
    // x3 = relu(fc2(x2))                                              :46
    RUN(gemm_launch(linear(b.x2, dd, P + L.ag_fc2_w, dd, P + L.ag_fc2_b, b.x3, H, (long)G * s.NA, H, dd, REFIL_GEMM_RELU), c.st));
    }
    // gi = x3 W_ih^T + b_ih for all steps, then the persistent recurrence   :49-55
    {
        refil_gemm_desc g = linear(b.x3, H, P + L.ag_w_ih, H, P + L.ag_b_ih, b.gi, 3 * H, (long)G * s.NA, 3 * H, H, 0);
        if (c.compose_agent) g = with_rows(g, c, rows_t(c, G));       // (x3 exists on the listed rows only)
        RUN(gemm_launch(g, c.st));
    }
    if (h0 || d.agent_ff) RUN(set_h0_launch(b.hsx, h0, G * d.B, d.T1, d.na, H, c.st));      // (h0 = NULL: the recurrence kernel zero-fills slot 0)
    }   // AG_PRE
    if (d.agent_ff) return 0;
    if (phases & AG_GRU) RUN(gru_forward_launch(agent_gru_desc(c, P, b, G, h0 == nullptr), c.st));
    // q = fc3(h), zero for inactive agents                            :57-60
    if (phases & AG_POST) {
        refil_gemm_desc g = linear(b.hsx, H, P + L.ag_fc3_w, H, P + L.ag_fc3_b, b.qv, d.A, (long)G * s.NA, d.A, H, 0);
        g.a_map = hs_rows(c, d.na);
        g.rowmask = c.w.amask; g.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(g, c.st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// the four attention hypernets up to the masked fc2 output (flex_qmix.py:41-50)
// variant order in ao/x2/x3: hyper_w_1 under nv0 masks, then hyper_w_final, hyper_b_1, V
// ------------------------------------------------------------------------------------------------
enum { HY_PRE = 1, HY_ATTN = 2, HY_POST = 4, HY_ALL = 7 };
// phases: HY_PRE fc1 + K/V/Q projections, HY_ATTN the attention cores, HY_POST the tails. `second` (HY_ATTN only): another set
// of hypernets under the SAME masks (the target mixer's, single-variant) whose attention cores join this launch -- 8 nets, one
// launch: a launch's fixed per-row cost (mask words, first operand fetch, ramp) is paid once
// `target`: these are the target mixer's hypernets (never differentiated: the fused launch stores no Q / K / V for them);
// `Psecond`: the parameters of `second`'s nets (the fused launch reads in_trans.weight itself)
static int hyper_forward(const Ctx& c, const float* P, const HyperBufs& b, int nv0, int phases = HY_ALL, const HyperBufs* second = nullptr,
                         bool target = false, const float* Psecond = nullptr) {
    const refil_dims& d = c.d; const Sizes& s = c.s; const refil_param_layout& L = c.L;
    const int h = d.hyp, M = d.M, nets = s.nets;
    // in_trans + attention core as one launch (attention_qkv.hip): no K/V / Q projection launches, no Q / K / V round trip
    const bool fused = (c.qkv & (target ? QKV_T_HYPER : QKV_L_HYPER)) != 0 && (!second || (c.qkv & QKV_T_HYPER));
    // The composed tail maps depend on the parameters alone: built in front of the projections they are off the dependent
    // launches between the attention core and the join of the chains (A/B on one box: cfg-T 1.752 -> 1.742 ms; at 16 entities,
    // where the step is launch-bound rather than saturated, 0.754 -> 0.758: there they stay behind the attention core).
    // REFIL_COMPOSE_EARLY=0/1 forces it
    static const int compose_env = [] { const char* e = getenv("REFIL_COMPOSE_EARLY"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool compose_early = g_tuning.compose_early >= 0 ? g_tuning.compose_early == 1 : (compose_env >= 0 ? compose_env == 1 : d.ne > 16);
    auto compose = [&]() -> int {
        ComposeArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.W2 = P + L.mix_fc2_w; ca.sW2 = L.mix_fc2_w_stride; ca.b2 = P + L.mix_fc2_b; ca.sb2 = L.mix_fc2_b_stride;
        ca.Wo = P + L.mix_out_w; ca.sWo = L.mix_out_w_stride; ca.bo = P + L.mix_out_b; ca.sbo = L.mix_out_b_stride;
        ca.Wc = b.wc; ca.bc = b.bc; ca.nets = nets; ca.M = M; ca.h = h;
        return c.skip_compose ? 0 : compose_forward_launch(ca, c.st);
    };
    if (phases & HY_PRE) {
    if (c.presum && compose_early) RUN(compose());
    RUN(gemm_launch(with_rows(linear(c.w.xe, s.Ep, P + L.mix_fc1_w, s.E, P + L.mix_fc1_b, b.x1, nets * h, s.NE, nets * h, s.E, REFIL_GEMM_RELU), c, rows_eh(c)), c.st));
    if (d.pooling) {
        refil_gemm_desc g = linear(b.x1, nets * h, P + L.mix_in_w, h, P + L.mix_in_w + (long)h * h, b.kv, 2 * h, s.NE, h, h, 0);
        g.batch = nets; g.sA = h; g.sB = L.mix_in_w_stride; g.sBias = L.mix_in_w_stride; g.sC = s.NEa * 2 * h;
        RUN(gemm_launch(g, c.st));
    } else if (!fused) {
        refil_gemm_desc g = linear(b.x1, nets * h, P + L.mix_in_w + (long)h * h, h, nullptr, b.kv, 2 * h, s.NE, 2 * h, h, 0);
        g.batch = nets; g.sA = h; g.sB = L.mix_in_w_stride; g.sC = s.NEa * 2 * h;
        RUN(gemm_launch(with_rows(g, c, rows_eh(c)), c.st));
        refil_gemm_desc q = linear(b.x1, nets * h, P + L.mix_in_w, h, nullptr, b.q, h, s.NA, h, h, 0);
        q.a_map = agent_rows(c);
        q.batch = nets; q.sA = h; q.sB = L.mix_in_w_stride; q.sC = s.NAa * h;
        RUN(gemm_launch(with_rows(q, c, rows_a(c)), c.st));
    }
    }   // HY_PRE
    if (phases & HY_ATTN) {
        // the hypernets' attention cores: ONE launch when the matrix-core kernel covers the shape (shared mask words, jobs
        // of a row pipelined), else one launch per net
        refil_attn_desc ad[8];
        AttnNetOpts ao[8];
        AttnQkvSrc qs[8];
        const int nsets = second ? 2 : 1;
        REFIL_CHECK(nsets * nets <= 8, "refil: too many hypernets for one attention launch");
        for (int set = 0; set < nsets; ++set) {
            const HyperBufs& bb = set == 0 ? b : *second;
            const int nv = set == 0 ? nv0 : 1;
            const float* Pn = set == 0 ? P : Psecond;
            const bool store = !(set == 0 ? target : true);      // (a `second` set is the target mixer's)
            for (int n = 0; n < nets; ++n) {
                if (fused) {
                    float* kvn = bb.kv + (long)n * s.NEa * 2 * h;
                    qs[set * nets + n] = AttnQkvSrc{bb.x1 + (long)n * h, Pn + L.mix_in_w + n * L.mix_in_w_stride,
                                                    store ? bb.q + (long)n * s.NAa * h : nullptr, store ? kvn : nullptr, store ? kvn + h : nullptr};
                }
                refil_attn_desc a = attn_base(c, h);
                attn_rows(c, a, true);
                a.Q = bb.q + (long)n * s.NAa * h; a.K = bb.kv + (long)n * s.NEa * 2 * h; a.V = a.K + h;
                a.O = bb.ao + (long)(n == 0 ? 0 : nv + n - 1) * s.NA * h; a.sO = s.NA * h;
                a.nvar = n == 0 ? nv : 1;
                a.var[0] = REFIL_MASK_ENTITY;
                a.var[1] = group_code(d, 0, false);
                a.var[2] = group_code(d, 1, false);
                ad[set * nets + n] = a;
                ao[set * nets + n] = AttnNetOpts{(c.presum && n > 0) ? 1 : 0, 0};
            }
        }
        int rc = -1;
        RUN(stream_after(c.sd, c.mwst, c.st));
        if (fused) {
            REFIL_CHECK(!second || Psecond, "refil: merged fused hypernet launch needs the second set's parameters");
            rc = attn_qkv_launch_multi(ad, ao, qs, nsets * nets, nets * h, c.st, c.presum ? c.w.nact : nullptr, 0);
            REFIL_CHECK(rc >= 0, "refil: fused in_trans + attention shape not instantiated");
        } else if (!d.pooling && attn_mfma_supported(d.ne, d.na, h / d.heads))
            rc = attn_mfma_launch_multi(ad, ao, nsets * nets, false, c.st, c.presum ? c.w.nact : nullptr, 0);
        if (rc > 0) return rc;
        if (rc < 0) {
            REFIL_CHECK(!c.presum, "refil: agent-sum attention shape not instantiated");
            for (int n = 0; n < nsets * nets; ++n) {
                if (d.pooling) RUN(pool_launch(ad[n], d.pooling, false, c.st));
                else RUN(attn_forward_launch(ad[n], c.st));
            }
        }
    }
    if (!(phases & HY_POST)) return 0;
    if (c.presum) {
        // out_trans o fc2 is one linear map per hypernet: x3 = mask(a W_c^T + b_c), W_c = W_2 W_o (kernels.h: ComposeArgs).
        // x2 is never formed. hyper_w_1 (matrix mode) per agent row; the other nets on the agent-summed rows.
        if (!compose_early) RUN(compose());
        refil_gemm_desc g = linear(b.ao, h, b.wc, h, b.bc, b.x3, M, (long)nv0 * s.NA, M, h, 0);
        g.rowmask = c.w.amask; g.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(with_rows(g, c, rows_h(c, nv0)), c.st));
        refil_gemm_desc f = linear(b.ao + (long)nv0 * s.NA * h, h, b.wc + (long)M * h, h, nullptr, b.x3 + (long)nv0 * s.NA * M, M, s.R, M, h, 0);
        f.batch = nets - 1; f.sA = s.NA * h; f.sB = (long)M * h; f.sC = s.NA * M;
        f.bias2 = b.bc + M; f.sBias = M; f.rowscale = c.w.nact; f.rowscale_mod = (int)s.R;      // + n_act[r] * b_c
        RUN(gemm_launch(f, c.st));
        return 0;
    }
    // out_trans and fc2, both with inactive agents zeroed (attention.py:65-67, flex_qmix.py:49-50)
    for (int part = 0; part < 2; ++part) {
        const long M_rows = part == 0 ? nv0 * s.NA : s.NA;
        const int batch = part == 0 ? 1 : nets - 1;
        const long voff = part == 0 ? 0 : nv0;
        const int net0 = part == 0 ? 0 : 1;
        refil_gemm_desc g = linear(b.ao + voff * s.NA * h, h, P + L.mix_out_w + net0 * L.mix_out_w_stride, h,
                                   P + L.mix_out_b + net0 * L.mix_out_b_stride, b.x2 + voff * s.NA * h, h, M_rows, h, h, 0);
        g.batch = batch; g.sA = s.NA * h; g.sB = L.mix_out_w_stride; g.sBias = L.mix_out_b_stride; g.sC = s.NA * h;
        g.rowmask = c.w.amask; g.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(g, c.st));
        refil_gemm_desc f = linear(b.x2 + voff * s.NA * h, h, P + L.mix_fc2_w + net0 * L.mix_fc2_w_stride, h,
                                   P + L.mix_fc2_b + net0 * L.mix_fc2_b_stride, b.x3 + voff * s.NA * M, M, M_rows, M, h, 0);
        f.batch = batch; f.sA = s.NA * h; f.sB = L.mix_fc2_w_stride; f.sBias = L.mix_fc2_b_stride; f.sC = s.NA * M;
        f.rowmask = c.w.amask; f.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(f, c.st));
    }
    return 0;
}

static MixArgs mix_args(const Ctx& c, const HyperBufs& b, int nv0, const float* qs, int G_qs, int t_off, int T) {
    const refil_dims& d = c.d; const Sizes& s = c.s;
    MixArgs m;
    memset(&m, 0, sizeof(m));
    m.x_w1 = b.x3; m.s_var = s.NA * d.M;
    m.lin = d.mixer_vdn ? 2 : d.mixer_lin;
    if (d.mixer_vdn) {
        // parameter-free sum: no hypernet outputs
    } else if (d.mixer_lin) {                                  // hypernets (hyper_w_1, V)
        m.x_v = b.x3 + (long)(nv0 + 0) * s.NA * d.M;
    } else {                                            // (hyper_w_1, hyper_w_final, hyper_b_1, V)
        m.x_wf = b.x3 + (long)(nv0 + 0) * s.NA * d.M;
        m.x_b1 = b.x3 + (long)(nv0 + 1) * s.NA * d.M;
        m.x_v = b.x3 + (long)(nv0 + 2) * s.NA * d.M;
    }
    m.qs = qs; m.s_qs_g = (long)d.B * T * d.na;
    m.amask = c.w.amask;
    m.B = d.B; m.T1 = d.T1; m.T = T; m.t_off = t_off; m.na = d.na; m.M = d.M;
    m.imagine = (nv0 == 3 && G_qs == 3) ? 1 : 0;
    m.softmax_w = d.softmax_mixing_weights; m.tanh_nl = d.mixer_tanh;
    m.presum = c.presum ? 1 : 0;
    m.t_last = c.lists ? c.w.t_last : nullptr;
    return m;
}

// ------------------------------------------------------------------------------------------------
// backward of one attention block: given d(x2) (already row-masked) produce dW_out, db_out, then
// through the attention core and in_trans down to d(x1) (ReLU-masked). Shared by agent + hypernets.
// ------------------------------------------------------------------------------------------------
struct AttnBlockBwd {
    int w;                    // embed width
    int nets;                 // 1 (agent) or 4 (hypernets)
    int nv0;                  // mask variants of net 0
    const float* P; float* Gr;          // params / grads flat buffers
    long in_w, in_w_stride, out_w, out_w_stride, out_b, out_b_stride;
    const float *x1, *kv, *q, *ao;      // forward activations (x1 [NE, nets*w], kv [nets][NE,2w], q [nets][NA,w], ao [NV][NA,w])
    const float* dx2;                    // [NV][NA,w]
    float *dao, *dq, *dkv, *dx1;         // scratch / outputs
    int var_first[3];                    // mask codes of net 0's variants
    int var_rest;                        // mask code of nets 1..3
    bool presum;                         // nets 1..3 run on agent-summed rows (Ctx::presum)
    bool hyper;                          // which entity-row list / dead-key flags apply (hypernets or agent nets)
};

static int attn_block_backward(const Ctx& c, const AttnBlockBwd& k) {
    const Sizes& s = c.s; const refil_dims& d = c.d;
    const int w = k.w;
    const long ldx1 = (long)k.nets * w;
    for (int part = 0; part < (k.presum ? 0 : (k.nets > 1 ? 2 : 1)); ++part) {     // (presum: d(attn out) already in k.dao)
        const long rows = part == 0 ? k.nv0 * s.NA : s.NA;
        const int batch = part == 0 ? 1 : k.nets - 1;
        const long voff = part == 0 ? 0 : k.nv0;
        const int net0 = part == 0 ? 0 : 1;
        // dW_out = dx2^T ao ; db_out = colsum(dx2)
        refil_gemm_desc gw = linear_dw(k.dx2 + voff * s.NA * w, w, k.ao + voff * s.NA * w, w,
                                       k.Gr + k.out_w + net0 * k.out_w_stride, w, k.Gr + k.out_b + net0 * k.out_b_stride,
                                       rows, w, w, c.w.partial, batch);
        gw.sA = s.NA * w; gw.sB = s.NA * w; gw.sC = k.out_w_stride; gw.sColsum = k.out_b_stride;
        RUN(launch_dw(c, gw));
        // d(attn_out) = dx2 W_out
        refil_gemm_desc gx = linear_dx(k.dx2 + voff * s.NA * w, w, k.P + k.out_w + net0 * k.out_w_stride, w,
                                       k.dao + voff * s.NA * w, w, rows, w, w, 0);
        gx.batch = batch; gx.sA = s.NA * w; gx.sB = k.out_w_stride; gx.sC = s.NA * w;
        RUN(gemm_launch(gx, c.st));
    }
    const RowList re = k.hyper ? rows_eh(c) : rows_ea(c);
    {
        refil_attn_desc ad[4];
        AttnNetOpts ao[4];
        for (int n = 0; n < k.nets; ++n) {
            refil_attn_desc a = attn_base(c, w);
            attn_rows(c, a, k.hyper);
            a.Q = k.q + (long)n * s.NAa * w; a.K = k.kv + (long)n * s.NEa * 2 * w; a.V = a.K + w;
            a.dO = k.dao + (long)(n == 0 ? 0 : k.nv0 + n - 1) * s.NA * w; a.sO = s.NA * w;
            a.dQ = k.dq + (long)n * s.NAa * w; a.dK = k.dkv + (long)n * s.NEa * 2 * w; a.dV = a.dK + w;
            a.nvar = n == 0 ? k.nv0 : 1;
            a.var[0] = k.var_first[0]; a.var[1] = k.var_first[1]; a.var[2] = k.var_first[2];
            if (n > 0) a.var[0] = k.var_rest;
            ad[n] = a;
            ao[n] = AttnNetOpts{0, (k.presum && n > 0) ? 1 : 0};       // presum: dO is one row per (b,t) for all its agents
        }
        int rc = -1;
        if (!d.pooling && attn_mfma_supported(d.ne, d.na, w / d.heads) && (k.nets == 1 || k.var_rest == k.var_first[0]))
            rc = attn_mfma_launch_multi(ad, ao, k.nets, true, c.st, nullptr, 0);
        if (rc > 0) return rc;
        if (rc < 0) {
            REFIL_CHECK(!(k.presum && k.nets > 1), "refil: broadcast-dO attention shape not instantiated");
            for (int n = 0; n < k.nets; ++n) {
                if (d.pooling) RUN(pool_launch(ad[n], d.pooling, true, c.st));       // d(in_trans output) -> dkv (first w columns)
                else RUN(attn_backward_launch(ad[n], c.st));
            }
        }
    }
    if (d.pooling) {
        // EntityPoolingLayer: dW_in = dE^T x1, db_in = colsum(dE);  dx1 = relu'(x1) * (dE W_in)
        refil_gemm_desc g = linear_dw(k.dkv, 2 * w, k.x1, (int)ldx1, k.Gr + k.in_w, w, k.Gr + k.in_w + (long)w * w, s.NE, w, w,
                                      c.w.partial, k.nets);
        g.sA = s.NEa * 2 * w; g.sB = w; g.sC = k.in_w_stride; g.sColsum = k.in_w_stride;
        RUN(launch_dw(c, g));
        refil_gemm_desc x = linear_dx(k.dkv, 2 * w, k.P + k.in_w, w, k.dx1, (int)ldx1, s.NE, w, w, REFIL_GEMM_RELU_BWD);
        x.aux = k.x1; x.batch = k.nets; x.sA = s.NEa * 2 * w; x.sB = k.in_w_stride; x.sC = w;
        RUN(gemm_launch(x, c.st));
        return 0;
    }
    // dW_in rows [w,3w) = dKV^T x1 ; rows [0,w) = dQ^T x1[agent rows]
    {
        refil_gemm_desc g = linear_dw(k.dkv, 2 * w, k.x1, (int)ldx1, k.Gr + k.in_w + (long)w * w, w, nullptr, s.NE, 2 * w, w,
                                      c.w.partial, k.nets);
        g.sA = s.NEa * 2 * w; g.sB = w; g.sC = k.in_w_stride;
        RUN(launch_dw(c, with_rows(g, c, re)));
        refil_gemm_desc q = linear_dw(k.dq, w, k.x1, (int)ldx1, k.Gr + k.in_w, w, nullptr, s.NA, w, w, c.w.partial, k.nets);
        q.b_map = refil_rowmap{d.na, d.ne, 0};
        q.sA = s.NAa * w; q.sB = w; q.sC = k.in_w_stride;
        if (!(c.tail_dw & 1)) RUN(launch_dw(c, with_rows(q, c, rows_a(c))));
    }
    // dx1 = relu'(x1) * (dKV W_kv + scatter(dQ W_q))
    {
        refil_gemm_desc g = linear_dx(k.dkv, 2 * w, k.P + k.in_w + (long)w * w, w, k.dx1, (int)ldx1, s.NE, 2 * w, w, REFIL_GEMM_RELU_BWD);
        g.aux = k.x1; g.batch = k.nets; g.sA = s.NEa * 2 * w; g.sB = k.in_w_stride; g.sC = w;
        RUN(gemm_launch(with_rows(g, c, re), c.st));
        refil_gemm_desc q = linear_dx(k.dq, w, k.P + k.in_w, w, k.dx1, (int)ldx1, s.NA, w, w, REFIL_GEMM_RELU_BWD | REFIL_GEMM_ACCUM);
        q.aux = k.x1; q.c_map = refil_rowmap{d.na, d.ne, 0};
        q.batch = k.nets; q.sA = s.NAa * w; q.sB = k.in_w_stride; q.sC = w;
        RUN(gemm_launch(with_rows(q, c, rows_a(c)), c.st));
    }
    if (c.tail_dw & 1) {
        refil_gemm_desc q = linear_dw(k.dq, w, k.x1, (int)ldx1, k.Gr + k.in_w, w, nullptr, s.NA, w, w, c.w.partial, k.nets);
        q.b_map = refil_rowmap{d.na, d.ne, 0};
        q.sA = s.NAa * w; q.sB = w; q.sC = k.in_w_stride;
        RUN(launch_dw_tail(c, with_rows(q, c, rows_a(c)), 1));
    }
    return 0;
}

static int copy_out(float* dst, const float* src, long n, hipStream_t st) {
    if (!dst) return 0;
    REFIL_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

// What the library remembers about a workspace arena between calls: the layout of its last carve (NaN-pattern regions are cleared
// when it changes), the early-prologue slot and the events behind the slots' last readers. Keyed by (device, arena address) -- the
// events belong to the device that was current when they were created -- per thread like the side streams whose work they order;
// refil_release_streams destroys the events and forgets the arenas.
struct Prev { refil_dims d; int mode; int slot; hipEvent_t slot_free[2]; hipEvent_t pre_done; uint64_t target_version; const void* target_ptr;
              int compose_state; };      // which composed target maps the arena holds: bit 0 hypernets (Ctx::presum), bit 1 agent (Ctx::compose_agent)
typedef std::pair<int, void*> PrevKey;
static thread_local std::map<PrevKey, Prev> g_prev;

static int make_ctx(Ctx& c, const refil_dims* dims, const refil_batch* batch, void* ws, size_t ws_bytes, CarveMode mode, void* stream) {
    REFIL_CHECK(dims && batch && ws, "refil: null dims/batch/workspace");
    if (int e = check_dims(*dims)) return e;
    c.d = *dims; c.s = sizes_of(*dims); c.b = *batch; c.st = (hipStream_t)stream;
    c.gst = c.st; c.gpartial = nullptr; c.sd = nullptr; c.mwst = c.st; c.tail_dw = 0; c.tpartial = nullptr; c.defer = nullptr;
    c.qkv = 0;
    c.same_layout = false; c.slot = 0; c.slot_free = nullptr; c.pre_done = nullptr; c.mw_nvar = c.s.G; c.target_same = false; c.prev = nullptr; c.agent_composed = false; c.skip_compose = false;
    param_layout(c.d, c.L);
    const char* pe = getenv("REFIL_PRESUM");          // read per call: tests compare both paths in one process
    const bool presum_on = !(pe && pe[0] == '0');
    c.presum = presum_on && !dims->mixer_lin && !dims->mixer_vdn && !dims->mixer_none && !dims->pooling &&
               attn_mfma_supported(dims->ne, dims->na, dims->hyp / dims->heads);
    c.compose_agent = presum_on && !dims->agent_ff && !dims->pooling && attn_mfma_supported(dims->ne, dims->na, dims->d / dims->heads);
    {
        const char* de = getenv("REFIL_DENSE");
        const refil_dims& d = *dims;
        const int E = in_dim(d);
        // every listed GEMM must take the weight-resident / streaming-dW kernels (gemm_wres_eligible): whole 32-column
        // tiles, reductions <= 128 (<= 256 through the ReLU and for the fc1 layers), 16-byte aligned rows, enough rows for a split reduction
        // (widths: whole 64-column tiles -- the dX launches through a ReLU run 64-wide tiles only --, the hypernets' tail writes
        // mixing_embed_dim columns: whole 32-column tiles; rnn_hidden_dim 32 / 128 and everything else: the dense schedule. Found by
        // the shape fuzz of round 4: with `% 32` here, 85 of 100 random production-size shapes were turned away by refil_gemm)
        const bool shapes = E % 4 == 0 && E <= 256 && d.d % 64 == 0 && d.d <= 128 && d.hyp % 64 == 0 && d.hyp <= 128 &&
                            d.M % 32 == 0 && d.H == 64 && c.s.NE >= 2048 && c.s.NA >= 512;
        c.lists = mode == CARVE_LEARNER && !(de && de[0] == '1') && shapes && c.presum && c.compose_agent;
        const char* me = getenv("REFIL_MASKWORDS");
        // (A/B with the words against the attention kernels' own mask phase: 32 / 48 entities 3.6 % / 4.7 % faster; 16 entities
        // 1.8 % (cfg2) / 2.1 % (cfg4) faster since the kernels use the row words to leave dead K / V / Q rows unfetched --
        // before that the separate pass cost more than it saved there. REFIL_MASKWORDS=0 switches them off)
        c.mwords = mode == CARVE_LEARNER && !(me && me[0] == '0') && !d.pooling && !d.mixer_vdn && !d.mixer_none &&
                   attn_mfma_supported(d.ne, d.na, d.d / d.heads) && attn_mfma_supported(d.ne, d.na, d.hyp / d.heads);
    }
    {
        static const int qkv_env = [] { const char* e = getenv("REFIL_ATTN_QKV"); return e ? atoi(e) : QKV_DEFAULT; }();
        const refil_dims& d = *dims;
        int q = g_tuning.attn_qkv >= 0 ? (int)g_tuning.attn_qkv : qkv_env;
        if (!c.mwords || d.pooling || !gemm_wres_split_on()) q = 0;          // (the fused launch is the bf16 x 6 arithmetic of the projections)
        if (!attn_qkv_supported(d.ne, d.na, d.heads, d.hyp / d.heads)) q &= ~(QKV_T_HYPER | QKV_L_HYPER);
        if (d.agent_ff || !attn_qkv_supported(d.ne, d.na, d.heads, d.d / d.heads)) q &= ~(QKV_T_AGENT | QKV_L_AGENT);
        // more than 32 entities or 16 agents (BASELINE configs[4]): the three-key-tile / two-agent-tile forms are parity-tested but were
        // never timed on a GPU (round 6 had none): the separate launches stay the default there; REFIL_ATTN_QKV_WIDE=1 /
        // refil_set_tuning("attn_qkv_wide", 1) takes the fused launch
        static const int wide_env = [] { const char* e = getenv("REFIL_ATTN_QKV_WIDE"); return e ? atoi(e) : 0; }();
        if (attn_qkv_wide(d.ne, d.na) && !(g_tuning.attn_qkv_wide >= 0 ? g_tuning.attn_qkv_wide == 1 : wide_env == 1)) q = 0;
        // (the fused launch keeps a row table in LDS that grows with B T1 / workgroups per slice: beyond ~350 k rows per 16 slices the
        // separate launches take over -- sized for the widest launch a chain can issue: both hypernet sets merged, both agents)
        const long Rq = (long)d.B * d.T1;
        if ((q & (QKV_T_HYPER | QKV_L_HYPER)) && !attn_qkv_fits(d.ne, d.na, d.heads, d.hyp / d.heads, Rq, d.T1, 2 * sizes_of(d).nets)) q &= ~(QKV_T_HYPER | QKV_L_HYPER);
        if ((q & (QKV_T_AGENT | QKV_L_AGENT)) && !attn_qkv_fits(d.ne, d.na, d.heads, d.d / d.heads, Rq, d.T1, 2)) q &= ~(QKV_T_AGENT | QKV_L_AGENT);
        c.qkv = q;
    }
    REFIL_CHECK(batch->entities && batch->entity_mask, "refil: batch.entities / entity_mask missing");
    REFIL_CHECK(!dims->entity_last_action || batch->actions, "refil: batch.actions missing");
    REFIL_CHECK((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "refil: workspace must be 256-byte aligned");
    Arena a{(char*)ws, ws_bytes, 0, false};
    carve(a, c.d, c.w, mode);
    REFIL_CHECK(!a.overflow && a.off <= ws_bytes, "refil: workspace too small (%zu < %zu bytes)", ws_bytes, a.off);
    // Skipped rows are never written and what they hold only ever meets exact zeros -- it has to be FINITE. The caller zeroes
    // the arena once; afterwards the only bit patterns of a learner carve that read as NaN are t_last (-1) and the mask words
    // (all ones). When the layout on this workspace changes (another T1 / B, another entry point) float regions of the new
    // carve may overlay them: clear those regions of the previous carve first (a few MB, stream-ordered, only on a change).
    {
        int dev_now = 0;
        REFIL_HIP(hipGetDevice(&dev_now));
        auto& prev = g_prev;
        const PrevKey pkey{dev_now, ws};
        auto it = prev.find(pkey);
        const bool had = it != prev.end();
        const bool same = had && it->second.mode == CARVE_LEARNER && mode == CARVE_LEARNER && memcmp(&it->second.d, dims, sizeof(refil_dims)) == 0;
        if (had && it->second.mode == CARVE_LEARNER && !same) {
            Arena ao{(char*)ws, ws_bytes, 0, false};
            Work wo;
            carve(ao, it->second.d, wo, CARVE_LEARNER);
            const Sizes so = sizes_of(it->second.d);
            if (!ao.overflow) {
                // (t_last of the second prologue slot sits right behind its mask words: one range covers both)
                REFIL_HIP(hipMemsetAsync(wo.t_last, 0, (size_t)it->second.d.B * sizeof(int), c.st));
                REFIL_HIP(hipMemsetAsync(wo.mw_a, 0, (size_t)(reinterpret_cast<char*>(wo.alt.t_last + it->second.d.B) - reinterpret_cast<char*>(wo.mw_a)), c.st));
            }
        }
        if (!had) {
            Prev p0;
            memset(&p0, 0, sizeof(p0));
            for (auto& e : p0.slot_free) REFIL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            REFIL_HIP(hipEventCreateWithFlags(&p0.pre_done, hipEventDisableTiming));
            it = prev.emplace(pkey, p0).first;
        }
        Prev& pr = it->second;
        pr.d = *dims; pr.mode = (int)mode;
        pr.slot = same ? pr.slot ^ 1 : 0;
        c.same_layout = same; c.slot = pr.slot; c.slot_free = pr.slot_free; c.pre_done = pr.pre_done; c.prev = &pr;
        if (mode == CARVE_LEARNER && c.slot) use_alt_slot(c.w);
    }
    return 0;
}

static int run_prep(const Ctx& c, int first_step_zero, int phases = 3) {
    PrepArgs p;
    p.b = c.b; p.B = c.d.B; p.T1 = c.d.T1; p.ne = c.d.ne; p.na = c.d.na; p.ed = c.d.ed; p.A = c.d.A; p.Ep = c.s.Ep;
    p.last_action = c.d.entity_last_action; p.first_step_zero = first_step_zero;
    p.xe = c.w.xe; p.emc = c.w.emc; p.amask = c.w.amask; p.actf = c.w.actf; p.em0 = c.w.em0;
    const bool skip = c.lists && (phases & 2);
    return prep_launch(p, c.st, phases, skip ? c.w.kdead_a : nullptr, skip ? c.w.kdead_h : nullptr);
}

}  // namespace refil

using namespace refil;

extern "C" const char* refil_last_error(void) { return g_err; }
extern "C" int refil_set_mixer_grads_hook(refil_grads_hook hook, void* user) {
    g_mixer_hook = hook; g_mixer_hook_user = user;
    return 0;
}
extern "C" int refil_set_tuning(const char* name, int64_t value) {
    REFIL_CHECK(name, "refil_set_tuning: null name");
    if (!strcmp(name, "dw4_target")) g_tuning.dw4_target = value;
    else if (!strcmp(name, "dws_target")) g_tuning.dws_target = value;
    else if (!strcmp(name, "dw4_min_out")) g_tuning.dw4_min_out = value;
    else if (!strcmp(name, "dw_target")) g_tuning.dw_target = value;
    else if (!strcmp(name, "compose_early")) g_tuning.compose_early = value;
    else if (!strcmp(name, "gru_pd")) g_tuning.gru_pd = value;
    else if (!strcmp(name, "wres_split")) g_tuning.wres_split = value;
    else if (!strcmp(name, "dw_split")) g_tuning.dw_split = value;
    else if (!strcmp(name, "attn_qkv")) g_tuning.attn_qkv = value;
    else if (!strcmp(name, "attn_qkv_wide")) g_tuning.attn_qkv_wide = value;
    else if (!strcmp(name, "qkv_lds_budget")) attn_qkv_set_lds_budget((long)value);      // (bytes; <= 0: the device's 160 KB)
    else { set_error("refil_set_tuning: unknown knob '%s'", name); return 1; }
    return 0;
}
extern "C" int64_t refil_get_stat(const char* name) {
    if (!name) return -1;
    if (!strcmp(name, "learner_steps")) return g_stat_steps;
    if (!strcmp(name, "early_prologue_steps")) return g_stat_early;
    if (!strcmp(name, "early_target_hypernet_steps")) return g_stat_et_h;
    if (!strcmp(name, "early_target_agent_steps")) return g_stat_et_a;
    return -1;
}
extern "C" int refil_set_overlap(int on) { g_overlap = on < 0 ? -1 : (on != 0); return 0; }
extern "C" int refil_version(void) { return 1; }

extern "C" int refil_get_param_layout(const refil_dims* dims, refil_param_layout* out) {
    REFIL_CHECK(dims && out, "refil_get_param_layout: null argument");
    if (int e = check_dims(*dims)) return e;
    param_layout(*dims, *out);
    return 0;
}

extern "C" size_t refil_learner_workspace_bytes(const refil_dims* dims) {
    if (!dims || check_dims(*dims)) return 0;
    return workspace_bytes(*dims, CARVE_LEARNER);
}

static int learner_forward_backward(const refil_dims* dims, const refil_batch* batch, const float* params_live,
                                    const float* params_target, float* grads, void* workspace,
                                    size_t workspace_bytes_, const refil_debug_out* debug, void* stream, SideStream*& sd) {
    Ctx c;
    if (int e = make_ctx(c, dims, batch, workspace, workspace_bytes_, CARVE_LEARNER, stream)) return e;
    REFIL_CHECK(params_live && params_target && grads, "refil_learner_forward_backward: null parameter/grad buffer");
    REFIL_CHECK(dims->T1 >= 2, "refil_learner_forward_backward: need at least one transition (T1 >= 2)");
    REFIL_CHECK(batch->obs_mask && batch->actions && batch->avail_actions && batch->reward && batch->terminated && batch->filled,
                "refil_learner_forward_backward: incomplete batch");
    REFIL_CHECK(dims->gt_factors >= 0 && dims->gt_factors <= 2, "refil: gt_factors must be 0, 1 or 2");
    REFIL_CHECK(batch->t_limit == 0 || (batch->t_limit >= 2 && batch->t_limit <= dims->T1), "refil_learner_forward_backward: t_limit %d outside [2, T1 = %d] (0 = none)",
                batch->t_limit, dims->T1);
    REFIL_CHECK(!dims->imagine || batch->group_bits || dims->gt_factors == 1, "refil_learner_forward_backward: group_bits required when imagine=1");
    REFIL_CHECK(!(dims->gt_factors || dims->gt_obs_mask) || batch->gt_mask, "refil_learner_forward_backward: gt_mask required by gt_factors / gt_obs_mask");
    const refil_dims& d = c.d; const Sizes& s = c.s; const refil_param_layout& L = c.L; Work& w = c.w;
    const int T = d.T1 - 1, G = s.G, nv0 = s.nv0, H = d.H, h = d.hyp, M = d.M, dd = d.d;
    const long BT = (long)d.B * T;
    float* stats = grads + L.total;
    if (c.prev) {       // refil_batch.target_version: were the target parameters rewritten since this workspace's previous call?
        c.target_same = c.same_layout && batch->target_version != 0 && c.prev->target_version == batch->target_version &&
                        c.prev->target_ptr == (const void*)params_target;
        c.prev->target_version = batch->target_version; c.prev->target_ptr = params_target;
        // the composed out_trans o fc2 maps of the TARGET nets depend on params_target alone: kept from the previous call when it
        // built the same ones (a switch like REFIL_PRESUM may have changed between two calls of a test process)
        const int cs = (c.presum ? 1 : 0) | (c.compose_agent ? 2 : 0);
        c.skip_compose = c.target_same && c.prev->compose_state == cs;     // (applied to the target nets' forward only, below)
        c.prev->compose_state = cs;
    }
    const bool target_composed = c.skip_compose;
    c.skip_compose = false;
    REFIL_HIP(hipMemsetAsync(grads, 0, (L.total + REFIL_NSTAT) * sizeof(float), c.st));
    // (with a gradient hook the mixer's gradients must be complete when it fires: no deferral then)
    const char* defer_e = getenv("REFIL_DEFER_REDUCE");       // (read per call: tests compare the modes in one process)
    const int defer_env = defer_e ? atoi(defer_e) : 2;
    DeferredReduce deferred;
    deferred.pool = c.w.dpool; deferred.cap = dpool_floats(c.d); deferred.lo = grads; deferred.hi = grads + L.total;
    deferred.mode = defer_env;
    if (defer_env && !g_mixer_hook) c.defer = &deferred;

    // ---------------- forward ----------------
    // Streams (REFIL_NO_OVERLAP=1 / refil_set_overlap(0) serialise everything on the caller's stream):
    //   caller's stream  live agent (3 mask variants)          -> Q selection, mixing, loss -> agent backward
    //   sd->s            live + target hypernets                                            -> hypernet backward
    //   sd->g[0]         the agent nets' mask words, target agent;  in the backward: the agent chain's weight gradients
    //   sd->g[1]         the hypernets' mask words;                 in the backward: the hypernet chain's weight gradients
    // so that the latency-bound kernels (the recurrences: <= 128 workgroups) always have GEMM trains of other chains beside them.
    Ctx ca = c;                       // live agent chain: caller's stream
    Ctx ch = c;                       // hypernet chain: side stream + its own split-K scratch
    const bool overlap = overlap_enabled();
    ca.gpartial = w.partial; ch.gpartial = w.partial2; ca.tpartial = w.partial3; ch.tpartial = w.partial4;
    // Swept on one box (tools/sweep.sh, all 8 x 3 combinations): agent chain 7 + hypernet chain 2 = 2.014 ms against 2.050 with
    // every weight gradient on the weight-gradient streams (the agent settings alone lose 0.5 %: they only pay off once the
    // hypernet chain's stream takes its fc1 gradient and frees the other streams earlier)
    // (read per call: tests/test_gpu_learner.py compares the two placements in one process)
    const char* tail_ea = getenv("REFIL_TAIL_DW_A"); const char* tail_eh = getenv("REFIL_TAIL_DW_H");
    const int tail_a = tail_ea ? atoi(tail_ea) : 7, tail_h = tail_eh ? atoi(tail_eh) : 2;
    ca.tail_dw = tail_a; ch.tail_dw = tail_h;
    if (overlap) {
        RUN(side_stream(sd));
        sd->next_ev = 2;                                   // (pool[0..1]: the final joins of the two weight-gradient streams)
        ch.st = sd->s; ch.gst = sd->s; ca.sd = ch.sd = sd;
        static const bool gstreams = [] { const char* e = getenv("REFIL_GRADSTREAM"); return !(e && e[0] == '0'); }();
        if (gstreams) { ch.gst = sd->g[1]; ca.gst = sd->g[0]; }
        // (the target agent on a third stream was measured slower: three GEMM trains thrash each other more than the lone
        // recurrence at the end of the forward costs)
        static const bool mwside = [] { const char* e = getenv("REFIL_MW_SIDE"); return !(e && e[0] == '0'); }();
        // (under stream capture the mask words stay on the chains' own streams: on ROCm 7.2 hipStreamEndCapture crashes when the
        // weight-gradient streams also carried them -- tools/probes/graph_capture.py with REFIL_HIPGRAPH_MW=1 reproduces it, the
        // stand-alone fork / join / re-fork patterns of tools/probes/capture_refork.hip do not)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(c.st, &cap);
        static const bool mw_cap = [] { const char* e = getenv("REFIL_HIPGRAPH_MW"); return e && e[0] == '1'; }();
        if (mwside && (cap == hipStreamCaptureStatusNone || mw_cap)) { ca.mwst = sd->g[0]; ch.mwst = sd->g[1]; }
    }
    // Early prologue (refil_batch.ready_event): the input assembly and the row lists depend on the batch fields alone. With the
    // caller's "fields are ready" event they are enqueued on the hypernet chain's stream, i.e. behind the previous step's
    // hypernet backward -- which ends before the agent chain's tail does -- and write the prologue slot the previous step does
    // not read; the slot's last reader (the step before the previous one) is awaited through its event. The caller's stream
    // joins in front of the first projection. Only on an unchanged layout (a moved carve, or another entry point in between,
    // may overlay anything) and outside stream capture. (The list kernel takes its four-launch form there: the one-launch form
    // spins on a grid-wide exchange, and beside a saturated GPU its workgroups do not become resident together.)
    bool early = false;
    {
        static const bool early_env = [] { const char* e = getenv("REFIL_EARLY"); return !(e && e[0] == '0'); }();
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(c.st, &cap);
        early = early_env && overlap && c.lists && batch->ready_event && c.same_layout && cap == hipStreamCaptureStatusNone;
    }
    hipStream_t P = c.st;                                  // stream of the prologue
    if (early) {
        // REFIL_EARLY_ON: 1 (default) = the hypernet chain's stream, 2 / 3 = a weight-gradient stream
        static const int on = [] { const char* e = getenv("REFIL_EARLY_ON"); return e ? atoi(e) : 1; }();
        P = on == 2 ? sd->g[1] : (on == 3 ? sd->g[0] : sd->s);
        REFIL_HIP(hipStreamWaitEvent(P, (hipEvent_t)batch->ready_event, 0));
        REFIL_HIP(hipStreamWaitEvent(P, c.slot_free[c.slot], 0));
    }
    if (!c.lists) RUN(run_prep(c, 1, 3));
    else {
        // Row lists (one launch; its kernel also writes the contiguous mask copies). The input assembly does not depend on
        // them (all rows: cheaper than waiting for the lists).
        // (A/B on one box: the input assembly on the caller's stream, in front of the list kernel, beats the side stream by
        // 0.3 % at cfg-T and 1.6 % at cfg2 -- a cross-stream join at the head of the step costs more than the ~30 us of overlap)
        static const bool prep_side = [] { const char* e = getenv("REFIL_PREP_SIDE"); return e && e[0] == '1'; }();
        const hipStream_t ps = (overlap && prep_side && !early) ? sd->g[0] : P;
        if (ps != P) RUN(stream_after(sd, c.st, ps));
        {
            Ctx cp = c;
            cp.st = ps; cp.lists = false;
            RUN(run_prep(cp, 1, 2));
        }
        ListArgs la;
        memset(&la, 0, sizeof(la));
        la.b = c.b; la.B = d.B; la.T1 = d.T1; la.ne = d.ne; la.na = d.na; la.learner = 1; la.use_gt_obs = d.gt_obs_mask;
        la.emc = w.emc; la.em0 = w.em0; la.amask = w.amask; la.actf = w.actf;
        la.t_last = w.t_last; la.kdead_a = w.kdead_a; la.kdead_h = w.kdead_h;
        la.cnt = w.lcnt; la.off = w.loff; la.list_ea = w.list_ea; la.list_eh = w.list_eh; la.list_a = w.list_a; la.counts = w.counts;
        la.ever = w.ever; la.list_t = w.list_t;
        la.rep[0] = ListArgs::Rep{w.list_t3, 1, s.G, (int)(s.G * s.NA)};
        la.rep[1] = ListArgs::Rep{w.list_h, 0, s.nv0, (int)(s.NV * s.NA)};
        la.rep[2] = ListArgs::Rep{w.list_ht, 0, 1, (int)(s.nets * s.NA)};
        la.hint_out = row_hints_dev();
        if (la.hint_out) { la.err_out = la.hint_out + 8; la.err_host = row_hints() + 8; }
        la.sync = early ? nullptr : w.lsync;
        RUN(lists_launch(la, P));
        if (ps != P) RUN(stream_after(sd, ps, c.st));      // (the chains fork from c.st: they start with the inputs assembled)
    }
    if (early) {                                           // join: the caller's stream continues behind the prologue
        REFIL_HIP(hipEventRecord(c.pre_done, P));
        REFIL_HIP(hipStreamWaitEvent(c.st, c.pre_done, 0));
    }
    // Early target forward. The target mixer's hypernets and the target agent depend on the batch fields and on params_target
    // alone -- not on what the previous step's optimiser is still writing, not on this step's partition draw (they run under the
    // plain observability / entity masks: variant-0 words, built here without the partition bits). With an early prologue and
    // unchanged target parameters (refil_batch.target_version) their forward is enqueued right behind the prologue, on the same
    // stream: it runs beside the under-filled END of the previous step (its last weight gradients, the reductions, the
    // optimiser) instead of widening this step's forward -- 22 % of the step's dense FLOPs leave the two chains between the
    // start of the step and their join. Same kernels on the same data: bit-identical to the in-order schedule
    // (tests/test_gpu_early.py). The outputs (Work::ta / th) were last read at the previous step's join, which precedes that
    // step's hypernet backward on this stream. REFIL_EARLY_TARGET: bit 0 hypernets, bit 1 agent (default 1; 0 = off). Measured
    // (interleaved A/B on one box, tools/sweep.sh): hypernets early cfg-T 1.702 -> 1.675 ms, cfg2 0.757 -> 0.746; the agent early
    // as well LOSES (cfg-T 1.797, cfg2 0.809, cfg4 1.481 -> 1.58): it gives up the two-net projection launches and the shared
    // recurrence launch, and its nine launches sit in front of the live hypernets on the same stream.
    const bool hypernets = !d.mixer_vdn && !d.mixer_none;
    const char* et_e = getenv("REFIL_EARLY_TARGET");      // (read per call: tests/test_gpu_early.py runs both settings in one process)
    const int et_env = et_e ? atoi(et_e) : 1;
    const bool et_ok = early && c.target_same && c.mwords && !debug && P == sd->s;
    const bool et_h = et_ok && (et_env & 1) && hypernets && !d.pooling && attn_mfma_supported(d.ne, d.na, d.hyp / d.heads);
    const bool et_a = et_ok && (et_env & 2) && !d.agent_ff && c.compose_agent && !d.pooling && (G * d.B * d.na) % 16 == 0 &&
                      attn_mfma_supported(d.ne, d.na, d.d / d.heads);
    ++g_stat_steps; g_stat_early += early; g_stat_et_h += et_h; g_stat_et_a += et_a;
    hipEvent_t et_done = nullptr;
    if (et_h || et_a) {
        Ctx ce = c;                                        // everything on the prologue's stream, in order
        ce.st = P; ce.gst = P; ce.mwst = P; ce.sd = sd; ce.defer = nullptr; ce.mw_nvar = 1;
        ce.w.mw_a = w.mw_at; ce.w.rb_a = w.rb_at; ce.w.mw_h = w.mw_ht; ce.w.rb_h = w.rb_ht;
        ce.skip_compose = target_composed;     // (same target parameters as the previous call on this arena: its composed maps are still there)
        for (int hyper = 0; hyper < 2; ++hyper) {
            if (!(hyper ? et_h : et_a)) continue;
            refil_attn_desc a = attn_base(c, hyper ? d.hyp : d.d);
            a.t_last = w.t_last; a.kv_dead = hyper ? w.kdead_h : w.kdead_a; a.q_dead = w.amask;
            a.nvar = 1; a.var[0] = hyper ? REFIL_MASK_ENTITY : REFIL_MASK_OBS;
            RUN(attn_mask_words_launch(a, hyper ? w.mw_ht : w.mw_at, hyper ? w.rb_ht : w.rb_at, P));
        }
        if (et_h) RUN(hyper_forward(ce, params_target, w.th, 1, HY_ALL, nullptr, true));
        if (et_a) RUN(agent_forward(ce, params_target, w.ta, 1, nullptr, AG_ALL, true));
        et_done = sd->pool[sd->next_ev];
        sd->next_ev = (sd->next_ev + 1) % 128;
        REFIL_HIP(hipEventRecord(et_done, P));
    }
    if (c.mwords) {
        // mask words of every row, once per step: agent nets (observability variants) and hypernets (entity variants);
        // they are needed by the first attention launches only, so they are built beside the first projections
        for (int hyper = 0; hyper < 2; ++hyper) {
            refil_attn_desc a = attn_base(c, hyper ? d.hyp : d.d);
            if (c.lists) { a.t_last = w.t_last; a.kv_dead = hyper ? w.kdead_h : w.kdead_a; a.q_dead = w.amask; }
            a.nvar = s.G;
            a.var[0] = hyper ? REFIL_MASK_ENTITY : REFIL_MASK_OBS;
            a.var[1] = group_code(d, 0, !hyper);
            a.var[2] = group_code(d, 1, !hyper);
            const hipStream_t ms = hyper ? ch.mwst : ca.mwst;
            RUN(stream_after(sd, c.st, ms));
            RUN(attn_mask_words_launch(a, hyper ? w.mw_h : w.mw_a, hyper ? w.rb_h : w.rb_a, ms));
        }
    }
    RUN(stream_after(sd, c.st, ch.st));                    // fork: inputs assembled
    if (hypernets) {
        // A/B on one box: with imagined copies (live net 0 under three mask variants) the merged launch LOSES 1.1 % (cfg-T) to
        // 3.5 % (cfg2) -- the projections of both mixers in front of it no longer interleave with an attention launch of the
        // other chain; without them (qmix_atten: two symmetric light launches) it wins 2.5 % (cfg4). REFIL_HYPER_MERGE=0/1 forces it
        static const int hy_env = [] { const char* e = getenv("REFIL_HYPER_MERGE"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
        // (one merged launch is fused for both mixers or for neither)
        const bool hy_merge = !et_h && (hy_env >= 0 ? hy_env == 1 : nv0 == 1) && ((c.qkv & QKV_T_HYPER) != 0) == ((c.qkv & QKV_L_HYPER) != 0);
        if (hy_merge && 2 * s.nets <= 8 && !d.pooling && attn_mfma_supported(d.ne, d.na, d.hyp / d.heads)) {
            // live and target mixers' hypernets: projections of both, then ONE attention launch for all eight nets, then the tails
            RUN(hyper_forward(ch, params_live, w.lh, nv0, HY_PRE));
            Ctx cht = ch; cht.skip_compose = target_composed;
            RUN(hyper_forward(cht, params_target, w.th, 1, HY_PRE, nullptr, true));
            RUN(hyper_forward(ch, params_live, w.lh, nv0, HY_ATTN, &w.th, false, params_target));
            RUN(hyper_forward(ch, params_live, w.lh, nv0, HY_POST));
            RUN(hyper_forward(cht, params_target, w.th, 1, HY_POST, nullptr, true));
        } else {
        // the target mixer's hypernets first: A/B on one box -0.2 % (cfg-T) .. -0.7 % (cfg3, cfg5) against live-first -- the live
        // mixer's heavier attention launch (three mask variants) then runs beside the recurrence instead of the agents' GEMMs
        static const bool live_first = [] { const char* e = getenv("REFIL_HYPER_ORDER"); return e && e[0] == '0'; }();
        if (live_first) RUN(hyper_forward(ch, params_live, w.lh, nv0));
        if (!et_h) { Ctx cht = ch; cht.skip_compose = target_composed; RUN(hyper_forward(cht, params_target, w.th, 1, HY_ALL, nullptr, true)); }   // target mixer hypernets (et_h: already enqueued)
        if (!live_first) RUN(hyper_forward(ch, params_live, w.lh, nv0));          // live mixer hypernets
        }
    }
    if (overlap) REFIL_HIP(hipEventRecord(sd->ev[1], sd->s));
    // The join of the two chains. Opt-in (REFIL_JOIN_FUSED, bit 0 / bit 1): fc3 live + fc3 target + Q selection as ONE launch
    // per (b,t) row (qhead_kernel), and the Q head's backward as the epilogue of the mixing kernel -- five dependent launches
    // become two. Measured (one box, 3 interleaved rounds): cfg-T 1.863 ms off / 1.872 (bit 0) / 1.868 (bit 1) / 1.877 (both),
    // cfg3 3.236 / 3.253 / 3.248 / 3.272, cfg2 0.948 / 0.943 / 0.951 / 0.945, cfg5 2.794 / 2.792 / 2.755 / 2.757: at the join
    // the HYPERNET chain is the late one (it ends ~80 us after the recurrence), so shortening the agent chain's tail buys
    // nothing at the north-star shape, and whatever lengthens the mixing kernel delays the hypernets' backward behind it.
    static const bool fused_env = [] { const char* e = getenv("REFIL_MIX_FUSED"); return !(e && e[0] == '0'); }();
    const char* join_e = getenv("REFIL_JOIN_FUSED");        // (read per call: a test compares the paths in one process)
    const int join_env = join_e ? atoi(join_e) : 0;     // bit 0: qhead_kernel, bit 1: Q-head backward as the mixing kernel's epilogue
    const bool mix_fused = fused_env && hypernets && !d.mixer_lin;
    QHeadArgs qh;
    memset(&qh, 0, sizeof(qh));
    qh.hs = w.la.hsx; qh.ths = w.ta.hsx;
    qh.w3 = params_live + L.ag_fc3_w; qh.b3 = params_live + L.ag_fc3_b; qh.tw3 = params_target + L.ag_fc3_w; qh.tb3 = params_target + L.ag_fc3_b;
    qh.amask = w.amask; qh.actions = c.b.actions; qh.ac_sB = c.b.ac_sB; qh.ac_sT = c.b.ac_sT;
    qh.avail = c.b.avail_actions; qh.av_sB = c.b.av_sB; qh.av_sT = c.b.av_sT;
    qh.t_last = c.lists ? w.t_last : nullptr;
    qh.chosen = w.chosen; qh.tmax = w.tmax; qh.G = G; qh.B = d.B; qh.T1 = d.T1; qh.na = d.na; qh.A = d.A; qh.H = H; qh.double_q = d.double_q;
    const bool qhead_fused = (join_env & 1) && mix_fused && !d.agent_ff && qhead_eligible(qh);
    const bool qbwd_fused = (join_env & 2) && mix_fused && !d.agent_ff && H % 4 == 0;
    if (!d.agent_ff && (G * d.B * d.na) % 16 == 0) {
        // live (q_learner.py:86-89 / 107) and target (:111-113) agents on one stream: their recurrences share one launch
        static const bool dual_env = [] { const char* e = getenv("REFIL_AGENT_DUAL"); return !(e && e[0] == '0'); }();
        const bool dual = !et_a && dual_env && c.lists && c.compose_agent && !d.pooling && attn_mfma_supported(d.ne, d.na, d.d / d.heads) &&
                          (params_target - params_live) % 4 == 0;
        if (c.compose_agent) {
            // The composed fc2 o out_trans maps depend on the parameters alone: built at the HEAD of the chain (a 6 us kernel that
            // took 10-60 us between the attention core and the recurrence once the other chain's GEMMs shared the GPU); the
            // target agent's are kept while params_target is unchanged
            for (int n = 0; n < 2; ++n) {
                if (n == 1 && (et_a || target_composed)) continue;
                const float* Pn = n ? params_target : params_live; const AgentBufs& bn = n ? w.ta : w.la;
                ComposeArgs cg;
                memset(&cg, 0, sizeof(cg));
                cg.W2 = Pn + L.ag_fc2_w; cg.b2 = Pn + L.ag_fc2_b; cg.Wo = Pn + L.ag_out_w; cg.bo = Pn + L.ag_out_b;
                cg.Wc = bn.wc; cg.bc = bn.bc; cg.bd = bn.bd; cg.nets = 1; cg.M = H; cg.h = dd;
                RUN(compose_forward_launch(cg, ca.st));
            }
            ca.agent_composed = true;
        }
        if (dual) RUN(agent_entity_dual(ca, params_live, params_target, w.la, w.ta, G));
        RUN(agent_forward(ca, params_live, w.la, G, nullptr, AG_PRE | (dual ? AG_NO_ENTITY : 0)));
        if (!et_a) RUN(agent_forward(ca, params_target, w.ta, 1, nullptr, AG_PRE | (dual ? AG_NO_ENTITY : 0), true));
        const refil_gru_desc gl = agent_gru_desc(ca, params_live, w.la, G, true), gt = agent_gru_desc(ca, params_target, w.ta, 1, true);
        RUN(gru_forward_launch2(gl, et_a ? nullptr : &gt, ca.st));
        if (!qhead_fused) {
            RUN(agent_forward(ca, params_live, w.la, G, nullptr, AG_POST));
            if (!et_a) RUN(agent_forward(ca, params_target, w.ta, 1, nullptr, AG_POST, true));
        }
    } else {
        RUN(agent_forward(ca, params_live, w.la, G, nullptr, qhead_fused ? (AG_PRE | AG_GRU) : AG_ALL));
        RUN(agent_forward(ca, params_target, w.ta, 1, nullptr, qhead_fused ? (AG_PRE | AG_GRU) : AG_ALL, true));
    }
    if (et_done) REFIL_HIP(hipStreamWaitEvent(c.st, et_done, 0));      // the early target forward's outputs (agent Q values; hypernets: ev[1] below)
    if (qhead_fused) {
        // the join of the two chains: Q head (fc3), inactive-agent fill, chosen-action gather and double-Q target selection
        // in ONE launch per (b,t) row straight from the hidden states (two thin GEMMs over all rows + a gather kernel before)
        qh.q_out = debug ? w.la.qv : nullptr;
        RUN(qhead_launch(qh, ca.st));
    } else {
        QSelArgs q;
        q.q = w.la.qv; q.tq = w.ta.qv; q.actions = c.b.actions; q.ac_sB = c.b.ac_sB; q.ac_sT = c.b.ac_sT;
        q.avail = c.b.avail_actions; q.av_sB = c.b.av_sB; q.av_sT = c.b.av_sT;
        q.chosen = w.chosen; q.tmax = w.tmax; q.G = G; q.B = d.B; q.T1 = d.T1; q.na = d.na; q.A = d.A; q.double_q = d.double_q;
        RUN(qselect_launch(q, ca.st));                                            // :91-96,115-128
    }
    if (overlap) REFIL_HIP(hipStreamWaitEvent(c.st, sd->ev[1], 0));               // join: mixing needs the hypernet outputs
    MixArgs ml = mix_args(c, w.lh, nv0, w.chosen, G, 0, T);
    ml.q_tot = w.q_tot; ml.q_tot_im = w.q_tot_im;
    ml.ingroup_rows = (d.mixer_lin && d.imagine) ? w.ingroup : nullptr;
    MixArgs mt = mix_args(c, w.th, 1, w.tmax, 1, 1, T);
    mt.q_tot = w.tq_tot; mt.q_tot_im = nullptr;
    // gradient targets of the live mix (the backward, q_learner.py:176, is hand-scheduled)
    ml.gc_real = w.gc_real; ml.gc_im = w.gc_im;
    ml.dx_w1 = w.dx3h;
    if (d.mixer_vdn) {
        // nothing flows to hypernets
    } else if (d.mixer_lin) {
        ml.dx_v = w.dx3h + (long)(nv0 + 0) * s.NA * M;
    } else {
        ml.dx_wf = w.dx3h + (long)(nv0 + 0) * s.NA * M;
        ml.dx_b1 = w.dx3h + (long)(nv0 + 1) * s.NA * M;
        ml.dx_v = w.dx3h + (long)(nv0 + 2) * s.NA * M;
    }
    ml.dqs = w.dchosen;
    TdArgs t;
    t.q_tot = w.q_tot; t.q_tot_im = w.q_tot_im; t.tq_tot = w.tq_tot;
    t.nq = 1;
    if (d.mixer_none) {     // :131 `if self.mixer is not None` not taken: the per-agent values enter the loss directly
        t.q_tot = w.chosen; t.q_tot_im = nullptr; t.tq_tot = w.tmax; t.nq = d.na;
    }
    t.reward = c.b.reward; t.rw_sB = c.b.rw_sB; t.rw_sT = c.b.rw_sT;
    t.terminated = c.b.terminated; t.tm_sB = c.b.tm_sB; t.tm_sT = c.b.tm_sT;
    t.filled = c.b.filled; t.fl_sB = c.b.fl_sB; t.fl_sT = c.b.fl_sT;
    t.gc_real = d.mixer_none ? w.dchosen : w.gc_real; t.gc_im = w.gc_im; t.targets = w.targets; t.stats = stats;
    t.t_last = c.lists ? w.t_last : nullptr;
    t.ingroup_rows = ml.ingroup_rows;
    t.B = d.B; t.T = T; t.imagine = d.imagine; t.gamma = d.gamma; t.lmbda = d.lmbda;
    t.t_limit = c.b.t_limit;
    // FlexQMixer: live mix, target mix, TD error and the live mix's backward in one launch (REFIL_MIX_FUSED=0: four launches)
    QHeadBwd qb;
    memset(&qb, 0, sizeof(qb));
    if (qbwd_fused) {
        qb.dq = w.dqva; qb.dhs = w.dhs; qb.w3 = params_live + L.ag_fc3_w; qb.actions = c.b.actions; qb.ac_sB = c.b.ac_sB; qb.ac_sT = c.b.ac_sT;
        qb.ever = (c.lists && c.compose_agent) ? w.ever : nullptr; qb.A = d.A; qb.H = H;
    }
    if (mix_fused) {
        RUN(mix_train_launch(ml, mt, t, w.row_stats, c.st, qbwd_fused ? &qb : nullptr));   // :134-172 + backward of the mix (+ of the Q head)
    } else {
        if (!d.mixer_none) {
            RUN(mix_forward_launch(ml, c.st));                                    // :134-152
            RUN(mix_forward_launch(mt, c.st));                                    // :154
        }
        RUN(td_loss_launch(t, c.st));                                             // :157-172
    }
    if (debug) {
        RUN(copy_out(debug->q, w.la.qv, (long)G * s.NA * d.A, c.st));
        RUN(copy_out(debug->chosen_q, w.chosen, (long)G * BT * d.na, c.st));
        RUN(copy_out(debug->target_max_q, w.tmax, BT * d.na, c.st));
        if (!d.mixer_none) {          // (mixer=None: chosen_q / target_max_q ARE the values the loss is taken on)
            RUN(copy_out(debug->q_tot, w.q_tot, BT, c.st));
            if (d.imagine) RUN(copy_out(debug->q_tot_imagine, w.q_tot_im, BT, c.st));
            RUN(copy_out(debug->target_q_tot, w.tq_tot, BT, c.st));
            RUN(copy_out(debug->targets, w.targets, BT, c.st));
        }
    }

    // ---------------- backward (q_learner.py:176, hand-scheduled) ----------------
    if (mix_fused) {
        // the stat sums leave the critical path: folded on the hypernet chain's weight-gradient stream (joined before the
        // optimiser; the data-parallel hook hands that stream to the all-reduce)
        RUN(stream_after(sd, c.st, ch.gst));
        RUN(td_stats_launch(w.row_stats, (int)BT, stats, ch.gst));
    } else if (!d.mixer_none) {
        RUN(mix_backward_launch(ml, c.st));
    }
    if (overlap) {                                                                 // fork: agent backward chain
        REFIL_HIP(hipEventRecord(sd->ev[2], c.st));
        REFIL_HIP(hipStreamWaitEvent(sd->s, sd->ev[2], 0));
    }
    // (the hypernet chain -- the critical path -- is enqueued first, on the side stream)
    ComposeArgs hyp_compose;
    if (hypernets) {
    if (c.presum) {
        // composed tails (x3 = mask(a W_c^T + b_c)): G_c = g3^T a, g_c = colsum(g3) (weighted by n_act on the summed rows),
        // d(attention output) = g3 W_c straight into daoh; compose_backward turns (G_c, g_c) into the four parameter gradients
        refil_gemm_desc gw = linear_dw(w.dx3h, M, w.lh.ao, h, w.gwc, h, w.gbc, (long)nv0 * s.NA, M, h, ch.w.partial, 1);
        RUN(launch_dw(ch, with_rows(gw, ch, rows_h(ch, nv0))));
        refil_gemm_desc gx = linear_dx(w.dx3h, M, w.lh.wc, h, w.daoh, h, (long)nv0 * s.NA, M, h, 0);
        RUN(gemm_launch(with_rows(gx, ch, rows_h(ch, nv0)), ch.st));
        const long o3 = (long)nv0 * s.NA * M, oh = (long)nv0 * s.NA * h;
        refil_gemm_desc gw1 = linear_dw(w.dx3h + o3, M, w.lh.ao + oh, h, w.gwc + (long)M * h, h, nullptr, s.R, M, h, ch.w.partial, s.nets - 1);
        gw1.sA = s.NA * M; gw1.sB = s.NA * h; gw1.sC = (long)M * h;
        RUN(launch_dw(ch, gw1));
        refil_gemm_desc gb1 = linear_dw(w.dx3h + o3, M, w.nact, 1, w.gbc + M, 1, nullptr, s.R, M, 1, ch.w.partial, s.nets - 1);
        gb1.sA = s.NA * M; gb1.sB = 0; gb1.sC = M;
        RUN(launch_dw(ch, gb1));
        refil_gemm_desc gx1 = linear_dx(w.dx3h + o3, M, w.lh.wc + (long)M * h, h, w.daoh + oh, h, s.R, M, h, 0);
        gx1.batch = s.nets - 1; gx1.sA = s.NA * M; gx1.sB = (long)M * h; gx1.sC = s.NA * h;
        RUN(gemm_launch(gx1, ch.st));
        ComposeArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.W2 = params_live + L.mix_fc2_w; ca.sW2 = L.mix_fc2_w_stride; ca.b2 = params_live + L.mix_fc2_b; ca.sb2 = L.mix_fc2_b_stride;
        ca.Wo = params_live + L.mix_out_w; ca.sWo = L.mix_out_w_stride; ca.bo = params_live + L.mix_out_b; ca.sbo = L.mix_out_b_stride;
        ca.Gc = w.gwc; ca.gc = w.gbc;
        ca.dW2 = grads + L.mix_fc2_w; ca.db2 = grads + L.mix_fc2_b; ca.dWo = grads + L.mix_out_w; ca.dbo = grads + L.mix_out_b;
        ca.nets = s.nets; ca.M = M; ca.h = h;
        hyp_compose = ca;
        if (!(ch.tail_dw & 4)) RUN(compose_backward_launch(ca, ch.gst));          // (consumes G_c / g_c: stays behind them on the weight-gradient stream)
    } else {
    // hypernet tails: fc2 (flex_qmix.py:49)
    for (int part = 0; part < 2; ++part) {
        const long rows = part == 0 ? nv0 * s.NA : s.NA;
        const int batch = part == 0 ? 1 : s.nets - 1;
        const long voff = part == 0 ? 0 : nv0;
        const int net0 = part == 0 ? 0 : 1;
        refil_gemm_desc gw = linear_dw(w.dx3h + voff * s.NA * M, M, w.lh.x2 + voff * s.NA * h, h,
                                       grads + L.mix_fc2_w + net0 * L.mix_fc2_w_stride, h,
                                       grads + L.mix_fc2_b + net0 * L.mix_fc2_b_stride, rows, M, h, ch.w.partial, batch);
        gw.sA = s.NA * M; gw.sB = s.NA * h; gw.sC = L.mix_fc2_w_stride; gw.sColsum = L.mix_fc2_b_stride;
        RUN(launch_dw(ch, gw));
        refil_gemm_desc gx = linear_dx(w.dx3h + voff * s.NA * M, M, params_live + L.mix_fc2_w + net0 * L.mix_fc2_w_stride, h,
                                       w.dx2h + voff * s.NA * h, h, rows, M, h, 0);
        gx.batch = batch; gx.sA = s.NA * M; gx.sB = L.mix_fc2_w_stride; gx.sC = s.NA * h;
        gx.rowmask = w.amask; gx.rowmask_mod = (int)s.NA;
        RUN(gemm_launch(gx, ch.st));
    }
    }
    {
        AttnBlockBwd k;
        k.w = h; k.nets = s.nets; k.nv0 = nv0; k.P = params_live; k.Gr = grads; k.presum = c.presum; k.hyper = true;
        k.in_w = L.mix_in_w; k.in_w_stride = L.mix_in_w_stride; k.out_w = L.mix_out_w; k.out_w_stride = L.mix_out_w_stride;
        k.out_b = L.mix_out_b; k.out_b_stride = L.mix_out_b_stride;
        k.x1 = w.lh.x1; k.kv = w.lh.kv; k.q = w.lh.q; k.ao = w.lh.ao; k.dx2 = w.dx2h;
        k.dao = w.daoh; k.dq = w.dqh; k.dkv = w.dkvh; k.dx1 = w.dx1h;
        k.var_first[0] = REFIL_MASK_ENTITY;
        k.var_first[1] = group_code(d, 0, false);
        k.var_first[2] = group_code(d, 1, false);
        k.var_rest = REFIL_MASK_ENTITY;
        RUN(attn_block_backward(ch, k));
        if (c.presum && (ch.tail_dw & 4)) RUN(compose_backward_launch(hyp_compose, ch.gst));
        // the four hypernet fc1 layers: dW = dx1^T xe (one [4h,E] GEMM), db = colsum(dx1)
        refil_gemm_desc g = linear_dw(w.dx1h, s.nets * h, w.xe, s.Ep, grads + L.mix_fc1_w, s.E, grads + L.mix_fc1_b, s.NE, s.nets * h, s.E,
                                      ch.w.partial, 1);
        RUN(launch_dw_tail(ch, with_rows(g, ch, rows_eh(ch)), 2));
    }
    }
    // every kernel that writes a mixer gradient (and, earlier, the stat sums) is enqueued: on ch.gst they are complete
    if (g_mixer_hook) {
        RUN(stream_after(sd, ch.st, ch.gst));
        g_mixer_hook(g_mixer_hook_user, (void*)ch.gst);
    }
    // agent: chosen-Q gather + inactive-agent fill, then fc3
    {
        QSelBwdArgs q;
        q.dchosen = w.dchosen; q.actions = c.b.actions; q.ac_sB = c.b.ac_sB; q.ac_sT = c.b.ac_sT; q.amask = w.amask;
        q.dq = w.dqva; q.G = G; q.B = d.B; q.T1 = d.T1; q.na = d.na; q.A = d.A;
        q.w3 = nullptr; q.dhs = nullptr; q.H = 0; q.ever = nullptr;
        if (!d.agent_ff) { q.w3 = params_live + L.ag_fc3_w; q.dhs = w.dhs; q.H = H; }     // recurrent agent: + d(hidden) = dq W3
        if (!d.agent_ff && c.lists && c.compose_agent) q.ever = w.ever;     // (the recurrence and the fc3 gradient read list_t rows only)
        if (!qbwd_fused) RUN(qselect_bwd_launch(q, ca.st));                // (fused: written by the mixing kernel's epilogue)
        const long rows = (long)G * s.NA;
        ComposeArgs ag_compose;
        bool ag_compose_pending = false;
        if (d.agent_ff) {
            // feed-forward agent: q = fc2(x2), x2 = relu(masked out_trans)  (entity_ff_agent.py:44-52)
            RUN(launch_dw(ca, linear_dw(w.dqva, d.A, w.la.x2, dd, grads + L.ag_fc2_w, dd, grads + L.ag_fc2_b, rows, d.A, dd, ca.w.partial, 1)));
            refil_gemm_desc gx2 = linear_dx(w.dqva, d.A, params_live + L.ag_fc2_w, dd, w.dx2a, dd, rows, d.A, dd, REFIL_GEMM_RELU_BWD);
            gx2.aux = w.la.x2;                 // x2 = 0 on inactive rows, so relu' also applies the row mask
            RUN(gemm_launch(gx2, ca.st));
        } else {
            refil_gemm_desc gw = linear_dw(w.dqva, d.A, w.la.hsx, H, grads + L.ag_fc3_w, H, grads + L.ag_fc3_b, rows, d.A, H, ca.w.partial, 1);
            gw.b_map = hs_rows(c, d.na);
            RUN(launch_dw(ca, with_rows(gw, ca, rows_t(ca, G))));
            // BPTT
            refil_gru_desc g;
            memset(&g, 0, sizeof(g));
            g.hsx = w.la.hsx; g.w_hh = params_live + L.ag_w_hh; g.b_hh = params_live + L.ag_b_hh;
            g.save_r = w.la.sr; g.save_z = w.la.sz; g.save_n = w.la.sn; g.save_ghn = w.la.sg;
            g.dhs = w.dhs; g.dgi = w.dgi; g.dgh = w.dgh; g.NR = G * d.B * d.na; g.T1 = d.T1; g.na = d.na; g.H = H;
            if (c.lists) { g.t_last = w.t_last; g.B = d.B; }
            if (c.lists && c.compose_agent) g.ever = w.ever;
            RUN(gru_backward_launch(g, ca.st));
            // dW_hh = d(gh)^T h_{t-1}: the recurrence stores d(gh) as [dgi_r, dgi_z | dgh_n] (its r / z blocks ARE dgi's), so the
            // gradient is two launches: rows [0, 2H) from dgi's first 2H columns, rows [2H, 3H) from the n block
            refil_gemm_desc ghh = linear_dw(w.dgi, 3 * H, w.la.hsx, H, grads + L.ag_w_hh, H, grads + L.ag_b_hh, rows, 2 * H, H, ca.w.partial, 1);
            ghh.b_map = hs_rows(c, 0);
            refil_gemm_desc ghn = linear_dw(w.dgh, H, w.la.hsx, H, grads + L.ag_w_hh + 2L * H * H, H, grads + L.ag_b_hh + 2 * H, rows, H, H, ca.w.partial, 1);
            ghn.b_map = hs_rows(c, 0);
            const bool tl = c.compose_agent;           // (the listed rows are the ones the forward computed x3 on)
            RUN(launch_dw(ca, tl ? with_rows(ghh, ca, rows_t(ca, G)) : ghh));
            RUN(launch_dw(ca, tl ? with_rows(ghn, ca, rows_t(ca, G)) : ghn));
            refil_gemm_desc gih = linear_dw(w.dgi, 3 * H, w.la.x3, H, grads + L.ag_w_ih, H, grads + L.ag_b_ih, rows, 3 * H, H, ca.w.partial, 1);
            RUN(launch_dw(ca, tl ? with_rows(gih, ca, rows_t(ca, G)) : gih));
            refil_gemm_desc gx3 = linear_dx(w.dgi, 3 * H, params_live + L.ag_w_ih, H, w.dx3a, H, rows, 3 * H, H, REFIL_GEMM_RELU_BWD);
            gx3.aux = w.la.x3;
            RUN(gemm_launch(tl ? with_rows(gx3, ca, rows_t(ca, G)) : gx3, ca.st));
            if (c.compose_agent) {
                // composed fc2 o out_trans: G_c = dx3^T a (a = 0 on inactive rows), colsum over all rows -> db_2, over the
                // active rows -> the b_o terms; d(attention out) = dx3 W_c on the active rows
                RUN(launch_dw(ca, with_rows(linear_dw(w.dx3a, H, w.la.ao, dd, w.gwca, dd, w.gbca, rows, H, dd, ca.w.partial, 1), ca, rows_t(ca, G))));
                refil_gemm_desc gb = linear_dw(w.dx3a, H, w.actf, 1, w.gbact, 1, nullptr, rows, H, 1, ca.w.partial, 1);
                gb.b_map = refil_rowmap{(int)s.NA, 0, 0};          // the G mask copies share the [NA] activity vector
                RUN(launch_dw(ca, with_rows(gb, ca, rows_t(ca, G))));
                refil_gemm_desc gx2 = linear_dx(w.dx3a, H, w.la.wc, dd, w.daoa, dd, rows, H, dd, 0);
                gx2.rowmask = w.amask; gx2.rowmask_mod = (int)s.NA;
                RUN(gemm_launch(with_rows(gx2, ca, rows_t(ca, G)), ca.st));
                ComposeArgs cb;
                memset(&cb, 0, sizeof(cb));
                cb.W2 = params_live + L.ag_fc2_w; cb.b2 = params_live + L.ag_fc2_b; cb.Wo = params_live + L.ag_out_w; cb.bo = params_live + L.ag_out_b;
                cb.Gc = w.gwca; cb.gc = w.gbact; cb.gc_b2 = w.gbca;
                cb.dW2 = grads + L.ag_fc2_w; cb.db2 = grads + L.ag_fc2_b; cb.dWo = grads + L.ag_out_w; cb.dbo = grads + L.ag_out_b;
                cb.nets = 1; cb.M = H; cb.h = dd;
                ag_compose = cb; ag_compose_pending = (ca.tail_dw & 4) != 0;
                if (!ag_compose_pending) RUN(compose_backward_launch(cb, ca.gst));
            } else {
            // fc2
            RUN(launch_dw(ca, linear_dw(w.dx3a, H, w.la.x2, dd, grads + L.ag_fc2_w, dd, grads + L.ag_fc2_b, rows, H, dd, ca.w.partial, 1)));
            refil_gemm_desc gx2 = linear_dx(w.dx3a, H, params_live + L.ag_fc2_w, dd, w.dx2a, dd, rows, H, dd, 0);
            gx2.rowmask = w.amask; gx2.rowmask_mod = (int)s.NA;
            RUN(gemm_launch(gx2, ca.st));
            }
        }
        AttnBlockBwd k;
        k.w = dd; k.nets = 1; k.nv0 = G; k.P = params_live; k.Gr = grads; k.presum = c.compose_agent;   // (d(attn out) already in daoa)
        k.hyper = false;
        k.in_w = L.ag_in_w; k.in_w_stride = 0; k.out_w = L.ag_out_w; k.out_w_stride = 0; k.out_b = L.ag_out_b; k.out_b_stride = 0;
        k.x1 = w.la.x1; k.kv = w.la.kv; k.q = w.la.q; k.ao = w.la.ao; k.dx2 = w.dx2a;
        k.dao = w.daoa; k.dq = w.dqa; k.dkv = w.dkva; k.dx1 = w.dx1a;
        k.var_first[0] = REFIL_MASK_OBS;
        k.var_first[1] = group_code(d, 0, true);
        k.var_first[2] = group_code(d, 1, true);
        k.var_rest = REFIL_MASK_OBS;
        RUN(attn_block_backward(ca, k));
        if (ag_compose_pending) RUN(compose_backward_launch(ag_compose, ca.gst));
        RUN(launch_dw_tail(ca, with_rows(linear_dw(w.dx1a, dd, w.xe, s.Ep, grads + L.ag_fc1_w, s.E, grads + L.ag_fc1_b, s.NE, dd, s.E, ca.w.partial, 1), ca, rows_ea(ca)), 2));
    }
    if (deferred.n && deferred.mode == 2) {
        // every stream reduces what it produced, behind its last launch (the other streams' tails run beside it)
        bool done[64] = {};
        for (int i = 0; i < deferred.n; ++i) {
            if (done[i]) continue;
            ReduceK mine[64];
            int m = 0;
            for (int j = i; j < deferred.n; ++j)
                if (deferred.on[j] == deferred.on[i]) { mine[m++] = deferred.r[j]; done[j] = true; }
            RUN(reduce_multi_launch(mine, m, deferred.on[i]));
        }
        deferred.n = 0;
    }
    if (overlap) {                                                                 // join: hypernet chain, both weight-gradient streams
        REFIL_HIP(hipEventRecord(sd->ev[3], sd->s));
        REFIL_HIP(hipStreamWaitEvent(c.st, sd->ev[3], 0));
        for (int i = 0; i < 2; ++i) RUN(stream_after(sd, sd->g[i], c.st));
    }
    if (deferred.n) RUN(reduce_multi_launch(deferred.r, deferred.n, c.st));
    {   // this step's prologue slot has no reader left behind this point (not under capture: a captured record would be a graph
        // node, and the eager step that later waits on the event would wait on nothing)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(c.st, &cap);
        if (cap == hipStreamCaptureStatusNone) REFIL_HIP(hipEventRecord(c.slot_free[c.slot], c.st));
    }
    static const bool defer_log = [] { const char* e = getenv("REFIL_GEMM_LOG"); return e && e[0] == '1'; }();
    if (defer_log) fprintf(stderr, "refil: %d deferred reductions, %.1f of %.1f MiB of partials\n", deferred.n, deferred.used / 262144.0, dpool_floats(c.d) / 262144.0);
    return 0;
}

extern "C" int refil_learner_forward_backward(const refil_dims* dims, const refil_batch* batch, const float* params_live,
                                              const float* params_target, float* grads, void* workspace,
                                              size_t workspace_bytes_, const refil_debug_out* debug, void* stream) {
    SideStream* sd = nullptr;
    const int rc = learner_forward_backward(dims, batch, params_live, params_target, grads, workspace, workspace_bytes_, debug, stream, sd);
    if (rc) side_stream_rejoin(sd, (hipStream_t)stream);       // a launch failed after the fork: join before reporting
    return rc;
}

extern "C" int refil_learner_step(const refil_dims* dims, const refil_batch* batch, float* params_live, const float* params_target,
                                  float* grads, float* square_avg, const refil_opt_hyper* hyper, void* comm,
                                  void* workspace, size_t workspace_bytes_, void* scratch, void* stream) {
    REFIL_CHECK(dims && hyper && square_avg && scratch, "refil_learner_step: null argument");
    if (int e = refil_learner_forward_backward(dims, batch, params_live, params_target, grads, workspace, workspace_bytes_, nullptr, stream)) return e;
    refil_param_layout L;
    param_layout(*dims, L);
    if (comm) RUN(refil_allreduce_flat(grads, L.total + REFIL_NSTAT, comm, stream));
    return refil_clip_rmsprop_step(params_live, grads, square_avg, L.total, hyper->lr, hyper->alpha, hyper->eps, hyper->weight_decay,
                                   hyper->grad_norm_clip, grads + L.total, scratch, stream);
}

// Diagnostics of the row lists of the LAST learner step run on `workspace` with these dims (synchronises `stream`):
// out[0..5] = {row lists active (0/1), listed entity rows of the agent nets, of the hypernets, listed agent rows,
//              live (b,t) rows, B * T1}. Benchmarks report the live-row fractions next to the dense FLOP count.
extern "C" int refil_learner_row_counts(const refil_dims* dims, void* workspace, size_t workspace_bytes_, int32_t* out, void* stream) {
    REFIL_CHECK(dims && workspace && out, "refil_learner_row_counts: null argument");
    if (int e = check_dims(*dims)) return e;
    Ctx c;
    refil_batch dummy;
    memset(&dummy, 0, sizeof(dummy));
    dummy.entities = reinterpret_cast<const float*>(workspace); dummy.entity_mask = reinterpret_cast<const uint8_t*>(workspace);
    dummy.actions = reinterpret_cast<const int64_t*>(workspace);
    if (int e = make_ctx(c, dims, &dummy, workspace, workspace_bytes_, CARVE_LEARNER, stream)) return e;
    out[0] = c.lists ? 1 : 0; out[5] = (int32_t)c.s.R;
    out[1] = (int32_t)c.s.NE; out[2] = (int32_t)c.s.NE; out[3] = (int32_t)c.s.NA; out[4] = (int32_t)c.s.R;
    // make_ctx advanced the workspace's prologue slot as a learner step does (section 3a: the steps alternate between two copies of the
    // lists). This is a query: the LAST step's lists sit in the slot it came from, and the alternation is put back as it was. (Found in
    // round 6 by tests/test_gpu_learner.py::test_row_counts_match_the_batch: after one step the call returned the other slot's zeros;
    // bench.py never noticed -- it trains on one batch, both slots hold the same counts.)
    const int* counts = c.w.counts;
    if (c.prev && c.same_layout) {
        c.prev->slot ^= 1;
        Arena a2{(char*)workspace, workspace_bytes_, 0, false};
        Work w2;
        carve(a2, *dims, w2, CARVE_LEARNER);
        counts = c.prev->slot ? w2.alt.counts : w2.counts;
    }
    if (c.lists) {
        REFIL_HIP(hipMemcpyAsync(out + 1, counts, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c.st));
        REFIL_HIP(hipStreamSynchronize(c.st));
    }
    return 0;
}

// The calling thread's hypernet-chain stream on the current device (created lazily): the stream the early prologue of a
// learner step runs on. A producer of batches (ReplayBuffer.sample) enqueues its gather THERE -- behind the previous step's
// hypernet backward, in front of the next step's prologue -- instead of on a stream of its own (a fifth busy stream shares a
// hardware queue with one of the step's four and slowed the step by 7-13 %).
extern "C" int refil_side_stream(void** out) {
    REFIL_CHECK(out, "refil_side_stream: null out");
    SideStream* sd = nullptr;
    if (int e = side_stream(sd)) return e;
    *out = (void*)sd->s;
    return 0;
}

// Destroys this thread's side streams and events (all devices). Safe to call at any time: they are re-created lazily.
extern "C" int refil_release_streams(void) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (auto& kv : g_prev) {                               // per-arena events (created on kv.first.first)
        if (hipSetDevice(kv.first.first) != hipSuccess) { (void)hipGetLastError(); continue; }
        for (auto& e : kv.second.slot_free) if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); }
        if (kv.second.pre_done) { (void)hipEventSynchronize(kv.second.pre_done); (void)hipEventDestroy(kv.second.pre_done); }
    }
    g_prev.clear();
    for (int dev = 0; dev < MAX_DEVICES; ++dev) {
        SideStream& sd = g_side[dev];
        if (!sd.ok) continue;
        if (hipSetDevice(dev) == hipSuccess) {
            (void)hipStreamSynchronize(sd.s);
            for (auto& e : sd.ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
            for (auto& e : sd.pool) { if (e) (void)hipEventDestroy(e); e = nullptr; }
            for (auto& g : sd.g) { if (g) { (void)hipStreamSynchronize(g); (void)hipStreamDestroy(g); } g = nullptr; }
            (void)hipStreamDestroy(sd.s);
        }
        sd.s = nullptr; sd.ok = false;
    }
    (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return 0;
}

extern "C" int refil_agent_forward(const refil_dims* dims, const refil_batch* batch, int32_t first_step_zero,
                                   const float* params, const float* h0, float* h_out, float* q_out, void* workspace,
                                   size_t workspace_bytes_, void* stream) {
    Ctx c;
    if (int e = make_ctx(c, dims, batch, workspace, workspace_bytes_, CARVE_AGENT_FWD, stream)) return e;
    REFIL_CHECK(params && q_out, "refil_agent_forward: null params / q_out");
    REFIL_CHECK(batch->obs_mask, "refil_agent_forward: obs_mask missing");
    REFIL_CHECK(dims->gt_factors >= 0 && dims->gt_factors <= 2, "refil: gt_factors must be 0, 1 or 2");
    REFIL_CHECK(!dims->imagine || batch->group_bits || dims->gt_factors == 1, "refil_agent_forward: group_bits required when imagine=1");
    REFIL_CHECK(!(dims->gt_factors || dims->gt_obs_mask) || batch->gt_mask, "refil_agent_forward: gt_mask required by gt_factors / gt_obs_mask");
    RUN(run_prep(c, first_step_zero));
    RUN(agent_forward(c, params, c.w.la, c.s.G, h0));
    RUN(copy_out(q_out, c.w.la.qv, (long)c.s.G * c.s.NA * c.d.A, c.st));
    if (h_out && !c.d.agent_ff) RUN(get_hT_launch(c.w.la.hsx, h_out, c.s.G * c.d.B, c.d.T1, c.d.na, c.d.H, c.st));
    return 0;
}

extern "C" size_t refil_agent_workspace_bytes(const refil_dims* dims) {
    if (!dims || check_dims(*dims)) return 0;
    return workspace_bytes(*dims, CARVE_AGENT_FWD);
}
extern "C" size_t refil_mixer_workspace_bytes(const refil_dims* dims) {
    if (!dims || check_dims(*dims)) return 0;
    return workspace_bytes(*dims, CARVE_MIXER_FWD);
}

extern "C" int refil_mixer_forward(const refil_dims* dims, const refil_batch* batch, int32_t t0, int32_t T,
                                   const float* params, const float* agent_qs, const float* agent_qs_imagine,
                                   float* q_tot, float* q_tot_imagine, float* ingroup_sum, void* workspace,
                                   size_t workspace_bytes_, void* stream) {
    Ctx c;
    if (int e = make_ctx(c, dims, batch, workspace, workspace_bytes_, CARVE_MIXER_FWD, stream)) return e;
    REFIL_CHECK(!dims->mixer_none, "refil_mixer_forward: dims.mixer_none has no mixing network");
    REFIL_CHECK((params || dims->mixer_vdn) && agent_qs && q_tot, "refil_mixer_forward: null pointer");
    REFIL_CHECK(t0 >= 0 && T > 0 && t0 + T <= dims->T1, "refil_mixer_forward: step range [%d,%d) outside the batch", t0, t0 + T);
    const bool im = agent_qs_imagine != nullptr;
    REFIL_CHECK(dims->gt_factors >= 0 && dims->gt_factors <= 2, "refil: gt_factors must be 0, 1 or 2");
    REFIL_CHECK(!im || ((batch->group_bits || dims->gt_factors == 1 || batch->mask_words) && q_tot_imagine && dims->imagine),
                "refil_mixer_forward: imagined mix needs imagine=1, group_bits (or gt_factors, or explicit mask words) and q_tot_imagine");
    if (im && batch->mask_words) {
        // imagine_groups = (Wmask, Imask) tensors (flex_qmix.py:85-94), packed by the caller: the attention kernels read
        // the words instead of deriving the masks from the partition bits
        REFIL_CHECK(batch->mask_row_bits, "refil_mixer_forward: mask_words needs mask_row_bits");
        REFIL_CHECK(!dims->pooling && !dims->mixer_vdn && attn_mfma_supported(dims->ne, dims->na, dims->hyp / dims->heads),
                    "refil_mixer_forward: explicit mask tensors need attention hypernets of a shape the MFMA kernels take "
                    "(ne=%d na=%d head dim=%d)", dims->ne, dims->na, dims->hyp / dims->heads);
        c.mwords = true;
        c.w.mw_h = const_cast<unsigned long long*>(reinterpret_cast<const unsigned long long*>(batch->mask_words));
        c.w.rb_h = const_cast<unsigned long long*>(reinterpret_cast<const unsigned long long*>(batch->mask_row_bits));
    }
    REFIL_CHECK(!dims->gt_factors || batch->gt_mask, "refil_mixer_forward: gt_factors needs batch.gt_mask");
    RUN(run_prep(c, 1));
    const int nv0 = im ? 3 : 1;
    if (!dims->mixer_vdn) RUN(hyper_forward(c, params, c.w.lh, nv0));
    // chosen-Q layout expected by the mix kernel: [G][B,T,na]; the ABI hands real [B,T,na] and
    // imagined [B,T,2na] = cat(W, I) (q_learner.py:96): de-interleave into scratch (reuse lh.q)
    float* qs = c.w.chosen;
    const long BTn = (long)dims->B * T * dims->na;
    REFIL_HIP(hipMemcpyAsync(qs, agent_qs, BTn * sizeof(float), hipMemcpyDeviceToDevice, c.st));
    if (im) {
        REFIL_HIP(hipMemcpy2DAsync(qs + BTn, dims->na * sizeof(float), agent_qs_imagine, 2 * dims->na * sizeof(float),
                                   dims->na * sizeof(float), (size_t)dims->B * T, hipMemcpyDeviceToDevice, c.st));
        REFIL_HIP(hipMemcpy2DAsync(qs + 2 * BTn, dims->na * sizeof(float), agent_qs_imagine + dims->na, 2 * dims->na * sizeof(float),
                                   dims->na * sizeof(float), (size_t)dims->B * T, hipMemcpyDeviceToDevice, c.st));
    }
    MixArgs m = mix_args(c, c.w.lh, nv0, qs, im ? 3 : 1, t0, T);
    m.q_tot = q_tot; m.q_tot_im = q_tot_imagine;
    const bool want_ing = ingroup_sum && im && dims->mixer_lin;
    m.ingroup_rows = want_ing ? c.w.ingroup : nullptr;
    RUN(mix_forward_launch(m, c.st));
    if (want_ing) RUN(sum_launch(c.w.ingroup, (long)dims->B * T, ingroup_sum, c.st));
    return 0;
}
