// Internal launch API shared by the .hip translation units (not part of the C ABI).
#pragma once
#include "common.h"
#include "../../include/refil_hip.h"

namespace refil {

// which per-row inputs a pre-softmax mask variant reads (include/refil_hip.h: REFIL_MASK_*)
__host__ __device__ inline bool mask_uses_obs(int c) {
    return c <= REFIL_MASK_OBS_INTERACT || c == REFIL_MASK_OBS_GTW || c == REFIL_MASK_OBS_GTI || c == REFIL_MASK_OBS_RGTW || c == REFIL_MASK_OBS_RGTI;
}
__host__ __device__ inline bool mask_uses_gt(int c) { return c >= REFIL_MASK_OBS_GTW; }
__host__ __device__ inline bool mask_uses_groups(int c) {   // same(i,j): group bits + inactive-at-t0
    return c == REFIL_MASK_OBS_WITHIN || c == REFIL_MASK_OBS_INTERACT || c == REFIL_MASK_WITHIN || c == REFIL_MASK_INTERACT || c >= REFIL_MASK_OBS_RGTW;
}
__host__ __device__ inline bool mask_uses_inactive0(int c) { return mask_uses_groups(c) || c == REFIL_MASK_GTW || c == REFIL_MASK_GTI; }

// A split GEMM's pending reduction (sum of its partials into C / colsum): what reduce_partials needs and nothing else.
struct ReduceK {
    const float* partial; float* C; float* colsum;
    long sC, sColsum;
    RowMap cmap;
    int ldc, M, N, batch, splits, flags;
};
constexpr int RED_MULTI = 28;        // reductions one reduce_multi launch takes
// defer != nullptr: a split launch writes its partials only and describes the reduction in *defer (splits == 0 there when the
// launch needed none); the caller runs it later with reduce_multi_launch -- many reductions, one launch
// Schedule knobs a caller may set at run time (refil_set_tuning: QLearner's first-call autotuner measures them in situ per
// shape); -1 = the environment switch / built-in rule decides
struct Tuning { long dw4_target = -1, dw4_min_out = -1, dw_target = -1, compose_early = -1, gru_pd = -1, wres_split = -1, dw_split = -1, attn_qkv = -1, dws_target = -1, attn_qkv_wide = -1; };
extern Tuning g_tuning;
int gemm_launch(const refil_gemm_desc& d, hipStream_t st, ReduceK* defer = nullptr);
int reduce_multi_launch(const ReduceK* r, int n, hipStream_t st);
// weight-resident kernel for short-reduction forward projections (gemm_wres.hip); gemm_launch dispatches to it
bool gemm_wres_eligible(const refil_gemm_desc& d);
bool gemm_wres_split_on();      // the projections run the bf16 x 6 form (refil_set_tuning "wres_split" / REFIL_WRES_SPLIT)
// streaming weight-gradient kernel (gemm_dw.hip): writes the split partials, gemm_launch runs the reduction
bool gemm_dw_stream_eligible(const refil_gemm_desc& d);
int gemm_dw_stream_launch(const refil_gemm_desc& d, hipStream_t st);
int gemm_wres_launch(const refil_gemm_desc& d, hipStream_t st);
// second-generation weight-gradient kernel (gemm_dw4.hip): 4 x 4 MFMA tiles per wave, in-workgroup split reduction
bool gemm_dw4_enabled();
bool gemm_dw4_eligible(const refil_gemm_desc& d);
int gemm_dw4_splits(const refil_gemm_desc& d);
int gemm_dw4_launch(const refil_gemm_desc& d, hipStream_t st);
int attn_forward_launch(const refil_attn_desc& d, hipStream_t st);
int attn_backward_launch(const refil_attn_desc& d, hipStream_t st);
int pool_launch(const refil_attn_desc& d, int mode, bool bwd, hipStream_t st);
int gru_forward_launch(const refil_gru_desc& d, hipStream_t st);
int gru_forward_launch2(const refil_gru_desc& d, const refil_gru_desc* second, hipStream_t st);   // two recurrences, one launch
int gru_backward_launch(const refil_gru_desc& d, hipStream_t st);

// entities || one-hot(prev action) -> xe [R*ne, Ep]; contiguous copies of the masks
struct PrepArgs {
    refil_batch b;
    int B, T1, ne, na, ed, A, Ep, last_action, first_step_zero;
    float* xe; uint8_t* emc; uint8_t* amask; uint8_t* em0;
    float* actf;   // [R*na] 1.0 for active agents, 0.0 for inactive ones (row weights of bias terms)
};
int prep_launch(const PrepArgs& a, hipStream_t st, int phases = 3, const uint8_t* skip_a = nullptr, const uint8_t* skip_h = nullptr);

// Row lists of a learner step: which (b, t, entity) rows can influence the loss. Everything is decided on the device
// (no host round trip); consumers read the counts from device memory.
//   t_last[b]   last step of episode b any loss term depends on: 1 + the last t < T with
//               mask[b,t] = filled[b,t] * (1 - terminated[b,t-1]) != 0 (q_learner.py:68-72); steps t > t_last[b] are dead
//               for the live AND the target nets (double-Q reads the live net at t+1, q_learner.py:121-126). -1: none.
//   agent nets  entity (b,t,j) is needed as a key iff some agent row can attend to it under the least restrictive
//               pre-mask (obs_mask[b,t,i,j] == 0 for some i; every imagined variant ORs more onto it,
//               entity_rnn_agent.py:116-117), or it is an active agent itself (its own query row)
//   hypernets   needed iff not (entity_mask[b,t,j] and entity_mask[b,0,j]): the plain mixer masks by the step's entity
//               mask (flex_qmix.py:43-46), the imagined masks by the first step's (entity_rnn_agent.py:99-114)
//   agents      query rows of active agents (entity_mask[b,t,i] == 0); inactive agents' outputs are zeroed by the
//               post-mask (attention.py:66-67) and receive no gradient
// Lists are in (b,t,entity) order (deterministic reductions) and padded to a multiple of 64 entries plus 128 with `trash`
// indices (one row past the logical buffer: readable, overwritable scratch).
struct ListArgs {
    refil_batch b;
    int B, T1, ne, na, learner;            // learner = 0: no filled/terminated (every step is live)
    int use_gt_obs;                        // dims.gt_obs_mask: the agent nets' pre-mask is gt_mask
    uint8_t* emc; uint8_t* em0;            // contiguous entity masks (this step / step 0): WRITTEN here (prep phase 1 is folded in)
    uint8_t* amask; float* actf;           // [R*na] agent mask copy / 1.0 for active agents: written here as well
    int* hint_out;                         // optional: device-visible pinned host memory [8] receiving a copy of `counts`
    int* t_last;                           // [B]
    uint8_t* kdead_a; uint8_t* kdead_h;    // [R*ne] 1 = K/V row of the agent nets / hypernets is not computed
    int* cnt;                              // [3][R] per-row counts (scratch), lists: 0 agent-net entities, 1 hypernet entities, 2 agents
    int* off;                              // [3][R+1] exclusive scans (scratch)
    int* list_ea; int* list_eh; int* list_a;   // [NE+256], [NE+256], [NA+256]
    int* counts;                           // [8]: the three list lengths, live (b,t) rows, then the derived lists' lengths
    // derived agent-row lists for the layers behind the attention cores (rows = variant * NA + agent row):
    //   list_t   agent rows (b,t,i) of live steps whose agent is active at SOME live step of the episode: the recurrent
    //            tail (fc2, GRU input gates, their gradients) -- an agent that is never active never produces a Q-value
    //            that is used (entity_rnn_agent.py:57-60) and its hidden state feeds nothing else
    //   rep[k]   `copies` shifted copies of list_a (src 0) or list_t (src 1): entry = src[j] + c * NA, padded with `trash`
    uint8_t* ever;                         // [B*na] scratch: agent active at some live step
    int* list_t;                           // [NA+256]
    struct Rep { int* list; int src, copies, trash; } rep[4];   // lengths -> counts[4 + k]; list = NULL: unused
    unsigned long long* sync;              // [B][8] scratch of the one-launch version: per-episode list lengths as {tag, value} granules
    unsigned tag;                          // set by lists_launch
    int* err_out; const int* err_host;     // optional: device / host view of a sticky pinned error word (one-launch form: grid exchange timed out)
};
int lists_launch(const ListArgs& a, hipStream_t st);

// hsx slot 0 <- h0 (or zeros)
int set_h0_launch(float* hsx, const float* h0, int GB, int T1, int na, int H, hipStream_t st);
// h_out[gb,i,:] <- hsx slot T1
int get_hT_launch(const float* hsx, float* h_out, int GB, int T1, int na, int H, hipStream_t st);

struct QSelArgs {
    const float* q;        // [G, B*T1*na, A] live agent Q (inactive agents zeroed)
    const float* tq;       // [B*T1*na, A] target agent Q
    const int64_t* actions; long ac_sB, ac_sT;
    const int32_t* avail; long av_sB, av_sT;
    float* chosen;         // [G,B,T,na]
    float* tmax;           // [B,T,na]
    int G, B, T1, na, A, double_q;
};
int qselect_launch(const QSelArgs& a, hipStream_t st);

struct QSelBwdArgs {
    const float* dchosen;  // [G,B,T,na]
    const int64_t* actions; long ac_sB, ac_sT;
    const uint8_t* amask;  // [B*T1*na]
    float* dq;             // [G, B*T1*na, A]
    int G, B, T1, na, A;
    // optional: also d(hidden state) = dq W3 (W3 [A,H] row-major: fc3.weight). dq has one non-zero per row, so the product
    // is a scaled row of W3 -- written here instead of by a [rows x A] x [A x H] GEMM on the critical path
    const float* w3; float* dhs; int H;
    const uint8_t* ever;   // optional [B, na]: rows of agents that are never active are not written (nothing reads them)
};
int qselect_bwd_launch(const QSelBwdArgs& a, hipStream_t st);

// Q head + selection of the recurrent agents in ONE launch at the join of the two chains (entity_rnn_agent.py:57-60 and
// q_learner.py:91,109,115-128): workgroup (b,t) computes q = fc3(h) of the live agents' G copies at step t and of the live
// (copy 0) and target agents at step t+1 straight from the hidden states, zeroes inactive agents, and writes the chosen-action
// Qs and the (double-Q) target values -- instead of two thin GEMMs over all rows + a gather kernel.
struct QHeadArgs {
    const float* hs; const float* ths;         // hidden states [G*B, T1+1, na, H] / [B, T1+1, na, H] (slot t+1 = h_t)
    const float* w3; const float* b3; const float* tw3; const float* tb3;     // fc3 of the live / target agent: [A,H], [A]
    const uint8_t* amask;                      // [B*T1*na] 1 = inactive agent
    const int64_t* actions; long ac_sB, ac_sT;
    const int32_t* avail; long av_sB, av_sT;
    const int* t_last;                         // optional [B]
    float* chosen; float* tmax;                // [G,B,T,na], [B,T,na]
    float* q_out;                              // optional [G, B*T1*na, A]: the live agents' Q values (debug copies)
    int G, B, T1, na, A, H, double_q;
};
bool qhead_eligible(const QHeadArgs& a);
int qhead_launch(const QHeadArgs& a, hipStream_t st);

// what the fused mixing kernel (mix_train_launch) needs to also write the Q head's backward (q_learner.py:91's gather and
// fc3's input gradient): dq [G, B*T1*na, A] = one-hot(action) * d(chosen), dhs [G, B*T1*na, H] = d(chosen) * fc3.weight[action]
struct QHeadBwd {
    float* dq; float* dhs; const float* w3;    // dhs == NULL: off
    const int64_t* actions; long ac_sB, ac_sT;
    const uint8_t* ever;                       // optional [B, na]: rows of never-active agents are not written
    int A, H;
};

// FlexQMixer monotonic mixing on top of the hypernet outputs (flex_qmix.py:96-121)
struct MixArgs {
    const float* x_w1; long s_var;   // [nvar][R*na, M] masked fc2 outputs of hyper_w_1 (variant stride)
    const float* x_wf; const float* x_b1; const float* x_v;   // [R*na, M]
    const float* qs; long s_qs_g;    // [G][B,T,na] chosen agent Qs (copy stride)
    float* q_tot; float* q_tot_im;   // [B,T]
    // backward
    const float* gc_real; const float* gc_im;   // [B,T] dL/dq_tot
    float* dx_w1; float* dx_wf; float* dx_b1; float* dx_v;   // same layouts as x_*
    float* dqs;                       // [G][B,T,na]
    const uint8_t* amask;             // [R*na]
    int B, T1, T, t_off, na, M, imagine, softmax_w, tanh_nl;
    int lin;                          // LinearFlexQMixer (flex_qmix.py:136-172): x_wf/x_b1 unused
    float* ingroup_rows;              // lin + imagine: per-(b,t) in-group weight mass sum_{i<na} w1[i] (or NULL)
    const int* t_last;                // optional [B]: steps t > t_last[b] were skipped upstream (they carry no loss): their
                                      // q_tot is written as 0 and they receive exact-zero gradients
    int presum;                       // x_wf / x_b1 / x_v (and their gradients) are ONE row per (b,t): the sum over the
                                      // active agents of the hypernet output, [R, M] instead of [R*na, M]
};
int mix_forward_launch(const MixArgs& a, hipStream_t st);
int mix_backward_launch(const MixArgs& a, hipStream_t st);

struct TdArgs {
    const float* q_tot; const float* q_tot_im; const float* tq_tot;
    const float* reward; long rw_sB, rw_sT;
    const uint8_t* terminated; long tm_sB, tm_sT;
    const int64_t* filled; long fl_sB, fl_sT;
    float* gc_real; float* gc_im; float* targets; float* stats;
    const int* t_last;                // optional [B]: rows with t > t_last[b] hold stale values (they have mask == 0)
    const float* ingroup_rows;        // optional [B,T] -> stats[REFIL_STAT_INGROUP_SUM]
    int B, T, imagine; float gamma, lmbda;
    int t_limit;                      // refil_batch.t_limit: transitions t >= t_limit - 1 carry no loss (0: none cut)
    int nq;                           // values per (b,t): 1 (mixed q_tot) or n_agents (args.mixer = None: per-agent TD, q_learner.py:131,161)
};
int td_loss_launch(const TdArgs& a, hipStream_t st);
// FlexQMixer learner step in one launch per (b,t) row: live mix, target mix (the step after), TD error and the live mix's
// backward (q_learner.py:134-172 and its gradient). live: forward + backward pointers, t_off 0; targ: forward pointers,
// t_off 1; td: the batch scalars and the outputs q_tot / q_tot_im / tq_tot / targets / gc_real / gc_im (kept for parity
// checks). row_stats [B*T][8]: per-row terms of the stat sums, folded into td.stats by td_stats_launch.
int mix_train_launch(const MixArgs& live, const MixArgs& targ, const TdArgs& td, float* row_stats, hipStream_t st, const QHeadBwd* qb = nullptr);
int td_stats_launch(const float* row_stats, int rows, float* stats, hipStream_t st);

int sum_launch(const float* x, long n, float* out, hipStream_t st);   // out[0] = sum(x)

// out_trans followed by fc2 is ONE linear map per hypernet (no non-linearity in between, flex_qmix.py:47-50):
//   W_c = W_2 W_o [M,h],  b_c = W_2 b_o + b_2.   compose: build (W_c, b_c) of `nets` hypernets;
//   decompose: turn (dL/dW_c, dL/db_c) into dW_2 = G W_o^T + g b_o^T, dW_o = W_2^T G, db_o = W_2^T g, db_2 = g (stored, not added)
struct ComposeArgs {
    const float* W2; long sW2; const float* b2; long sb2; const float* Wo; long sWo; const float* bo; long sbo;   // parameters
    float* Wc; float* bc;                    // [nets][M][h], [nets][M]
    float* bd;                               // optional [nets][M]: W_2 b_o alone (b_c - b_2)
    const float* gc_b2;                      // optional: the vector that becomes db_2 when it differs from gc
    const float* Gc; const float* gc;        // backward inputs, same layouts
    float* dW2; float* db2; float* dWo; float* dbo;   // gradient outputs (parameter strides)
    int nets, M, h;
};
int compose_forward_launch(const ComposeArgs& a, hipStream_t st);
int compose_backward_launch(const ComposeArgs& a, hipStream_t st);
// attention_mfma.hip: matrix-core attention with the agent-sum / broadcast-dO options; -1 = tile shape not instantiated
bool attn_mfma_supported(int ne, int na, int hd);
int attn_mfma_launch_ex(const refil_attn_desc& d, bool bwd, hipStream_t st, int sum_agents, float* nact, int bcast_do, int zero_dead);
int attn_mask_words_launch(const refil_attn_desc& d, unsigned long long* mwords, unsigned long long* rbits, hipStream_t st);
// several attention blocks that share rows and masks (the hypernets of a mixer) in ONE launch
struct AttnNetOpts { int sum_agents, bcast_do; };
int attn_mfma_launch_multi(const refil_attn_desc* descs, const AttnNetOpts* opts, int n, bool bwd, hipStream_t st, float* nact, int zero_dead);

// attention_qkv.hip: in_trans + attention core in one launch (the layer's input and weight instead of Q / K / V); -1 = shape not instantiated
struct AttnQkvSrc { const float* X; const float* W; float* Qo; float* Ko; float* Vo; };
bool attn_qkv_supported(int ne, int na, int heads, int hd);
bool attn_qkv_wide(int ne, int na);
bool attn_qkv_fits(int ne, int na, int heads, int hd, long R, int T1, int nnets);
void attn_qkv_set_lds_budget(long bytes);
int attn_qkv_launch_multi(const refil_attn_desc* descs, const AttnNetOpts* opts, const AttnQkvSrc* src, int n, int ldx, hipStream_t st,
                          float* nact, int zero_dead);

// device view of the calling thread's sticky error word of the one-launch row lists on the current device (learner.hip), or NULL
const int* lists_error_word_dev();
int clip_rmsprop_launch(float* params, const float* grads, float* sq, long n, float lr, float alpha, float eps,
                        float wd, float clip, float* stats, float* scratch, hipStream_t st);

}  // namespace refil
