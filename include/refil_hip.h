/* refil_hip.h -- C ABI of the MI355X-native REFIL learner hot path (librefil_hip.so).
 *
 * The reference (shariqiqbal2810/REFIL) has no FFI: its hot path is the Python call
 *     QLearner.train(batch, t_env, episode_num)            src/learners/q_learner.py:66-201
 * issuing stock PyTorch ops. This library is what a native replacement of that call binds to:
 * plain device pointers + sizes + a HIP stream, no torch types. Every entry point
 *   - borrows (never owns) the caller's device memory,
 *   - enqueues work on `stream` (a hipStream_t passed as void*; NULL = default stream) and returns
 *     without synchronising,
 *   - returns 0 on success, non-zero on error (message via refil_last_error(), thread-local).
 * All floating-point data is fp32 (the reference's dtype); masks are uint8 {0,1} with 1 = masked /
 * inactive; actions int64; avail_actions int32; filled int64 (src/run.py:178-192 scheme).
 *
 * Layout of this header:
 *   1. problem dimensions + flat parameter layout
 *   2. the learner step      (replaces q_learner.py:66-178)
 *   3. MAC / mixer forwards  (replace basic_controller.py:28-67, flex_qmix.py:79-121)
 *   4. building-block operators (exported for tests and for composing new paths)
 */
#ifndef REFIL_HIP_H
#define REFIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* 1. dimensions and parameter layout                                                          */
/* ------------------------------------------------------------------------------------------ */

/* Hyper-parameters the path reads from `args` (src/config/default.yaml:35-58, algs/refil.yaml:3-33;
 * runtime-injected n_agents/n_actions/n_entities/entity_shape: src/run.py:152-176). */
typedef struct refil_dims {
    int32_t B;            /* episodes in the (per-rank) minibatch                                  */
    int32_t T1;           /* stored timesteps = transitions + 1 (batch.max_seq_length)            */
    int32_t ne;           /* n_entities (<= 64)                                                    */
    int32_t na;           /* n_agents = first na entities                                          */
    int32_t ed;           /* entity_shape (raw feature width)                                      */
    int32_t A;            /* n_actions                                                             */
    int32_t d;            /* attn_embed_dim (agent)                                                */
    int32_t heads;        /* attn_n_heads                                                          */
    int32_t H;            /* rnn_hidden_dim (32, 64 or 128)                                        */
    int32_t hyp;          /* hypernet_embed                                                        */
    int32_t M;            /* mixing_embed_dim (<= 64)                                              */
    int32_t entity_last_action;      /* append one-hot previous action to agent entities          */
    int32_t imagine;                 /* 1: REFIL ('imagine' in args.agent), 0: qmix_atten          */
    int32_t softmax_mixing_weights;  /* 1: softmax, 0: abs      (flex_qmix.py:102-113)            */
    int32_t mixer_tanh;              /* 0: elu, 1: tanh         (flex_qmix.py:75-77)              */
    int32_t double_q;                /* q_learner.py:121-128                                       */
    int32_t agent_ff;                /* 1: feed-forward agent fc1->relu(attn)->fc2 (entity_ff_agent.py:29-57),
                                        0: recurrent agent (entity_rnn_agent.py:31-64)                */
    int32_t mixer_lin;               /* 1: LinearFlexQMixer (flex_qmix.py:124-172), 0: FlexQMixer       */
    int32_t mixer_vdn;               /* 1: VDNMixer (modules/mixers/vdn.py:9-10): q_tot = sum_i q_i, no hypernets / parameters */
    int32_t gt_factors;              /* 1: imagine groups = ground-truth factors batch.gt_mask
                                        (entity_ff_agent.py:93-95) instead of the random split;
                                        2: random split OR-ed with them (use_rand_gt_factors, :111-114) */
    int32_t gt_obs_mask;             /* 1: batch.gt_mask replaces obs_mask (entity_ff_agent.py:34-35)   */
    int32_t pooling;                 /* 0: EntityAttentionLayer; 1 / 2: EntityPoolingLayer 'mean' / 'max'
                                        (attention.py:82-132, pooling_type of default.yaml:43). The in_trans slot of
                                        the parameter layout then holds W[w,w] followed by its bias[w].      */
    int32_t mixer_none;              /* 1: args.mixer = None (q_learner.py:19-21,131): no mixing network, the TD loss is taken
                                        per agent on [B,T,n_agents] (mask expanded, :161). Not with imagine (the reference's
                                        shapes do not broadcast there) */
    float gamma;
    float lmbda;
} refil_dims;

/* Offsets (in floats) of every tensor inside the flat parameter buffer the library works on.
 * One buffer holds the agent (EntityAttentionRNNAgent, entity_rnn_agent.py:8-25) followed by the
 * FlexQMixer (flex_qmix.py:61-73). Mixer tensors of the four hypernets are stored net-major per
 * field in the order hyper_w_1, hyper_w_final, hyper_b_1, V, so that field f of net n lives at
 * mix_<f> + n * mix_<f>_stride and the four fc1 weights form one [4*hyp, E] matrix.
 * Weights keep PyTorch's [out, in] row-major layout. */
typedef struct refil_param_layout {
    int64_t total;                    /* floats in the flat buffer (multiple of 4)                 */
    int64_t agent_total;              /* floats of the agent part (mixer part starts here)         */
    int64_t ag_fc1_w, ag_fc1_b, ag_in_w, ag_out_w, ag_out_b, ag_fc2_w, ag_fc2_b;
    int64_t ag_w_ih, ag_w_hh, ag_b_ih, ag_b_hh, ag_fc3_w, ag_fc3_b;
    int64_t mix_fc1_w, mix_fc1_b, mix_in_w, mix_out_w, mix_out_b, mix_fc2_w, mix_fc2_b;
    int64_t mix_fc1_w_stride, mix_fc1_b_stride, mix_in_w_stride, mix_out_w_stride, mix_out_b_stride,
            mix_fc2_w_stride, mix_fc2_b_stride;
} refil_param_layout;

int refil_get_param_layout(const refil_dims* dims, refil_param_layout* out);

/* ------------------------------------------------------------------------------------------ */
/* 2. learner step                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* One sampled replay minibatch = the EpisodeBatch fields QLearner.train reads
 * (q_learner.py:68-73, entity_controller.py:11-30). Leading dims are [B, T1]; sB/sT are the strides
 * (in ELEMENTS of that field) of those two dims, inner dims are contiguous -- so a time-truncated
 * view batch[:, :max_t_filled] (run.py:269-270) can be passed without a copy. */
typedef struct refil_batch {
    const float*   entities;      int64_t ent_sB, ent_sT;     /* [B,T1,ne,ed]   */
    const uint8_t* obs_mask;      int64_t om_sB, om_sT;       /* [B,T1,ne,ne]   */
    const uint8_t* entity_mask;   int64_t em_sB, em_sT;       /* [B,T1,ne]      */
    const int64_t* actions;       int64_t ac_sB, ac_sT;       /* [B,T1,na,1]    */
    const int32_t* avail_actions; int64_t av_sB, av_sT;       /* [B,T1,na,A]    */
    const float*   reward;        int64_t rw_sB, rw_sT;       /* [B,T1,1]       */
    const uint8_t* terminated;    int64_t tm_sB, tm_sT;       /* [B,T1,1]       */
    const int64_t* filled;        int64_t fl_sB, fl_sT;       /* [B,T1,1]       */
    const uint8_t* gt_mask;       int64_t gt_sB, gt_sT;       /* [B,T1,na,ne] ground-truth factor mask (group_matching env,
                                                                 src/run.py:187-188); NULL unless gt_factors / gt_obs_mask */
    const uint8_t* group_bits;    /* [B,ne] the random 2-way entity split (entity_rnn_agent.py:94-96);
                                     drawn by the host so that seeds reproduce the reference's masks.
                                     Ignored when dims.imagine == 0. */
    /* refil_mixer_forward only: the imagined groups as explicit masks -- what FlexQMixer.forward receives as
     * imagine_groups = (Wmask, Imask) (flex_qmix.py:85-94) -- packed per (b,t) row as 64-bit key words in the layout
     * of refil_attn_desc.mask_words with 3 variants: [B*T1][3][16*ceil(na/16)], variant 0 = the entity mask, 1 = Wmask,
     * 2 = Imask rows of the agents (bit j set = key entity j masked; bits >= ne and padded agents all ones), and
     * mask_row_bits [B*T1][3] = (0, 0, word of inactive entities). Replaces group_bits / gt factors when non-NULL.
     * Needs a shape the MFMA attention kernels take and attention (not pooling) hypernets. */
    const uint64_t* mask_words;
    const uint64_t* mask_row_bits;
    /* refil_learner_forward_backward only, optional: a hipEvent_t recorded behind the last write of the per-step fields above
     * (entities ... gt_mask; group_bits / mask_words are NOT covered: they follow `stream`). The step's input assembly and row
     * lists depend on those fields alone; with the event they are enqueued on one of the library's side streams into one of two
     * alternating workspace slots, so that they run beside the END of the previous step on `stream` instead of in front of this
     * one. The caller must not rewrite the fields before the work enqueued on `stream` by this call has completed.
     * NULL: everything is ordered behind `stream` (what the reference's train() does implicitly). Results are identical.
     * Leave it NULL on the first call after the workspace was (re)allocated or zeroed on `stream`: the early work is not
     * ordered behind that zero fill. */
    void* ready_event;
    /* refil_learner_forward_backward only, optional: a counter the caller changes whenever params_target has been (or is being,
     * on `stream`) rewritten since its previous call with this workspace (q_learner.py:203-207 `_update_targets`, checkpoint
     * loads); 0 = unknown. The target networks' forward depends on the batch fields and on params_target alone: on a call whose
     * ready_event is set AND whose target_version (non-zero) and params_target pointer equal the previous call's, it is
     * enqueued right behind the early prologue, on the library's hypernet-chain stream -- beside the END of the previous step --
     * instead of inside this step's forward, and the composed out_trans o fc2 maps of the target nets
     * (functions of params_target alone) are kept from the previous call instead of being rebuilt. Same kernels, same data:
     * results are bit-identical. Pass 0 on the first call after the workspace was (re)allocated or zeroed. The
     * caller must keep params_target unchanged until the work enqueued on `stream` by this call has completed, or change
     * target_version at the next call. */
    uint64_t target_version;
    /* refil_learner_forward_backward only, optional: the caller's time trim. The reference's run loop trains on
     * batch[:, :max_t_filled()] (run.py:269-270); with t_limit = that length (2 <= t_limit <= T1) the FULL-length batch gives the
     * same step -- transitions t >= t_limit - 1 carry no loss weight, their steps are never computed -- while dims.T1, and with it
     * the workspace layout and the early paths above, stay the same from step to step. 0: every transition of the batch counts. */
    int32_t t_limit;
} refil_batch;

/* Scalars produced by a step, as a device array of REFIL_NSTAT floats (sums over this rank's shard,
 * un-normalised, so that they can be all-reduced together with the gradients). */
enum {
    REFIL_STAT_MASK_SUM = 0,   /* sum(mask)                          q_learner.py:165 normaliser    */
    REFIL_STAT_TD_SQ,          /* sum((mask*td)^2)                   -> q loss numerator            */
    REFIL_STAT_IM_TD_SQ,       /* sum((mask*td_imagine)^2)           -> im_loss numerator           */
    REFIL_STAT_TD_ABS,         /* sum(|mask*td|)                     q_learner.py:193               */
    REFIL_STAT_QTOT_SUM,       /* sum(q_tot*mask)                    q_learner.py:194               */
    REFIL_STAT_TARGET_SUM,     /* sum(targets*mask)                  q_learner.py:195               */
    REFIL_STAT_GRAD_NORM,      /* written by refil_clip_rmsprop_step q_learner.py:177               */
    REFIL_STAT_INGROUP_SUM,    /* LinearFlexQMixer only: sum over (b,t) of the in-group mixing weight mass
                                  sum_{i<na} w1[i] of the imagined mix (flex_qmix.py:166-170, unmasked rows)  */
    REFIL_NSTAT
};

/* Optional copies of intermediates for parity tests (any pointer may be NULL). */
typedef struct refil_debug_out {
    float* q;              /* [G,B,T1,na,A] live per-agent Q (G = 3 if imagine else 1)             */
    float* chosen_q;       /* [G,B,T,na]    Q of the taken actions (q_learner.py:91,109)           */
    float* target_max_q;   /* [B,T,na]      (q_learner.py:121-128)                                 */
    float* q_tot;          /* [B,T]                                                                */
    float* q_tot_imagine;  /* [B,T]                                                                */
    float* target_q_tot;   /* [B,T]                                                                */
    float* targets;        /* [B,T]         (q_learner.py:157)                                     */
} refil_debug_out;

/* Bytes of scratch HBM refil_learner_forward_backward needs for `dims` (0 on invalid dims). */
size_t refil_learner_workspace_bytes(const refil_dims* dims);

/* Forward (live + target nets, mixers, TD loss) and the full hand-written backward.
 * Replaces q_learner.py:66-176 (everything up to and including loss.backward()).
 *   params_live / params_target : flat buffers laid out by refil_get_param_layout
 *   grads  : [layout.total + REFIL_NSTAT] floats. grads[0:total] receives d(SUM-loss)/d(param),
 *            i.e. the gradient of (1-lmbda)*sum(td^2) + lmbda*sum(td_im^2) WITHOUT the 1/sum(mask)
 *            normaliser; grads[total:] receives the REFIL_STAT_* sums. Under data parallelism the
 *            caller all-reduces (SUM) this one buffer; refil_clip_rmsprop_step then applies
 *            1/sum(mask) -- exactly the reference's global-mean loss (q_learner.py:165,171). */
int refil_learner_forward_backward(const refil_dims* dims, const refil_batch* batch,
                                   const float* params_live, const float* params_target,
                                   float* grads, void* workspace, size_t workspace_bytes,
                                   const refil_debug_out* debug, void* stream);

/* Data-parallel overlap (optional). The flat gradient buffer is [agent | mixer | REFIL_NSTAT stat sums]; the mixer part
 * and the stat sums are final as soon as the hypernets' backward is, long before the agent's BPTT finishes. A hook set
 * here is called by refil_learner_forward_backward (on the calling thread, during enqueue) once every kernel writing
 * grads[agent_total : total + REFIL_NSTAT] has been enqueued; `stream` is the HIP stream on which they complete, so a
 * collective enqueued behind that stream reduces the mixer bucket while the agent chain is still running. NULL: off. */
typedef void (*refil_grads_hook)(void* user, void* stream);
int refil_set_mixer_grads_hook(refil_grads_hook hook, void* user);

/* One-shot all-reduce(SUM) over peer memory for the step's small [gradients | stats] message (replaces the ring
 * all-reduce torch.distributed issues for q_learner.py:176's gradients under data parallelism):
 * every rank stages its buffer in IPC-exported device memory, signals a flag, and sums all ranks' staged buffers in rank
 * order (identical results on every rank) -- one hop over the point-to-point xGMI links instead of 2 (N-1).
 *   create   allocates two staging buffers + a flag word and writes their three IPC handles to handles_out
 *            [3][REFIL_IPC_HANDLE_BYTES]; the caller exchanges them between the ranks (any side channel);
 *   connect  takes every rank's handles [world][3][REFIL_IPC_HANDLE_BYTES] (own entry ignored) and maps the peers;
 *   allreduce  in place on `inout` (n_floats of create), stream-ordered, no host synchronisation; every rank must call it
 *            the same number of times. A peer that does not arrive within REFIL_ONESHOT_TIMEOUT_S (default 120 s) makes
 *            the reduction write NaN over `inout` (nothing behind it can use rank-local gradients unnoticed) and record
 *            the call number in a host-mapped status word instead of hanging the GPU: refil_oneshot_status reads it
 *            without synchronising, and every later refil_oneshot_allreduce fails with that message.
 * world <= 16. Needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver only supports dmabuf IPC. */
#define REFIL_IPC_HANDLE_BYTES 64
int refil_oneshot_create(int32_t world, int32_t rank, int64_t n_floats, uint8_t* handles_out, void** ctx_out);
int refil_oneshot_connect(void* ctx, const uint8_t* all_handles);
int refil_oneshot_allreduce(void* ctx, float* inout, void* stream);
int refil_oneshot_status(void* ctx, int32_t* timed_out);
int refil_oneshot_destroy(void* ctx);

/* The step's ONE collective for a non-Python host (the reference has none: src/ holds no torch.distributed call; SURVEY.md
 * section 8e is the specification): in-place all-reduce(SUM) of the flat fp32 buffer [gradients | stat sums]
 * (refil_learner_forward_backward's `grads`, n_floats = total + REFIL_NSTAT) on the caller's RCCL communicator
 * (ncclComm_t), enqueued on `stream`; the global sum(mask) normaliser arrives in the same message and is applied by
 * refil_clip_rmsprop_step. librccl.so is resolved at the first call (REFIL_RCCL_LIB overrides the name). */
int refil_allreduce_flat(float* buf, int64_t n_floats, void* comm, void* stream);

/* One whole training step in ONE call (QLearner.train, q_learner.py:66-178: forward, loss, backward, the data-parallel collective
 * when `comm` is given, clip_grad_norm_ + RMSprop): refil_learner_forward_backward, refil_allreduce_flat (comm != NULL: the caller's
 * ncclComm_t; NULL: single process) and refil_clip_rmsprop_step enqueued back to back on `stream` -- same kernels, same order, same
 * results bit for bit as the three calls; one FFI crossing per step for a host whose per-call overhead matters (small shards).
 * `hyper` = {lr, alpha, eps, weight_decay, grad_norm_clip} (RMSprop as the reference configures it, q_learner.py:37-38,177).
 * `scratch`: >= 4096 bytes of device memory for the gradient-norm reduction. */
typedef struct refil_opt_hyper { float lr, alpha, eps, weight_decay, grad_norm_clip; } refil_opt_hyper;
int refil_learner_step(const refil_dims* dims, const refil_batch* batch, float* params_live, const float* params_target,
                       float* grads, float* square_avg, const refil_opt_hyper* hyper, void* comm,
                       void* workspace, size_t workspace_bytes, void* scratch, void* stream);

/* Diagnostics (synchronises the stream): which rows the LAST refil_learner_forward_backward on this workspace / dims
 * actually processed. The step skips rows that cannot influence the loss -- entity rows no query can attend to, query rows
 * of inactive agents, steps after an episode's last loss-carrying step -- through device-side row lists (no host round
 * trip; results identical to the dense schedule up to the summation order of the weight gradients; REFIL_DENSE=1 turns
 * it off). out[0..5] = {lists active, entity rows (agent nets), entity rows (hypernets), agent rows, live (b,t) rows,
 * B*T1}; with lists inactive the dense counts are returned. */
int refil_learner_row_counts(const refil_dims* dims, void* workspace, size_t workspace_bytes, int32_t* out, void* stream);

/* clip_grad_norm_ + RMSprop.step on the flat buffers (q_learner.py:37-38,177-178):
 *   g = grads[0:n] / grads[n + REFIL_STAT_MASK_SUM];  norm = ||g||_2  (stored to grads[n+GRAD_NORM]);
 *   g *= min(1, clip / (norm + 1e-6));  g += weight_decay * p;
 *   sq = alpha*sq + (1-alpha)*g^2;  p -= lr * g / (sqrt(sq) + eps).
 * scratch: >= 4096 bytes of device memory. */
int refil_clip_rmsprop_step(float* params, const float* grads, float* square_avg, int64_t n,
                            float lr, float alpha, float eps, float weight_decay, float grad_norm_clip,
                            float* grads_stats, void* scratch, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* 3. forwards used outside the learner                                                        */
/* ------------------------------------------------------------------------------------------ */

/* BasicMAC.forward / EntityMAC._build_inputs / (Imagine)EntityAttentionRNNAgent.forward
 * (basic_controller.py:28-67, entity_controller.py:11-30, entity_rnn_agent.py:31-64,87-126).
 * Runs T1 = dims.T1 steps starting from hidden state h0.
 *   prev_actions : actions of the step BEFORE each stored step: prev_actions[b,t] is the action whose
 *                  one-hot is appended at step t; pass `actions` shifted by the caller, or set
 *                  first_step_zero=1 to use actions[b,t-1] with zeros at t=0 (the t=None case).
 *   h0    : [G,B,na,H] or NULL for zeros;   h_out : [G,B,na,H] final hidden state (may be NULL)
 *   q_out : [G,B,T1,na,A] */
size_t refil_agent_workspace_bytes(const refil_dims* dims);
int refil_agent_forward(const refil_dims* dims, const refil_batch* batch, int32_t first_step_zero,
                        const float* params, const float* h0, float* h_out, float* q_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* FlexQMixer.forward (flex_qmix.py:79-121) / LinearFlexQMixer.forward (:136-172, dims.mixer_lin) over steps
 * [t0, t0+T) of the batch. ingroup_sum (device float, may be NULL): LinearFlexQMixer's ret_ingroup_prop numerator.
 *   agent_qs [B,T,na]; agent_qs_imagine [B,T,2*na] or NULL; q_tot / q_tot_imagine [B,T]. */
size_t refil_mixer_workspace_bytes(const refil_dims* dims);
int refil_mixer_forward(const refil_dims* dims, const refil_batch* batch, int32_t t0, int32_t T,
                        const float* params, const float* agent_qs, const float* agent_qs_imagine,
                        float* q_tot, float* q_tot_imagine, float* ingroup_sum,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Device-side replay sampling: for every field copy the sampled episodes from the ring storage into a staging
 * minibatch, dst[b] <- src[episode_ids[b]] (reference: ReplayBuffer.sample -> EpisodeBatch.__getitem__,
 * src/components/episode_buffer.py:123-159,233-240; one index_select per field there). Byte-level: a field is
 * [capacity, src_episode_bytes] in the buffer and [B, dst_episode_bytes] in the staging batch; copy_bytes (<= both)
 * allows copying only the first max_t steps. episode_ids: int64 on the device, values in [0, capacity). */
typedef struct refil_gather_field {
    const void* src; void* dst;
    int64_t src_episode_bytes, dst_episode_bytes, copy_bytes;
    /* unpack_width > 0: the field is stored BIT-PACKED in the buffer (refil_pack_mask_bits: one uint64 per row of
     * unpack_width <= 64 mask bytes, e.g. obs_mask[b,t,i,:] -- episode_buffer.py keeps the bytes) and the gather expands it:
     * dst byte (row, j) = bit j of src word `row`. src_episode_bytes = 8 * rows, copy_bytes counts DESTINATION bytes. */
    int32_t unpack_width, reserved;
} refil_gather_field;
/* dst[r] = sum_j (src[r * width + j] != 0) << j for `rows` rows of `width` <= 64 mask bytes (the storage format of byte masks
 * in the device replay buffer: 8 bytes per row instead of `width`). */
int refil_pack_mask_bits(const uint8_t* src, uint64_t* dst, int64_t rows, int32_t width, void* stream);
int refil_replay_gather(const refil_gather_field* fields, int32_t n_fields, const int64_t* episode_ids,
                        int32_t B, int64_t capacity, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* 4. building-block operators                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* physical row of logical row r = (r / grp) * gstride + (r % grp) + off   (grp == 0: identity) */
typedef struct refil_rowmap { int32_t grp, gstride, off; } refil_rowmap;

enum {
    REFIL_GEMM_RELU       = 1,   /* C = max(C, 0)                                                   */
    REFIL_GEMM_RELU_BWD   = 2,   /* C *= (aux > 0), aux indexed like C                              */
    REFIL_GEMM_ACCUM      = 4,   /* C += result                                                     */
    REFIL_GEMM_A_OUTC     = 8,   /* A element (m,k) at A[k*lda + m] instead of A[m*lda + k]         */
    REFIL_GEMM_B_OUTC     = 16,  /* B element (n,k) at B[k*ldb + n] instead of B[n*ldb + k]         */
    REFIL_GEMM_COLSUM_A   = 32   /* also emit sum_k A(m,k) -> colsum[m] (bias gradients)            */
};

/* C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 * nn.Linear forward is A=x [rows,in], B=W [out,in]; dX = dY W uses B_OUTC; dW = dY^T X uses
 * A_OUTC|B_OUTC with the row dimension as the reduction. `batch` independent problems are addressed
 * with the s* strides. splits > 1 cuts the reduction into `splits` deterministic partial sums which a
 * second kernel adds up (partials live in `partial`, >= batch*splits*M*N floats (+ batch*splits*M for colsum)). */
typedef struct refil_gemm_desc {
    const float* A; const float* B; float* C;
    const float* bias;            /* [N] added to every row, or NULL                               */
    const float* aux;             /* RELU_BWD source (same ld / rowmap as C), or NULL              */
    const uint8_t* rowmask;       /* rowmask[r % rowmask_mod] != 0 -> row r of C is zeroed; or NULL */
    float* colsum;                /* COLSUM_A output [M]                                           */
    float* partial;               /* split-K scratch                                               */
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int64_t sA, sB, sC, sBias, sColsum;   /* per-batch strides in floats                           */
    refil_rowmap a_map, b_map, c_map;     /* applied to the MEMORY row index of A / B / C          */
    int32_t rowmask_mod;
    int32_t batch, splits, flags;
    /* optional second bias scaled per row (plain x W^T products only): C[r][:] += rowscale[r % rowscale_mod] * bias2[:]
     * (per-batch stride of bias2 = sBias). Used for bias terms that apply to a subset / a multiplicity of the rows. */
    const float* bias2; const float* rowscale; int32_t rowscale_mod;
    /* optional ROW LIST (device memory): the product runs over the logical rows r in [0, *row_count) and row r lives at
     * memory row row_index[r] (a_map / b_map / c_map are applied on top). For x W^T and dY W products the list addresses
     * the rows of A and C (and aux) and M is only the upper bound of *row_count; for dY^T X (A_OUTC|B_OUTC) it addresses
     * the reduction rows of A and B and K is the upper bound. The count never visits the host, so rows that cannot
     * influence a training step (padded entities, steps after an episode's end) are skipped without a synchronisation.
     * The list must be padded up to a multiple of 64 entries plus 128 with the index of a scratch row that may be read (finite
     * values) and overwritten. Supported by the weight-resident (bound M >= 256, N % 32 == 0, K <= 128, or <= 256 with
     * RELU_BWD; no rowmask / bias2) and the streaming-dW kernels (splits >= 2); other shapes return an error. */
    const int32_t* row_index; const int32_t* row_count;
    int32_t row_count_hint;   /* host-side guess of *row_count (e.g. the previous step's value), 0 = unknown: only shapes
                                 the launch grid / tile width, never the result */
} refil_gemm_desc;

int refil_gemm(const refil_gemm_desc* desc, void* stream);

/* pre-softmax mask variants of the entity attention (closed forms: SURVEY.md section 8a-5)           */
enum {
    REFIL_MASK_OBS = 0,        /* obs_mask[b,t,i,j]                           entity_rnn_agent.py:39-45 */
    REFIL_MASK_OBS_WITHIN,     /* !same(i,j) | obs_mask                       entity_rnn_agent.py:116   */
    REFIL_MASK_OBS_INTERACT,   /*  same(i,j) | obs_mask                       entity_rnn_agent.py:117   */
    REFIL_MASK_ENTITY,         /* inactive_i(t) | inactive_j(t)               flex_qmix.py:43-46        */
    REFIL_MASK_WITHIN,         /* !same(i,j)                                  entity_rnn_agent.py:111   */
    REFIL_MASK_INTERACT,       /*  same(i,j) | inactive0_i | inactive0_j      entity_rnn_agent.py:112   */
    /* ground-truth factor variants (entity_ff_agent.py:93-95,117-121): gt = gt_mask[b,t,i,j]            */
    REFIL_MASK_OBS_GTW,        /*  gt | obs_mask                                                        */
    REFIL_MASK_OBS_GTI,        /* !gt | obs_mask                                                        */
    REFIL_MASK_GTW,            /*  gt | inactive0_i | inactive0_j                                       */
    REFIL_MASK_GTI,            /* !gt | inactive0_i | inactive0_j                                       */
    /* randomised ground-truth factors (entity_ff_agent.py:111-121): within = !same | gt, interact = !within */
    REFIL_MASK_OBS_RGTW,       /*  !same | gt | obs_mask                                                */
    REFIL_MASK_OBS_RGTI,       /* (same & !gt) | obs_mask                                               */
    REFIL_MASK_RGTW,           /*  !same | gt                                                           */
    REFIL_MASK_RGTI,           /* (same & !gt) | inactive0_i | inactive0_j                              */
    REFIL_MASK_COUNT
};

/* Multi-head masked attention core of EntityAttentionLayer (attention.py:48-64) for `nvar` mask
 * variants sharing Q/K/V: logits = QK^T / sqrt(hd), mask -> -inf, softmax over entities, fully
 * masked row -> 0, times V, heads merged. One (row, head) per workgroup. Backward recomputes the
 * softmax and accumulates dQ/dK/dV over the variants. */
typedef struct refil_attn_desc {
    const float* Q;  int32_t ldq;          /* [R*na, >=w]   row r*na+i                             */
    const float* K;  const float* V; int32_t ldkv;   /* [R*ne, ...] row r*ne+j                     */
    float* O;        int64_t sO;  int32_t ldo;       /* variant v at O + v*sO, [R*na, w]           */
    const float* dO;                                  /* backward: same layout as O                */
    float* dQ; float* dK; float* dV;                  /* backward outputs, same layouts as Q/K/V   */
    int32_t R, T1, ne, na, heads, hd;                 /* row r -> episode b = r / T1               */
    int32_t nvar; int32_t var[3];
    const uint8_t* obs_mask; int64_t om_sB, om_sT;    /* [B,T1,ne,ne]                              */
    const uint8_t* ent_mask;                          /* contiguous [R,ne]                         */
    const uint8_t* ent_mask0;                         /* [B,ne] entity_mask at t=0                 */
    const uint8_t* group_bits;                        /* [B,ne]                                    */
    const uint8_t* gt_mask;  int64_t gt_sB, gt_sT;    /* [B,T1,na,ne], for the *_GT* variants      */
    /* optional row skipping (all NULL: every row is processed). t_last[b]: rows of episode b with t > t_last[b] are
     * left untouched (forward and backward). kv_dead [R*ne] / q_dead [R*na]: rows of K,V / Q (and dO) whose producer
     * skipped them; they enter as zeros whatever the buffers hold (with precomputed mask words they are not even fetched)
     * and the backward leaves their dK / dV / dQ rows unwritten (a dead key must be masked in every variant, a dead
     * query's output must be discarded by the caller: those gradient rows are exact zeros nobody may read). */
    const int32_t* t_last; const uint8_t* kv_dead; const uint8_t* q_dead;
    /* optional: the mask words of every row built ahead of the launch by refil_attn_mask_words (one 64-bit word per
     * (row, variant, agent), bit j = key j masked; agents padded to a multiple of 16) and three words per row (dead K/V
     * rows, dead Q rows, inactive entities). Several launches that share the masks (forward / backward, live / target nets)
     * then skip their mask phase. mask_words_nvar = variants stored per row (>= nvar; 0: nvar). */
    const void* mask_words; const void* row_bits; int32_t mask_words_nvar;
} refil_attn_desc;

int refil_attn_forward(const refil_attn_desc* desc, void* stream);
/* mask_words: [R][nvar][16*ceil(na/16)] uint64, row_bits: [R][3] uint64 (see refil_attn_desc) for the variants of desc */
int refil_attn_mask_words(const refil_attn_desc* desc, void* mask_words, void* row_bits, void* stream);
int refil_attn_backward(const refil_attn_desc* desc, void* stream);

/* EntityAttentionLayer forward from the layer's INPUT (attention.py:46-64): in_trans (`query, key, value = in_trans(x).chunk(3)`,
 * no bias) and the attention core of refil_attn_forward in ONE launch -- Q / K / V never visit HBM unless asked for. The fp32
 * products of in_trans run as six bf16 matrix-pipe products of an exact 3-way operand split with fp32 accumulate (the "wres_split"
 * = 6 arithmetic of refil_set_tuning). What the learner step uses for its attention blocks (the target networks of
 * q_learner.py:111-113,154 store nothing; the live ones keep q_out / k_out / v_out for refil_attn_backward).
 *   attn   rows, mask variants, O / sO / ldo, t_last, and REQUIRED precomputed mask_words / row_bits (refil_attn_mask_words: the
 *          dead K/V / Q rows are taken from row_bits); attn.Q / K / V / kv_dead / q_dead are ignored, ldq / ldkv describe the stores
 *   X      layer input [R*ne, >= w] (leading dimension ldx), w = heads * hd; rows of dead entities are not read
 *   W_in   in_trans.weight [3w, w] row-major: rows [0,w) -> query, [w,2w) -> key, [2w,3w) -> value
 *   q_out / k_out / v_out   optional (NULL: not stored): the projections in the layouts of attn.Q / K / V; rows of dead queries /
 *          keys are not written
 * Dead keys keep refil_attn_desc's meaning: a K / V row that row_bits marks dead is a ZERO row -- where a mask variant leaves it visible it
 * takes part in the softmax with logit 0 and value 0, whether or not the key compaction computes its tile.
 * Row bits: an agent that row_bits marks alive as a QUERY is treated as alive as a key as well (its x row is read -- the query is
 * projected from the registers the key and the value come from --, its K / V rows are written when stores are asked for, and it is
 * attended to unless the mask words exclude it). The learner's row lists only produce rows where query-alive implies key-alive.
 * Shapes: n_entities <= 48, n_agents <= 32, head dim 16 or 32, w = 64 or 128; with more than 32 entities or more than 16 agents:
 * head dim 32 at w = 128 or head dim 16 at w = 64. Others, and launches whose row table does not fit the LDS
 * (about B*T1 > 350 k rows per 16 (net, head) slices on 256 CUs), return an error. */
typedef struct refil_attn_qkv_desc {
    refil_attn_desc attn;
    const float* X; int32_t ldx;
    const float* W_in;
    float* q_out; float* k_out; float* v_out;
} refil_attn_qkv_desc;
int refil_attn_qkv_forward(const refil_attn_qkv_desc* desc, void* stream);

/* Masked entity pooling core of EntityPoolingLayer (attention.py:114-123) under the same mask variants:
 *   O[v][r*na+i][c] = pool_j ( masked_v(i,j) ? 0 : K[r*ne+j][c] ),  pool = mean over ALL ne entities (mode 1) or
 *   max (mode 2; masked entities enter as zeros). K = the in_trans output [R*ne, w] (ldkv), w = heads*hd.
 * Backward: dK[r*ne+j][c] = sum over variants and agents of the routed dO (mean: dO/ne to every unmasked j;
 * max: to the FIRST j attaining the maximum, nothing if that j is a masked zero). Q/V/dQ/dV are ignored. */
int refil_pool_forward(const refil_attn_desc* desc, int32_t mode, void* stream);
int refil_pool_backward(const refil_attn_desc* desc, int32_t mode, void* stream);

/* nn.GRUCell unrolled over T1 steps (entity_rnn_agent.py:49-55) as ONE persistent kernel per
 * 16-row tile, W_hh fragments resident in registers. Logical row r in [0,NR) = (gb, i) with
 * gb = r / na; time-major storage inside each gb:
 *   gi   [(gb*T1 + t)*na + i, 3H]   = x_t W_ih^T + b_ih (precomputed by refil_gemm)
 *   hsx  [(gb*(T1+1) + t)*na + i, H]  slot 0 = h0 (read), slot t+1 = h_t (written)
 *   save_r/z/n/ghn [(gb*T1+t)*na+i, H] gates kept for the backward (NULL: inference)            */
typedef struct refil_gru_desc {
    const float* gi; float* hsx;
    const float* w_hh; const float* b_hh;
    float* save_r; float* save_z; float* save_n; float* save_ghn;
    /* backward */
    const float* dhs;        /* [(gb*T1+t)*na+i, H] external gradient on h_t (from fc3)           */
    float* dgi;              /* [(gb*T1+t)*na+i, 3H] output: d(gi) = (dr, dz, dn) pre-activation gradients          */
    float* dgh;              /* [(gb*T1+t)*na+i, H]  output: the n block of d(gh) ONLY -- d(gh) = (dgi_r, dgi_z, dgh): the
                                r / z blocks equal dgi's (GRUCell adds gi and gh there) and are not stored twice          */
    int32_t NR, T1, na, H;
    /* optional: t_last[b] for b = gb % B -- the recurrence of episode b stops after step t_last[b] (forward: later
     * hsx / save slots are left untouched; backward: starts there and writes zeros to dgi / dgh of the later steps). */
    const int32_t* t_last; int32_t B;
    int32_t zero_h0;         /* forward: 1 = the initial hidden state is zero (BasicMAC.init_hidden, basic_controller.py:41-42):
                              * the kernel writes slot 0 of hsx itself instead of reading it                           */
    const uint8_t* ever;     /* optional [B, na] (needs B): 0 = agent i of episode gb % B is never active, nothing downstream
                              * reads its rows: neither fetched nor written (their hsx / gate / gradient rows stay untouched) */
} refil_gru_desc;

int refil_gru_forward(const refil_gru_desc* desc, void* stream);
int refil_gru_backward(const refil_gru_desc* desc, void* stream);

/* Optional per-kernel timing (HIP events recorded on the launch stream around every kernel the
 * library launches, aggregated per kernel symbol). bench.py uses it for the roofline fraction.
 * refil_profile_collect synchronises the device and returns MINUS the number of entries written
 * (so 0 / positive values keep meaning "error code"). flops/bytes are ALGORITHMIC totals. */
typedef struct refil_profile_entry {
    char name[96];          /* kernel symbol, e.g. "gemm_kernel<2,2,2,2,false,false>"             */
    int64_t launches;
    double total_ms;
    double flops;           /* algorithmic fp32 FLOPs (2 m n k per product)                                 */
    double bytes;
    double flops_bf16x6;    /* the part of `flops` the kernel computes as six bf16 matrix-pipe products of an exact 3-way
                               operand split (roof: the dense bf16 peak / 6); the rest runs the fp32 matrix instructions */
} refil_profile_entry;
int refil_profile_enable(int on);
int refil_profile_collect(refil_profile_entry* out, int max_entries);

/* Two-stream overlap of the agent chain with the hypernet chain inside refil_learner_forward_backward:
 * 1 = on (default), 0 = serialise everything on the caller's stream (profiling one kernel at a time),
 * -1 = follow the environment variable REFIL_NO_OVERLAP. */
int refil_set_overlap(int on);

/* Schedule knobs that do not change the arithmetic of a step but its launch sizes / launch order (the summation order of the
 * split weight-gradient reductions follows the launch size: results agree to rounding): "dw4_target" / "dw_target" / "dws_target" workgroups
 * per 4x4-tile / streamed / bf16 x 6 weight-gradient launch, "dw4_min_out" smallest output taken by the 4x4-tile kernel, "compose_early"
 * 0 / 1, "gru_pd" 2 / 4 steps of prefetch in the 4-row recurrences. Two knobs choose the matrix instruction of the fp32 products
 * (same accuracy, different rounding): "wres_split" (projections with a reduction <= 256) and "dw_split" (weight gradients with
 * 65 .. 128-column outputs): 6 (default) = six bf16 matrix-pipe products of a 3-way operand split with fp32 accumulate,
 * 0 = v_mfma_f32_32x32x2_f32 (environment: REFIL_WRES_SPLIT / REFIL_DW_SPLIT). Operand range of the split form: an Inf operand gives NaN
 * (x - bf16(x) = Inf - Inf) where the fp32 instruction gives Inf, |x| >= 3.3962e38 rounds its leading piece to Inf; everything else,
 * subnormals included, behaves as the fp32 instruction does (tests/test_gpu_ops.py::test_wres_split_edge_operands).
 * "attn_qkv": which attention blocks of the learner step run in_trans + attention core as ONE launch (refil_attn_qkv_forward's kernel):
 * bit 0 target hypernets, 1 target agent, 2 live hypernets, 3 live agent; default 15 where the shape is instantiated and "wres_split"
 * is 6; 0 = the projection launches + attention-core launch (environment: REFIL_ATTN_QKV). "attn_qkv_wide": 1 = the same for more than
 * 32 entities / 16 agents (three key tiles / two agent tiles, one wave per SIMD; parity-tested, not yet timed: default 0; environment:
 * REFIL_ATTN_QKV_WIDE). "qkv_lds_budget": bytes of LDS a fused launch may use (default / <= 0: 160 KB) -- lowers the point at which the
 * learner falls back to the separate launches (tests).
 * value -1 restores the built-in rule (or its environment switch). Process-wide. The best setting depends on the shape
 * AND on what shares the GPU, so QLearner.train measures the candidates in situ on its first call per shape
 * (refil_amd/learners/q_learner.py: _autotune). No counterpart in the reference. */
int refil_set_tuning(const char* name, int64_t value);

/* Schedule counters of the calling thread since the library was loaded (diagnostics: which learner steps took the early paths of
 * refil_batch.ready_event / target_version). name: "learner_steps", "early_prologue_steps", "early_target_hypernet_steps",
 * "early_target_agent_steps". Returns -1 for an unknown name. No reference counterpart (q_learner.py runs everything in order). */
int64_t refil_get_stat(const char* name);

/* The calling thread's internal hypernet-chain stream on the current device (hipStream_t, created lazily; valid until
 * refil_release_streams). For producers of learner batches: work enqueued there runs behind the previous step's hypernet
 * backward and in front of the next step's early prologue (refil_batch.ready_event) -- ReplayBuffer.sample() puts its
 * gather launch there (episode_buffer.py:233-240 has no counterpart: the reference samples synchronously). */
int refil_side_stream(void** stream_out);

/* Destroys the calling thread's internal side streams / events (all devices). They are re-created lazily by
 * the next refil_learner_forward_backward; call before tearing the HIP context down. */
int refil_release_streams(void);

const char* refil_last_error(void);
int refil_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REFIL_HIP_H */
