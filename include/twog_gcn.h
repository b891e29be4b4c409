/*
 * twog_gcn.h -- C ABI of lib2ggcn_hip.so: the MI355X (gfx950) kernels of the 2G-GCN hot path.
 *
 * The reference (tanqiu98/2G-GCN) is pure Python/PyTorch and has no FFI of its own; its hot path is TGGCN.forward
 * (vhoi/models.py:584-933) and the ATen ops that forward dispatches. Each entry point below replaces one group of
 * those ops (cited per function; all citations are relative to the reference tree). Conventions:
 *   - every pointer is a DEVICE pointer to fp32 unless stated, owned by the caller, and must stay alive until the work
 *     enqueued on `stream` has completed (torch autograd's saved tensors guarantee that on the Python side);
 *   - no entry point allocates, synchronises the device, or keeps global mutable state (re-entrant per stream);
 *   - return value: 0 on success, a negative hipError_t / negative argument-error code otherwise; nothing throws;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *
 * Strided rows: most kernels address matrix rows through `twog_rows_t`, so a (clip, entity) row set living inside a
 * larger (clip, time, entity, feature) tensor is used in place, with no gather/cat copy:
 *      address(row r, col c) = ptr + (r / inner) * ld_outer + (r % inner) * ld_inner + c        (c contiguous)
 */
#ifndef TWOG_GCN_H
#define TWOG_GCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float* ptr;
    int64_t ld_outer; /* elements between consecutive outer groups    */
    int64_t ld_inner; /* elements between consecutive rows in a group */
    int32_t inner;    /* rows per outer group (<= 1: plain row stride = ld_outer) */
    int32_t pad_;
} twog_rows_t;

/* Human-readable build/info string (arch, tile sizes). */
const char* twog_version(void);

/* ===============================================================================================================
 * Dense projections on fp32 MFMA (v_mfma_f32_32x32x2_f32). Replaces every nn.Linear of build_mlp
 * (pyrutils/torch/models.py:31-36), the GRU/GRUCell input and hidden projections (vhoi/models.py:267-320), the
 * message MLPs (:323-520), the label heads (:552-580), the 1x1 convolutions of Geo_gcn (pyrutils/torch/
 * models_gcn.py:57-63, :95-96, :35) and, with the k-major operand forms, all their backward passes.
 *      C[m][n] = act( sum_k A(m,k) * B(n,k) + bias[n] ) (+ C[m][n] if accumulate)
 * a_kmajor = 0: A stored [M rows][K contiguous];            1: A stored [K rows][M contiguous].
 * b_kmajor = 0: B stored [N rows][K contiguous] (nn.Linear weight); 1: B stored [K rows][N contiguous].
 * The strided index of each operand goes through twog_rows_t. All problems of one call share the layout flags and run
 * as grouped launches. `workspace` (may be NULL) enables deterministic split-K for tall-reduction/small-output shapes:
 * caller-owned scratch whose first 16 KB (arrival tickets) are ZERO when it is first handed over and are afterwards left to
 * the library; the k-slices of a tile write partial tiles (slabs) behind the tickets and an ordered-reduce launch adds them
 * in slice order (bit-reproducible) and runs the epilogue. TWOG_GEMM_LA=1: the slice that arrives last at a tile's ticket
 * adds them inside the launch instead (same results bit for bit; measured slower, off by default). One workspace must not
 * be used by launches that can run concurrently (different streams): give each stream its own.
 * =============================================================================================================== */
typedef struct {
    twog_rows_t A, B, C;
    const float* bias;  /* [N] or NULL */
    int32_t M, N, K;
    int32_t act;        /* 0 none, 1 relu */
    int32_t accumulate; /* 0: C = ..., 1: C += ... */
    int32_t batch;      /* >= 1 independent problems of this shape; operand of batch i = ptr + i * *_batch_stride */
    int64_t a_batch_stride, b_batch_stride, c_batch_stride;
    float* a_colsum;            /* NULL, or [M]: a_colsum[m] (+)= sum_k A(m,k) from the same pass over A -- the bias gradient that
                                   belongs to a weight gradient dW = dY^T X (nn.Linear / GRU input projection backward:
                                   pyrutils/torch/models.py:31-36, vhoi/models.py:267-320). Served only where
                                   twog_gemm_colsum_fused() says so; asked for elsewhere the call returns -5, nothing launched */
    int32_t a_colsum_accumulate; /* 0: a_colsum = ..., 1: a_colsum += ... */
    int32_t pad2_;
} twog_gemm_t;
int twog_gemm_f32(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* workspace,
                  size_t workspace_bytes, void* stream);
/* 1 if twog_gemm_f32 with these arguments computes the a_colsum of every problem that asks for one inside the GEMM launch
 * (k-major A and B, the bf16x3 128x128 class: aligned operands, M % 4 == 0, whole 16-deep k-tiles, plain or grouped rows),
 * 0 if the caller has to take the column sums with twog_colsum / twog_colsum_n instead. No launch. */
int twog_gemm_colsum_fused(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* workspace,
                           size_t workspace_bytes);

/* Which kernel variant the calling thread's most recent twog_gemm_f32 chunk (or fused gate launch) selected -- lets a
 * test assert that it exercised the variant it was written for. Bit field: */
#define TWOG_GEMM_CLASS_TILE128 1  /* 128x128 tiles (else 64x64)                        */
#define TWOG_GEMM_CLASS_WAVES8  2  /* 8-wave workgroups (else 4)                        */
#define TWOG_GEMM_CLASS_KG      4  /* k-major operand with (outer, inner) grouped rows  */
#define TWOG_GEMM_CLASS_SPLITK  8  /* deterministic split-K + ordered reduce            */
#define TWOG_GEMM_CLASS_GATE    16 /* gate backward fused into the epilogue             */
#define TWOG_GEMM_CLASS_KSPLIT  32 /* 64x64 tiles, 8 waves, k-split inside the workgroup */
#define TWOG_GEMM_CLASS_GRUFWD  64 /* 64 x (64 units x 3 gates) tiles, GRU forward step in the epilogue */
#define TWOG_GEMM_CLASS_ROWS32  128 /* 32 x 64 tiles (chain launches of small batches)           */
#define TWOG_GEMM_CLASS_XSPLIT  256 /* reduction split over workgroups, combined inside the launch (last arriver) */
#define TWOG_GEMM_CLASS_X3      512 /* on the bf16 matrix cores: fp32 operands split exactly into 3 bf16, 6 products (128x128 class; 64x64 class when K >= 256) */
int twog_gemm_last_class(void);

/* The dependent launches of the recurrent chains (vhoi/models.py:983-1002 frame-level BiGRUs, :785-880 segment loop:
 * few output tiles, K = h ... 3h, every launch waiting for the previous one). `chain_ws` is a caller-owned scratch of
 * twog_chain_workspace_bytes() bytes whose first 16 KB (arrival tickets) are ZERO when it is first handed over and are
 * afterwards left to the library (every launch returns them to zero); with it the library may split the reduction of
 * such a launch over workgroups and combine the partial tiles inside the launch -- each workgroup publishes its partial
 * write-through and draws a ticket, the last arriver adds the partials in slice order (bit-reproducible) and runs the
 * epilogue; no grid barrier, no waiting. NULL: never split. One chain workspace must not be used by launches that can
 * run concurrently (different streams): give each stream its own. */
size_t twog_chain_workspace_bytes(void);
int twog_gemm_f32_chain(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* chain_ws,
                        size_t chain_ws_bytes, void* stream);

/* ===============================================================================================================
 * Geometric-level GCN (pyrutils/torch/models_gcn.py:6-100; called at vhoi/models.py:640-645).
 * x_geo = x_human + 2048 (geometry of human 0, vhoi/models.py:636-639), frame f at x_geo + f*frame_stride, node n,
 * feature c at [n*4 + c]; BatchNorm channel = c*N + n (models_gcn.py:47).
 * =============================================================================================================== */
int twog_gcn_max_nodes(void);
/* norm_data train-mode batch statistics (models_gcn.py:43-49): per-block fp64 partial sums
 * partials[n_blocks][2][4N] (sum, sum of squares; channel order). */
int twog_bn_stats(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes, double* partials, int n_blocks,
                  void* stream);
/* Folds BatchNorm into x^ = a*x + b per channel: ab[2][4N]; mean_invstd[2][4N] is kept for the backward pass.
 * training != 0: batch statistics (biased variance), running stats updated with momentum 0.1 / unbiased variance,
 * num_batches_tracked += 1 (torch BatchNorm1d semantics); training == 0: running statistics. */
int twog_bn_finalize(const double* partials, int n_blocks, int n_frames, int n_nodes, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     int training, float* ab, float* mean_invstd, const float* wq, const float* wk, const float* bq,
                     float* md_out, void* stream);
/* (md_out != NULL: the same launch folds the two similarity projections of compute_similarity, models_gcn.py:95-100, for
 * twog_gcn_fused_fwd: wq, wk [128][64] = get_s.s1 / s2 weights, bq [128] = get_s.s1 bias -> md_out [65][64] = [Mt | d],
 * Mt[n][k] = sum_o wk[o][n] wq[o][k], d[n] = sum_o wk[o][n] bq[o].) */
/* embed layer 1 (models_gcn.py:57-59): e1[(f,n)][64] = relu(W1 (a*x+b) + b1), W1 [64][4]. */
int twog_gcn_embed1_fwd(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes, const float* ab,
                        const float* w1, const float* b1, float* e1, void* stream);
/* Backward of layer 1 given de1 (already ReLU-masked): dw1[64][4], db1[64], dgamma[4N], dbeta[4N].
 * partials: scratch [n_blocks][320 + 8N] floats. */
int twog_gcn_embed1_bwd(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes, const float* ab,
                        const float* mean_invstd, const float* w1, const float* de1, float* partials, int n_blocks,
                        float* dw1, float* db1, float* dgamma, float* dbeta, void* stream);
/* The forward pass up to the aggregation as ONE kernel (csrc/geo_fused.hip): x^ = a*x+b (norm_data folded, :45-50),
 * e1 = relu(W1 x^ + b1) (:57-59, never stored), X = relu(W2 e1 + b2) (:60-63), S = softmax_j(x_i^T M x_j + d . x_j) -- the
 * similarity of compute_similarity (:95-100) with M = Wq^T Wk, d = Wk^T bq folded by the caller into md [65][64] =
 * [Mt | d] -- and Z = S X (:33-34). w1 [64][4], b1 [64], w2 [64][64], b2 [64]. Outputs: x_out [(f,n)][64] (NULL ok: only the
 * backward pass reads it), adj [f][N][N], z [(f,n)][64]. */
int twog_gcn_fused_fwd(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes, const float* ab,
                       const float* w1, const float* b1, const float* w2, const float* b2, const float* md,
                       float* x_out, float* adj, float* z, void* stream);
/* compute_similarity + aggregation per frame (models_gcn.py:95-100, :33-34): qk [(f,n)][256] = [theta(x) | phi(x)],
 * x [(f,n)][64]; s_out [f][N][N] = softmax_j(q_i . k_j) (no 1/sqrt(d)); z [(f,n)][64] = S x. */
int twog_gcn_attn_fwd(const float* qk, const float* x, int n_frames, int n_nodes, float* s_out, float* z,
                      void* stream);
/* Backward: dx_att [(f,n)][64] = S^T dz ; dqk [(f,n)][256] = [dP k | dP^T q], dP = S*(dS - rowsum(dS*S)), dS = dz x^T. */
int twog_gcn_attn_bwd(const float* qk, const float* x, const float* s, const float* dz, int n_frames, int n_nodes,
                      float* dx_att, float* dqk, void* stream);

/* ===============================================================================================================
 * GRU gate math (torch.nn.GRU / GRUCell semantics, gate order r,z,n) for the frame-level BiGRUs
 * (vhoi/models.py:983-1002) and the gated segment-level step h_t = u*GRUCell(x,h) + (1-u)*h (vhoi/models.py:1535-1564).
 * gi / gh are the input / hidden projections (biases included) produced by twog_gemm_f32.
 * =============================================================================================================== */
typedef struct {
    twog_rows_t gi;     /* [rows][3h]  W_ih x + b_ih                                      */
    twog_rows_t gi2;    /* optional second addend of gi (ptr NULL if unused)              */
    twog_rows_t gh;     /* [rows][3h]  W_hh h_prev + b_hh                                 */
    twog_rows_t h_prev; /* [rows][h]   (ptr NULL => zeros)                                */
    twog_rows_t h_out;  /* [rows][h]                                                      */
    twog_rows_t save;   /* [rows][4h]  r, z, n, (W_hn h + b_hn) for backward (ptr NULL ok) */
    const float* u;     /* gate per row or NULL (plain GRU): u[(r/u_inner)*u_ld_outer + (r%u_inner)*u_ld_inner] */
    int64_t u_ld_outer, u_ld_inner;
    int32_t u_inner;
    int32_t rows, hidden;
    int32_t pad_;
} twog_gru_step_t;
int twog_gru_step_fwd(const twog_gru_step_t* steps, int n_steps, void* stream);

typedef struct {
    twog_rows_t dh;      /* [rows][h] gradient wrt h_out of this step                                  */
    twog_rows_t dh2;     /* optional second addend of dh (carried gradient; ptr NULL if unused)         */
    twog_rows_t save;    /* [rows][4h] from the forward step                                            */
    twog_rows_t h_prev;  /* [rows][h] (ptr NULL => zeros)                                               */
    twog_rows_t dgi;     /* out [rows][3h]                                                              */
    twog_rows_t dgh;     /* out [rows][3h]                                                              */
    twog_rows_t dh_prev; /* out [rows][h]: direct path (1-u)*dh + u*dh*z; the W_hh path is added by a GEMM */
    const float* u;      /* as in twog_gru_step_t                                                       */
    float* du;           /* += sum_j dh_j*(gru_j - h_prev_j) per row (atomic add), addressed like u; NULL ok */
    int64_t u_ld_outer, u_ld_inner;
    int32_t u_inner;
    int32_t rows, hidden;
    int32_t dh_prev_accumulate; /* 1: dh_prev += */
} twog_gru_step_bwd_t;
int twog_gru_step_bwd(const twog_gru_step_bwd_t* steps, int n_steps, void* stream);

/* Frame-level bidirectional GRU recurrence (vhoi/models.py:983-1002), all entities of a type batched as rows,
 * both directions and up to 4 entity types advancing in the same launches. */
typedef struct {
    const float* gi;     /* [bs][T][E][6h] W_ih x + b_ih: cols [0,3h) forward direction, [3h,6h) reverse */
    const float* w_hh_f; /* [3h][h] weight_hh_l0          */
    const float* b_hh_f; /* [3h]                          */
    const float* w_hh_r; /* [3h][h] weight_hh_l0_reverse  */
    const float* b_hh_r;
    float* out;          /* [bs][T][E][2h] (forward | reverse), = h_fr of the reference                  */
    float* save;         /* [2][bs][T][E][4h]                                                            */
    float* tmp_gh;       /* scratch [2][bs*E][3h]                                                        */
    float* zeros;        /* [bs*E][h] zeros (initial state)                                              */
    int32_t E;
    int32_t pad_;
} twog_bigru_t;
int twog_bigru_fwd(const twog_bigru_t* types, int n_types, int bs, int T, int hidden, void* chain_ws,
                   size_t chain_ws_bytes, void* stream); /* chain_ws: see twog_gemm_f32_chain (NULL ok) */
/* The same recurrence as ONE persistent launch (csrc/gru_persist.hip): a workgroup per compute unit owns 16 hidden
 * units x 3 gates of one (type, direction) weight for the whole sequence -- split once into bf16 planes, resident in
 * LDS as MFMA fragments -- and a chunk of the type's 16-row tiles, up to four per wave; the steps are ordered inside
 * the launch by agent-scope counters between same-numbered waves. Same inputs, outputs and saved tensors as
 * twog_bigru_fwd (tmp_gh / zeros unused). sync: device memory, >= 1024 uint32, ZERO when the launch starts.
 * twog_bigru_persistent_supported: 0 = shape not served (hidden not 64 / 128 / 256 / 512, or the (type, direction, chunk) x
 * hidden / 16 workgroups do not fit the device), 1 = served, 2 = served and the faster path (at most one tile per
 * wave: small batches).
 * Residency (both persistent entry points, forward and backward): every workgroup of the grid must be resident at once.
 * The launch first asks the runtime (hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units >= grid) and returns
 * TWOG_PERSIST_NOT_RESIDENT (-3) WITHOUT launching when the device cannot hold the grid. What the runtime cannot see --
 * another tenant of the GPU holding compute units -- is caught inside the launch: every wait is bounded
 * (TWOG_PERSIST_SPIN_LIMIT spins, default 2^24 ~ seconds); when a bound runs out the launch sets sync[128] != 0 and every
 * wave leaves (no trap, the context survives). After the launch the CALLER reads sync[128]: non-zero means the outputs are
 * incomplete and the pass must be re-run with the launch-per-step entry point (twog_bigru_fwd / twog_bigru_bwd: same
 * buffers, written in place, so the re-run is idempotent). kernels.py does exactly that. */
#define TWOG_PERSIST_NOT_RESIDENT (-3)
int twog_bigru_persistent_supported(const twog_bigru_t* types, int n_types, int bs, int hidden);
int twog_bigru_fwd_persistent(const twog_bigru_t* types, int n_types, int bs, int T, int hidden, void* sync, void* stream);

typedef struct {
    const float* d_out;  /* [bs][T][E][2h] gradient wrt out                                              */
    const float* save;
    const float* out;
    const float* w_hh_f;
    const float* w_hh_r;
    float* d_gi;         /* out [bs][T][E][6h] gradient wrt gi                                           */
    float* d_gh;         /* out [bs][T][E][6h] gradient wrt (W_hh h_prev + b_hh)                         */
    float* carry;        /* scratch [2][bs*E][h]                                                         */
    int32_t E;
    int32_t pad_;
} twog_bigru_bwd_t;
int twog_bigru_bwd(const twog_bigru_bwd_t* types, int n_types, int bs, int T, int hidden, void* chain_ws,
                   size_t chain_ws_bytes, void* stream);
/* Backward through time as ONE persistent launch, the small-batch form (at most one 16-row tile per wave: the shapes
 * for which twog_bigru_bwd_persistent_supported returns 2): a workgroup owns 16 hidden units of one (type, direction),
 * keeps the 3h x 16 slice of W_hh that produces THEIR carried gradient in LDS as bf16x3 MFMA fragments and the carried
 * gradient itself in registers; per step it runs the gate backward of its units, publishes their d_gh columns
 * (write-through) and multiplies the complete d_gh rows -- all workgroups' columns -- with its slice. Same outputs as
 * twog_bigru_bwd (carry unused). sync: device memory, >= 1024 uint32, ZERO when the launch starts. */
int twog_bigru_bwd_persistent_supported(const twog_bigru_bwd_t* types, int n_types, int bs, int hidden);
int twog_bigru_bwd_persistent(const twog_bigru_bwd_t* types, int n_types, int bs, int T, int hidden, void* sync, void* stream);

/* ===============================================================================================================
 * Fusion-level attention message passing (vhoi/models.py:1004-1475, :1693-1754) for message_type 'v2', granularity
 * 'v1' (generic), aggregation 'att', attention styles 'v2'/'v3' (dot / scaled dot product).
 * One instance = one (clip, frame) at frame level, one clip at segment level; entities: H humans, O objects, plus the
 * single-sender geometry relations. Per instance and relation with receiver r and senders s:
 *      w_rs = softmax_s( q_r . k_s * scale ) over non-virtual senders (objects_mask), NaN -> 0 (:1750-1753)
 *      out_r = [recv_mask_r] * sum_s w_rs * msg_s
 * relations: hh (human<-other humans), oh (human<-objects), sh (human<-geometry), ho (object<-humans),
 *            so (object<-geometry), oo (object<-other objects). A relation is off when its msg ptr is NULL.
 * =============================================================================================================== */
typedef struct {
    twog_rows_t feat_h;  /* [inst*H][D] query/key features of humans */
    twog_rows_t feat_o;  /* [inst*O][D] of objects                   */
    twog_rows_t msg_hh, msg_ho; /* [inst*H][hidden] messages sent BY humans (to humans / to objects)          */
    twog_rows_t msg_oh, msg_oo; /* [inst*O][hidden] messages sent BY objects (to humans / to objects)         */
    twog_rows_t msg_so, msg_sh; /* [inst][hidden]   messages sent by the geometry node (to objects / humans)  */
    twog_rows_t out_hh, out_oh, out_sh; /* [inst*H][hidden] received by humans  */
    twog_rows_t out_ho, out_so, out_oo; /* [inst*O][hidden] received by objects */
    const float* obj_mask; /* [clips][O], clip = inst / inst_per_clip; NULL => all real */
    float* att;            /* saved weights [inst][H*H + H*O + O*H + O*O] (hh, oh, ho, oo); NULL ok in forward */
    int32_t n_inst, inst_per_clip, H, O, D, hidden;
    float scale;           /* 1 ('v2') or 1/sqrt(D) ('v3', :1745) */
    int32_t recv_mask_ho;  /* 1: out_ho and out_so are multiplied by obj_mask of the receiver (frame level,
                              vhoi/models.py:720,729); 0: not (segment level, :841-843) */
} twog_attn_t;
int twog_attn_fwd(const twog_attn_t* a, int n, void* stream);
int twog_attn_limits(int* max_h, int* max_o);

typedef struct {
    twog_attn_t f;       /* the forward descriptor (features, messages, saved att; out_* unused) */
    twog_rows_t dout_hh, dout_oh, dout_sh, dout_ho, dout_so, dout_oo; /* incoming gradients wrt out_*         */
    twog_rows_t dmsg_hh, dmsg_ho, dmsg_oh, dmsg_oo, dmsg_so, dmsg_sh; /* out: gradient wrt sender messages    */
    twog_rows_t dfeat_h, dfeat_o; /* out: gradient wrt features [..][D]                                       */
    const float* dw_extra;        /* optional [n_inst][natt]: added to dL/d(weights) before the softmax backward (the
                                     part of the gradient that reaches the weights outside this kernel: twog_ssp_bwd) */
    int32_t dfeat_accumulate;     /* 1: dfeat += */
    int32_t relu_mask_dmsg;       /* 1: dmsg *= (msg > 0), i.e. gradient wrt the pre-ReLU activation of the message MLP */
} twog_attn_bwd_t;
int twog_attn_bwd(const twog_attn_bwd_t* a, int n, void* stream);

/* ===============================================================================================================
 * Segment-level gated bidirectional recurrence with message passing (vhoi/models.py:785-880, :1535-1564,
 * :1051, :1145, :1239, :1334). Buffers are (clip, time, entity)-ordered; [2] = direction (0 forward, 1 backward).
 * nsh/nso = number of enabled sender MLPs on human/object states ((hh|ho) / (oh|oo)); nmh/nmo = number of message
 * blocks received by a human/object ((hh|oh) / (ho|oo)).
 * =============================================================================================================== */
typedef struct {
    int32_t bs, T, H, O, hidden;
    int32_t msg_segment;                    /* message_segment */
    int32_t rel_hh, rel_ho, rel_oh, rel_oo; /* enabled relations */
    float att_scale;
    int32_t pad_;
    const float* gi_h;     /* [bs][T][H][6h] frame part of W_ih x + b_ih: fcell cols [0,3h), bcell [3h,6h) */
    const float* gi_o;     /* [bs][T][O][6h] */
    const float* u_h;      /* hard gates [bs][T][H] */
    const float* u_o;      /* [bs][T][O] */
    const float* obj_mask; /* [bs][O] */
    const float* w_hh_h[2]; const float* b_hh_h[2]; /* human_segment_rnn_{f,b}cell.weight_hh [3h][h], bias_hh */
    const float* w_hh_o[2]; const float* b_hh_o[2];
    const float* w_ihm_h[2]; /* &weight_ih[0][first message column] of the human cells, row stride ld_ih_h */
    const float* w_ihm_o[2];
    int64_t ld_ih_h, ld_ih_o;
    const float* w_smsg_h; const float* b_smsg_h; /* packed sender MLPs on human states [(nsh*h)][h], [(nsh*h)] */
    const float* w_smsg_o; const float* b_smsg_o; /* packed sender MLPs on object states */
    float* hs_h;    /* out [bs][T][H][2h] (forward | backward states) */
    float* hs_o;    /* out [bs][T][O][2h] */
    float* save_h;  /* [2][bs][T][H][4h] */
    float* save_o;  /* [2][bs][T][O][4h] */
    float* msrc_h;  /* [2][bs][T][H][nsh*h] post-ReLU sender messages from humans */
    float* msrc_o;  /* [2][bs][T][O][nso*h] */
    float* mg_h;    /* [2][bs][T][H][nmh*h] aggregated messages received by humans */
    float* mg_o;    /* [2][bs][T][O][nmo*h] */
    float* att;     /* [2][T][bs][H*H + 2*H*O + O*O] */
    float* tmp_gim_h; /* scratch [2][bs*H][3h] */
    float* tmp_gim_o; /* scratch [2][bs*O][3h] */
    float* tmp_gh_h;  /* scratch [2][bs*H][3h] */
    float* tmp_gh_o;  /* scratch [2][bs*O][3h] */
    float* zeros;     /* [bs*max(H,O)][h] zeros */
} twog_segrnn_t;
int twog_segrnn_fwd(const twog_segrnn_t* desc, void* chain_ws, size_t chain_ws_bytes, void* stream); /* chain_ws: see twog_gemm_f32_chain */
/* The same forward recurrence as ONE persistent launch (csrc/seg_persist.hip; small batches): per (direction, clip chunk)
 * and slice of 16 hidden units four workgroups -- P1a / P1b: sender MLPs, attention weights (the chunk's Gram matrix on the
 * matrix cores), aggregated messages received by humans / objects and W_hh h_prev; P2h / P2o: W_ih[:, messages] on the
 * complete message rows, gate math, h_t -- two in-launch hand-offs per step (agent-scope counters, write-through stores,
 * sc1 loads). Same inputs, outputs and saved tensors as twog_segrnn_fwd (tmp_gim_* / zeros unused, tmp_gh_* carries
 * W_hh h_prev between the roles). twog_segrnn_persistent_supported: 2 = served on the current device (message_segment with
 * all four relations, hidden 64 / 128 / 256 / 512, a chunk of clips fits one 16-row tile of humans and at most two of
 * objects, 2 x chunks x 4 x hidden / 16 workgroups fit the device), 0 = not served. sync: device memory,
 * twog_segrnn_persistent_sync_bytes(), ZERO at launch; its last 128-byte line is the error word (uint32 index
 * bytes / 4 - 32). Residency and soft failure: exactly as twog_bigru_fwd_persistent (TWOG_PERSIST_NOT_RESIDENT; error word
 * != 0 after the launch -> re-run the pass with twog_segrnn_fwd on the same buffers). */
int twog_segrnn_persistent_supported(const twog_segrnn_t* desc);
size_t twog_segrnn_persistent_sync_bytes(void);
int twog_segrnn_fwd_persistent(const twog_segrnn_t* desc, void* sync, void* stream);

typedef struct {
    const float* d_hs_h; /* [bs][T][H][2h] gradient wrt hs_h */
    const float* d_hs_o;
    float* d_gi_h;  /* out [bs][T][H][6h] gradient wrt the full W_ih x + b_ih (frame + message part) */
    float* d_gi_o;
    float* d_gh_h;  /* out [bs][T][H][6h] gradient wrt W_hh h_prev + b_hh */
    float* d_gh_o;
    float* d_u_h;   /* += [bs][T][H] gradient wrt the hard gates (caller zeroes) */
    float* d_u_o;
    float* d_pre_h; /* out [2][bs][T][H][nsh*h] gradient wrt the pre-ReLU sender-MLP activations */
    float* d_pre_o;
    float* carry_h; /* scratch [2][bs*H][h] */
    float* carry_o; /* scratch [2][bs*O][h] */
    float* tmp_dmg_h; /* scratch [2][bs*H][nmh*h] */
    float* tmp_dmg_o; /* scratch [2][bs*O][nmo*h] */
    float* trash;     /* scratch [bs*max(H,O)][h] */
    /* scratch [2][T][16][bs*H] / [2][T][16][bs*O] or NULL. When both are given, the gate backward of chain step s-1 runs
     * in the epilogue of the last GEMM of step s (no separate gate launches) and parks its per-row partial sums of
     * d_u here; they are added into d_u_* in fixed order after the loop (bit-reproducible). NULL: separate launches. */
    float* du_part_h;
    float* du_part_o;
} twog_segrnn_bwd_t;
int twog_segrnn_bwd(const twog_segrnn_t* desc, const twog_segrnn_bwd_t* bdesc, void* chain_ws, size_t chain_ws_bytes,
                    void* stream);
/* Backward through time as ONE persistent launch (csrc/seg_persist.hip; the shapes twog_segrnn_persistent_supported
 * serves): per slice of 16 columns Q2h / Q2o keep the carried state gradient of their units in registers, run the gate
 * backward and publish d_gi / d_gh columns; Q1h / Q1o turn the complete d_gi rows into the gradients of the aggregated
 * messages, and those -- with the SAVED attention weights -- into d_pre columns and their slice's share of the score
 * gradients; two in-launch hand-offs per step. Same outputs as twog_segrnn_bwd (carry_*, tmp_dmg_*, trash, du_part_* unused).
 * scratch: twog_segrnn_bwd_persistent_scratch_bytes(desc) bytes of device memory, contents undefined. sync, residency and
 * soft failure as for twog_segrnn_fwd_persistent (after a launch that gave up d_u_* are untouched). */
size_t twog_segrnn_bwd_persistent_scratch_bytes(const twog_segrnn_t* desc);
int twog_segrnn_bwd_persistent(const twog_segrnn_t* desc, const twog_segrnn_bwd_t* bdesc, void* scratch, size_t scratch_bytes,
                               void* sync, void* stream);
/* hipGraph cache of the time loops (twog_bigru_*, twog_segrnn_*): number of captured loops and of hash-bucket hits whose
 * descriptor bytes differed (each was resolved by the byte compare; see csrc/graph_cache.h). Diagnostics / tests. */
int twog_graph_cache_stats(int64_t* entries, int64_t* collisions);

/* ===============================================================================================================
 * Optional position features and the less common gate strategies (disabled in every shipped configuration).
 * twog_pos_embed_fwd: out[(b,t,e)][hidden] = relu(w * s + b) (positional_encoding_style 'e': time_position_mlp /
 *   segment_length_mlp = build_mlp([1, h], ['relu']), vhoi/models.py:259-264) or the periodic embedding
 *   [sin(s / w_k) | cos(s / w_k)], w_k = 1e4^(k / (h/2 - 1)) (make_periodic_embedding, :1778-1794). The scalar s of row
 *   (b,t,e) is s[row] when `s` is given, else the time feature (t + 1) (/ steps[b] when `divide`;
 *   _assemble_time_tensor, :935-952). s_out (optional) receives the scalars [bs*T*E] for the backward pass.
 * twog_periodic_embed_bwd: ds[row] of the periodic embedding (the 'e' style uses a GEMM and twog_colsum).
 * twog_seglen_fwd / _bwd: _assemble_segment_length_tensor (:954-981): per (clip, entity) scan over time of the hard
 *   gates u [bs][T][E]; the backward ADDS into du.
 * twog_mul: out[i] = a[i] * b[i] (+ out[i] if accumulate): 'conditional_on_human' object gates (:1531-1532).
 * twog_scale_rows: x[r][:] *= s[r] (receiver mask of a relational message, :720/:729).
 * =============================================================================================================== */
int twog_pos_embed_fwd(const float* s, const float* steps, int bs, int T, int E, int divide, const float* w,
                       const float* b, int periodic, int hidden, twog_rows_t out, float* s_out, void* stream);
int twog_periodic_embed_bwd(twog_rows_t dout, const float* s, int rows, int hidden, float* ds, void* stream);
int twog_seglen_fwd(const float* u, const float* steps, int bs, int T, int E, int divide, float* s_out, void* stream);
int twog_seglen_bwd(const float* u, const float* steps, int bs, int T, int E, int divide, const float* ds, float* du,
                    void* stream);
int twog_mul(const float* a, const float* b, float* out, int64_t n, int accumulate, void* stream);
int twog_scale_rows(twog_rows_t x, const float* s, int rows, int cols, void* stream);

/* ===============================================================================================================
 * Sender-side projection of aggregated messages. The objects' segment-level GRUCells read
 * cat[h_f, m_ho, m_so, m_oo] (vhoi/models.py:748) through W_ih; m_ho[k] = mask_k sum_h att[k][h] msg_h (:1099-1143, :720)
 * and m_so[k] = mask_k msg_s (:1384-1429, :729) are linear in the senders' messages, so the caller projects the H + 1
 * sender rows of a frame (GEMM) and these kernels scatter / gather them over the O receivers:
 *   fwd: gi[(inst,k)][:] += mask[clip][k] * ( sum_h att[inst][att_off + k*H + h] * ph[(inst,h)][:] + ps[inst][:] )
 *   bwd: qh[(inst,h)][:] = sum_k mask_k att[k][h] dgi[(inst,k)][:]   qs[inst][:] = sum_k mask_k dgi[(inst,k)][:]
 *        dw[inst][att_off + k*H + h] = mask_k <dgi[(inst,k)], ph[(inst,h)]>   (feeds twog_attn_bwd's dw_extra)
 * ph / ps / qh / qs may be NULL (relation off). cols % 4 == 0, H <= 4, O <= 16.
 * =============================================================================================================== */
int twog_ssp_fwd(float* gi, const float* ph, const float* ps, const float* att, const float* mask, int n_inst,
                 int inst_per_clip, int H, int O, int cols, int natt, int att_off, void* stream);
/* gather alone, with arbitrary placement: qh[(inst,h)][c] = sum_k att(inst)[att_off + k*H + h] * dgi[(inst*O + k)*dgi_ld + c],
 * att(inst) = att + clip*att_ld_clip + frame*att_ld_frame. Used for the segment-level human->object message block of the
 * objects' W_ih gradient (its weights are stored [time][clip][natt], the two directions share the d_gi rows). */
int twog_ssp_gather(const float* dgi, int64_t dgi_ld, const float* att, int64_t att_ld_clip, int64_t att_ld_frame,
                    int att_off, float* qh, int n_inst, int inst_per_clip, int H, int O, int cols, void* stream);
int twog_ssp_bwd(const float* dgi, const float* ph, const float* att, const float* mask, float* qh, float* qs, float* dw,
                 int n_inst, int inst_per_clip, int H, int O, int cols, int natt, int att_off, void* stream);

/* ===============================================================================================================
 * General message passing of ONE relation (receiver set R, sender set S per instance): the message / aggregation forms
 * of the reference beyond the shipped one (twog_attn_* covers sender-only messages + dot-product attention for the
 * four entity relations at once): relational messages (compute_relational_message, vhoi/models.py:1667-1690),
 * 'specific' granularity (:1712-1713), attention styles 'concat' / 'general' (:1739-1746), distance-based attention
 * (:1757-1775) and mean pooling (:1034-1037). Linear layers on cat[receiver, sender] are split by the caller into a
 * receiver and a sender projection (GEMMs); this kernel combines them per pair:
 *      out[r] = recv_mask[r] * sum_s w[r][s] * M(r, s)
 *      M(r, s) = msg[s]                              (TWOG_REL_MSG_SENDER)
 *              = relu(p_r[r] + p_s[s])               (TWOG_REL_MSG_PAIR)
 *      w[r][.] = send_mask[s]                        (TWOG_REL_SUM: masked sum, relational)
 *              = softmax_s(scale <q[r], k[s]> + score_bias [relu])   (TWOG_REL_DOT; 'general': k = A k precomputed)
 *              = softmax_s(relu(a_r[r] + c_s[s]))    (TWOG_REL_ADDITIVE: 'concat')
 *              = softmax_s(1 / (dist[r][s] + 1e-7)), dist == 0 excluded     (TWOG_REL_DISTANCE)
 *              = 1 / max(#valid senders, 1)          (TWOG_REL_MEAN)
 * over the valid senders (send_mask != 0, and s != r when exclude_self); no valid sender -> all-zero weights (the
 * reference's NaN -> 0). Rows of one instance must be equally strided (inner <= 1 or inner == R resp. S).
 * =============================================================================================================== */
#define TWOG_REL_SUM 0
#define TWOG_REL_DOT 1
#define TWOG_REL_ADDITIVE 2
#define TWOG_REL_DISTANCE 3
#define TWOG_REL_MEAN 4
#define TWOG_REL_MSG_SENDER 0
#define TWOG_REL_MSG_PAIR 1
typedef struct {
    twog_rows_t q, k;        /* [inst*R][D], [inst*S][D]      (TWOG_REL_DOT) */
    twog_rows_t msg;         /* [inst*S][hidden]              (TWOG_REL_MSG_SENDER) */
    twog_rows_t p_r, p_s;    /* [inst*R][hidden], [inst*S][hidden] (TWOG_REL_MSG_PAIR) */
    twog_rows_t out;         /* [inst*R][hidden] */
    const float* a_r;        /* [inst*R] (TWOG_REL_ADDITIVE; the layer's bias folded in) */
    const float* c_s;        /* [inst*S] */
    const float* dist;       /* element (inst, r, s) at inst*dist_ld_inst + r*dist_ld_r + s*dist_ld_s (TWOG_REL_DISTANCE) */
    int64_t dist_ld_inst, dist_ld_r, dist_ld_s;
    const float* send_mask;  /* [clip][S] or NULL */
    const float* recv_mask;  /* [clip][R] or NULL */
    float* att;              /* out (fwd): weights [inst][R][S], or NULL */
    const float* score_bias; /* device scalar added to the dot score (the Bilinear bias of 'general'), or NULL */
    float scale;
    int32_t score_mode, msg_mode, relu_scores, exclude_self;
    int32_t n_inst, inst_per_clip, R, S, D, hidden;
    int32_t pad_;
} twog_relation_t;
int twog_relation_limits(void); /* max R, S */
int twog_relation_fwd(const twog_relation_t* rel, void* stream);
typedef struct {
    twog_relation_t f;
    twog_rows_t dout;        /* [inst*R][hidden] gradient wrt out */
    twog_rows_t dmsg;        /* out [inst*S][hidden] (MSG_SENDER; NULL = not needed) */
    twog_rows_t dp_r, dp_s;  /* out (MSG_PAIR) */
    twog_rows_t dq, dk;      /* out [..][D] (DOT; NULL = not needed) */
    float* da_r;             /* out [inst*R] (ADDITIVE) */
    float* dc_s;             /* out [inst*S] */
    float* dscore_sum;       /* out [inst]: sum over the pairs of d(raw score) = d score_bias per instance (DOT), or NULL */
    int32_t dq_accumulate, dk_accumulate; /* add into dq / dk instead of overwriting */
    int32_t relu_mask_dmsg;  /* dmsg *= (msg > 0): the sender MLP's ReLU folded in */
    int32_t pad_;
} twog_relation_bwd_t;
int twog_relation_bwd(const twog_relation_bwd_t* rel, void* stream);
/* n relations per call, several descriptors per launch: the general segment-level loop (vhoi/models.py:785-880 with the
 * message forms above) issues all relations of both directions of a chain step together. The descriptors of one call
 * must not ACCUMULATE into the same rows (dq / dk with *_accumulate set): they run concurrently. */
int twog_relation_fwd_n(const twog_relation_t* rels, int n, void* stream);
int twog_relation_bwd_n(const twog_relation_bwd_t* rels, int n, void* stream);

/* ===============================================================================================================
 * Segment-boundary gates (vhoi/models.py:1477-1533, :1620-1627; pyrutils/torch/distributions.py:4-53) with
 * discrete_networks_num_layers == 1: p = sigmoid(w . [column blocks of the entity row] + b); 'gs': Gumbel-sigmoid
 * with PRE-DRAWN noise, 'st' (noise == NULL): straight-through; hard = soft > thr; last step forced to 1 (:701-702).
 * Rows are (b, t, e)-ordered; hard/soft/p_save are [bs][T][E].
 * =============================================================================================================== */
typedef struct {
    twog_rows_t x;        /* entity rows; gate input = concat of n_seg column blocks of width `hidden` */
    int32_t seg_col[8];   /* starting column in x of weight block i */
    int32_t n_seg, hidden;
    const float* w;       /* [n_seg*hidden] */
    const float* b;       /* [1] or NULL    */
    const float* noise;   /* Gumbel noise [T][noise_entities][bs][2] (reference call order) or NULL ('st') */
    float* hard;
    float* soft;
    float* p_save;
    int32_t bs, T, E;
    int32_t noise_entities, noise_offset; /* this type's entity e uses noise slot noise_offset + e */
    int32_t force_last;   /* 1: hard[:, T-1, :] = 1 and no gradient through it */
    float threshold;
    int32_t pad_;
} twog_gate_t;
int twog_gate_fwd(const twog_gate_t* g, void* stream);
/* dlogit[row] = (d_soft[row] + d_hard[row]*st_mask[row]) * d soft / d logit. d_soft, d_hard, st_mask may be NULL
 * (st_mask NULL => 1; with force_last the last step's d_hard is dropped). */
int twog_gate_bwd(const twog_gate_t* g, const float* d_hard, const float* d_soft, const float* st_mask, float* dlogit,
                  void* stream);
/* dst[r][c] += s[r] * v[c]  (gate input gradient, one call per column block) */
int twog_rank1_update(twog_rows_t dst, const float* s, const float* v, int rows, int cols, void* stream);
/* out[c] (+)= sum_r rowscale[r] * x[r][c] (rowscale NULL => 1): bias gradients and gate weight gradients.
 * Deterministic two-pass reduction; partials: scratch [n_blocks][cols]. */
int twog_colsum(twog_rows_t x, const float* rowscale, int rows, int cols, float* out, int accumulate,
                float* partials, int n_blocks, void* stream);
/* Several column sums in one pair of launches (the bias gradients of a backward stage collected by the host: 45 per training
 * step). Problem i: out[c] (+)= sum_r rowscale[r] * x[r][c] with the arithmetic -- and so, bit for bit, the results -- of its own
 * twog_colsum call with the default row slicing. partials: scratch of twog_colsum_n_partial_floats(ops, n) floats (the chunks of
 * TWOG_COLSUM_MAX problems of one call run one after the other and share it). */
#define TWOG_COLSUM_MAX 16
typedef struct {
    twog_rows_t x;
    const float* rowscale;  /* [rows] or NULL */
    float* out;             /* [cols] */
    int32_t rows, cols;
    int32_t accumulate;
    int32_t pad_;
} twog_colsum_t;
size_t twog_colsum_n_partial_floats(const twog_colsum_t* ops, int n);
int twog_colsum_n(const twog_colsum_t* ops, int n, float* partials, size_t partial_floats, void* stream);

/* filter_soft_decisions (vhoi/models.py:1637-1664): local-maximum filter on soft gates [bs][T][E];
 * grad_mask = d hard / d soft (0 or 1) under the reference's straight-through + clamp semantics. */
int twog_filter_fwd(const float* soft, float* hard, float* grad_mask, int bs, int T, int E, float threshold,
                    void* stream);

/* reorder_hidden_states (vhoi/models.py:1567-1586) without the host sync: out[b,t,e,:] = hx[b,idx,e,:], idx = first
 * frame >= t whose gate is non-zero (t itself if none). hx/out: [bs][T][E][cols], gate: [bs][T][E]. */
int twog_reorder_fwd(const float* hx, const float* gate, float* out, int bs, int T, int E, int cols, void* stream);
int twog_reorder_bwd(const float* dout, const float* gate, float* dhx, int bs, int T, int E, int cols, void* stream);

/* Label-head epilogue (vhoi/models.py:909-917): log_softmax over classes of logits [(b,t,e)][C] + the
 * permute(0,3,1,2) store to [bs][C][T][E]; backward returns dlogits [(b,t,e)][C]. */
int twog_logsoftmax_permute_fwd(const float* logits, float* out, int bs, int T, int E, int C, void* stream);
int twog_logsoftmax_permute_bwd(const float* out, const float* dout, float* dlogits, int bs, int T, int E, int C,
                                void* stream);

/* Elementwise helpers between GEMMs in the backward pass: dx = dy * (y > 0) (ReLU'), dst += src. */
int twog_relu_bwd(twog_rows_t dy, twog_rows_t y, twog_rows_t dx, int rows, int cols, void* stream);
int twog_add_rows(twog_rows_t src, twog_rows_t dst, int rows, int cols, void* stream);
/* Several of these small row-wise operations in one launch (same callers as twog_relu_bwd / twog_add_rows /
 * twog_rank1_update, issued once per dependency level by the general segment-level loop). No two operations of one call
 * may write the same rows. */
enum { TWOG_ROWOP_RELU_BWD = 0, TWOG_ROWOP_ADD = 1, TWOG_ROWOP_RANK1 = 2 };
typedef struct {
    twog_rows_t a, b, dst; /* RELU_BWD: dst = a * (b > 0);  ADD: dst += a;  RANK1: dst[r][c] += s[r] * v[c] */
    const float* s;        /* RANK1: [rows] */
    const float* v;        /* RANK1: [cols] */
    int32_t kind, rows, cols, pad_;
} twog_rowop_t;
int twog_rowops(const twog_rowop_t* ops, int n_ops, void* stream);

/* ===============================================================================================================
 * Replay of a recorded call sequence for the further steps of a loop (the general segment-level loop,
 * vhoi/models.py:785-880 with the message forms of twog_relation_*): the host composes two consecutive chain steps a, b
 * as lists of calls (kind, flags, descriptor array in HOST memory) without issuing them; every 64-bit word w of every
 * descriptor of step a + k is  w(a) + k * (w(b) - w(a))  -- per-step operands live in consecutive slots of per-step
 * buffers, everything else is equal in a and b. Runs steps k_begin <= k < k_end in order (k = 0 is step a itself);
 * the calls are exactly the entry points above. workspace: the split-K scratch handed to twog_gemm_f32.
 * =============================================================================================================== */
enum { TWOG_TAPE_GEMM = 0, TWOG_TAPE_RELATION_FWD = 1, TWOG_TAPE_RELATION_BWD = 2, TWOG_TAPE_GRU_STEP_FWD = 3,
       TWOG_TAPE_GRU_STEP_BWD = 4, TWOG_TAPE_ROWOPS = 5 };
typedef struct {
    int32_t kind;     /* TWOG_TAPE_* */
    int32_t n;        /* descriptors in the call */
    int32_t flags;    /* GEMM: bit 0 a_kmajor, bit 1 b_kmajor */
    int32_t pad_;
    const void* desc; /* host array of n descriptors of the kind's type */
} twog_tape_entry_t;
int twog_tape_run(const twog_tape_entry_t* step_a, const twog_tape_entry_t* step_b, int n_entries, int k_begin, int k_end,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Fused Adam on one flat fp32 parameter buffer (torch.optim.Adam semantics; reference train.py:39). The gradient is
 * multiplied by grad_scale first (1/world_size after a sum all-reduce). */
/* Zero fill of `nbytes` bytes at `p` (any alignment) on `stream`: the step's workspace / gradient clears (torch.zeros and
 * Tensor.zero_() on the reference's path, e.g. optimizer.zero_grad(), pyrutils/torch/train_utils.py:146). */
int twog_fill_zero(void* p, size_t nbytes, void* stream);
/* Diagnostics (tests): n_blocks workgroups of 256 threads holding lds_bytes of LDS each spin for `usec` microseconds --
 * the co-tenant a persistent launch must survive (see TWOG_PERSIST_NOT_RESIDENT). No reference counterpart. */
int twog_debug_occupy(int n_blocks, int lds_bytes, int usec, void* stream);

/* Persistent launches checked at the end of a pass (see twog_bigru_fwd_persistent): `words` are the host-pinned copies of
 * their error words (device-readable), filled by asynchronous copies enqueued before this call. If any is non-zero, every
 * `out[i]` (n[i] floats) is overwritten with NaN -- the pass's results can then not be consumed silently while the host
 * check (which raises) is left for the end of the backward pass. Host plumbing; no counterpart in the reference. */
#define TWOG_GUARD_MAX 16
typedef struct {
    const uint32_t* words[TWOG_GUARD_MAX];
    float* out[TWOG_GUARD_MAX];
    int64_t n[TWOG_GUARD_MAX];
    int32_t n_words, n_out;
} twog_guard_t;
int twog_guard_outputs(const twog_guard_t* g, void* stream);

/* A stream restricted to n_cus compute units (hipExtStreamCreateWithCUMask; the low n_cus mask bits = n_cus / 8 CUs on each
 * XCD). Host plumbing of this library's own backward pass (weight-gradient GEMMs beside a launch-per-step recurrence that
 * leaves those CUs idle); no counterpart in the reference. Returns 0, or < 0 when the runtime refuses. */
int twog_stream_create_masked(int n_cus, void** stream_out);
/* A stream of the LOWEST priority the device offers (hipDeviceGetStreamPriorityRange / hipStreamCreateWithPriority): the side
 * stream of the same backward pass (TWOG_SIDE_PRIORITY=low) -- its GEMM workgroups then yield free compute units to the
 * launch-per-step recurrence on the caller's stream. Returns 0, or < 0 when the runtime refuses. */
int twog_stream_create_low_priority(void** stream_out);
int twog_stream_destroy(void* stream);
/* n_blocks <= TWOG_COPY_MAX copies dst[i] = src[i], i < n floats, of contiguous fp32 blocks in ONE launch. The host uses it
 * to rebuild, at EVERY forward call, the packed operands the time loops read (w_smsg_* / b_smsg_* of twog_segrnn_t: the
 * reference applies the four segment-level sender MLPs one by one, vhoi/models.py:1051-1098, :1145-1190, :1239-1285,
 * :1334-1383; here two of them share a GEMM). Nothing derived from a parameter is kept across calls. */
#define TWOG_COPY_MAX 16
typedef struct {
    const float* src;
    float* dst;
    int64_t n;
} twog_copy_t;
int twog_copy_blocks(const twog_copy_t* blocks, int n_blocks, void* stream);
int twog_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* Adjacency attention of Geo_gcn (compute_similarity + s.matmul(x), pyrutils/torch/models_gcn.py:86-100, :30-34) on the
 * matrix cores with the theta / phi projections folded: md = [65][64] = Mt (Mt[n][k] = sum_o Wk[o][n] Wq[o][k]) followed
 * by d = Wk^T bq; per frame P = X M + d, adj = softmax_j(P X^T), z = adj X (identical to softmax(theta phi^T) X: the
 * dropped terms are constant along j). x: [n_frames*N][64], adj: [n_frames][N][N], z: [n_frames*N][64].
 * Backward: dx_att = d/dX through adj and z (X's own ReLU/W2 path excluded), partials[n_blocks][65*64] = per-workgroup
 * sums of (dMt | dd), n_blocks = twog_gcn_attn2_bwd_blocks(n_frames); the caller column-sums them (twog_colsum). */
int twog_gcn_attn2_fwd(const float* x, const float* md, int n_frames, int n_nodes, float* adj, float* z, void* stream);
int twog_gcn_attn2_bwd_blocks(int n_frames);
int twog_gcn_attn2_bwd(const float* x, const float* md, const float* adj, const float* dz, int n_frames, int n_nodes,
                       float* dx_att, float* partials, int n_blocks, void* stream);

/* ===============================================================================================================
 * Multi-task loss: the criterion the training loop applies to the model's output list
 * (vhoi/losses.py:8-70 select_loss; pyrutils/torch/losses.py:7-51 multi_task_loss, binary_cross_entropy_loss,
 * budget_loss; F.nll_loss with ignore_index, reduction 'mean'). All terms of the list run in ONE forward and ONE
 * backward launch; sums are fp64 and ordered (deterministic); nothing syncs with the host (the reference calls
 * mask.sum().item() per term).
 *   kind 0  NLL:    input [outer][C][inner] log-probabilities, target int64 [outer][inner];
 *                   loss = w * sum_{target != ignore} -input[o][target][i] / count            (count 0 -> NaN, as torch)
 *   kind 1  BCE:    input, target float [inner] (outer = C = 1); m = target != ignore;
 *                   loss = w * sum_m -(t*max(log x,-100) + (1-t)*max(log(1-x),-100)) / count   (count 0 -> 0)
 *   kind 2  budget: loss = w * sum_m x / count                                                 (count 0 -> 0)
 * stats[term] = {sum, count} is written by the forward call and read by the backward call, which stores
 * d(loss_total)/d(input) = dlosses[term] * d(loss_term)/d(input) into terms[i].dinput (skipped where dinput is NULL).
 * =============================================================================================================== */
typedef struct {
    int32_t kind;
    int32_t n_classes;
    int64_t outer, inner;
    const float* input;
    const void* target;
    float* dinput;
    float weight;
    float ignore_value;
} twog_loss_t;
#define TWOG_LOSS_MAX_TERMS 16
#define TWOG_LOSS_BLOCKS 64
/* partials: workspace of n_terms * TWOG_LOSS_BLOCKS * 2 doubles; losses: [n_terms] floats. */
int twog_multitask_loss_fwd(const twog_loss_t* terms, int n_terms, double* partials, double* stats, float* losses,
                            void* stream);
int twog_multitask_loss_bwd(const twog_loss_t* terms, int n_terms, const double* stats, const float* dlosses,
                            void* stream);

/* ===============================================================================================================
 * Inference post-processing on the device.
 * twog_predict_labels: predict.py:64-70 (repeat_interleave by `downsampling` along time, match_shape :95-116 to T_out
 *   steps) + :195-201 (argmax over classes; ties -> first index): logp [bs][C][T][E] -> labels int64 [bs][T_out][E],
 *   labels[b][t][e] = argmax_c logp[b][c][min(t / downsampling, T - 1)][e].
 * twog_f1_at_k: pyrutils/metrics.py:7-81. y_true / y_pred int64 [n_seq][n_steps]; steps with y_true == ignore_value
 *   are dropped from both (use_ignore); f1[s] = F1@overlap of sequence s, valid[s] = 0 for sequences left empty (the
 *   reference skips them); the batch metric is sum(f1) / sum(valid). scratch: n_seq * n_steps bytes.
 * =============================================================================================================== */
int twog_predict_labels(const float* logp, int bs, int n_classes, int T, int E, int downsampling, int T_out,
                        int64_t* labels, void* stream);
int twog_f1_at_k(const int64_t* y_true, const int64_t* y_pred, int n_seq, int n_steps, int num_classes, double overlap,
                 int64_t ignore_value, int use_ignore, unsigned char* scratch, float* f1, float* valid, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TWOG_GCN_H */
