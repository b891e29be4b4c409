// General message passing of ONE relation (a set of receivers, a set of senders) -- every message / aggregation form of
// the reference that the tuned four-relations-in-one kernel (attn.hip: sender-only messages, dot-product attention) does
// not cover:
//   messages     sender-only  msg[s]                      (compute_non_relational_message 'generic', models.py:1693-1718)
//                pairwise     relu(P_r[r] + P_s[s])        ('specific' granularity, and the pairwise relation g of
//                                                           compute_relational_message, :1667-1690: Linear(cat[recv,
//                                                           sender]) splits into a receiver and a sender projection)
//   weights      sum          w = sender mask              (relational: sum of the masked pairwise relations)
//                dot          softmax(scale <q_r, k_s> (+b, relu))   ('v2'/'v3'; 'general': keys = A k, :1746)
//                additive     softmax(relu(a_r + c_s))     ('concat': Linear(cat[q, k]) -> 1 splits the same way, :1739-1741)
//                distance     softmax(1 / (d + 1e-7)), senders at distance 0 excluded  (:1757-1775)
//                mean         1 / max(#valid senders, 1)   (mean pooling, e.g. :1034-1037)
// masked softmax with -inf on invalid senders and NaN -> 0 when none is valid (:1750-1753). One workgroup per instance
// ((clip, frame) or clip); entity counts are tiny (<= 16 x 16 pairs), so everything but the feature rows lives in LDS.
// None of these forms is enabled by a shipped configuration: the kernels are written for correctness and coalesced
// access, not tuned like attn.hip.
#include "twog_common.h"

namespace {

constexpr int MAXE = 16;

struct RowSet {
    float* base;
    int64_t step;
    __device__ __forceinline__ float* row(int e) const { return base + e * step; }
    __device__ __forceinline__ bool on() const { return base != nullptr; }
};
__device__ __forceinline__ RowSet rowset(const twog_rows_t& m, int inst, int n) {
    RowSet r;
    if (!m.ptr) { r.base = nullptr; r.step = 0; return r; }
    if (m.inner <= 1) { r.base = m.ptr + (int64_t)inst * n * m.ld_outer; r.step = m.ld_outer; }
    else { r.base = m.ptr + (int64_t)inst * m.ld_outer; r.step = m.ld_inner; }
    return r;
}

__device__ __forceinline__ float wdot(const float* a, const float* b, int n, int lane) {
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) acc = fmaf(a[j], b[j], acc);
    return wave_sum(acc);
}

// validity and raw score of every (receiver, sender) pair -> sV [R*S] (0/1), sS [R*S]
__device__ __forceinline__ void pair_scores(const twog_relation_t& A, int inst, const RowSet& q, const RowSet& k, float* sV,
                                            float* sS, float* sRaw) {
    const int R = A.R, S = A.S;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int clip = inst / A.inst_per_clip;
    for (int p = wv; p < R * S; p += nw) {
        const int r = p / S, s = p - r * S;
        float valid = 1.f, mval = 1.f;
        if (A.send_mask) { mval = A.send_mask[(int64_t)clip * S + s]; if (mval == 0.f) valid = 0.f; }
        if (A.exclude_self && r == s) valid = 0.f;
        float sc = 0.f, raw = 0.f;
        if (A.score_mode == TWOG_REL_DOT) {
            raw = wdot(q.row(r), k.row(s), A.D, lane) * A.scale + (A.score_bias ? A.score_bias[0] : 0.f);
            sc = A.relu_scores ? fmaxf(raw, 0.f) : raw;
        } else if (A.score_mode == TWOG_REL_ADDITIVE) {
            raw = A.a_r[(int64_t)inst * R + r] + A.c_s[(int64_t)inst * S + s];
            sc = fmaxf(raw, 0.f);
        } else if (A.score_mode == TWOG_REL_DISTANCE) {
            const float d = A.dist[(int64_t)inst * A.dist_ld_inst + (int64_t)r * A.dist_ld_r + (int64_t)s * A.dist_ld_s];
            if (d == 0.f) valid = 0.f;
            sc = 1.0f / (d + 1e-7f);
        } else if (A.score_mode == TWOG_REL_SUM) {
            sc = mval;   // weight = the mask value itself
        }
        if (lane == 0) { sV[p] = valid; sS[p] = sc; if (sRaw) sRaw[p] = raw; }
    }
    __syncthreads();
}

// sW [R*S] from validity + scores
__device__ __forceinline__ void pair_weights(const twog_relation_t& A, const float* sV, const float* sS, float* sW) {
    const int R = A.R, S = A.S, r = threadIdx.x;
    if (r < R) {
        if (A.score_mode == TWOG_REL_SUM) {
            for (int s = 0; s < S; ++s) sW[r * S + s] = sV[r * S + s] != 0.f ? sS[r * S + s] : 0.f;
        } else if (A.score_mode == TWOG_REL_MEAN) {
            float cnt = 0.f;
            for (int s = 0; s < S; ++s) cnt += sV[r * S + s];
            cnt = fmaxf(cnt, 1.f);
            for (int s = 0; s < S; ++s) sW[r * S + s] = sV[r * S + s] / cnt;
        } else {
            float m = -INFINITY;
            for (int s = 0; s < S; ++s)
                if (sV[r * S + s] != 0.f) m = fmaxf(m, sS[r * S + s]);
            float sum = 0.f;
            for (int s = 0; s < S; ++s) {
                const float e = sV[r * S + s] != 0.f ? expf(sS[r * S + s] - m) : 0.f;
                sW[r * S + s] = e;
                sum += e;
            }
            for (int s = 0; s < S; ++s) sW[r * S + s] = sV[r * S + s] != 0.f ? sW[r * S + s] / sum : 0.f;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float recv_scale(const twog_relation_t& A, int inst, int r) {
    if (!A.recv_mask) return 1.f;
    return A.recv_mask[(int64_t)(inst / A.inst_per_clip) * A.R + r];
}

__device__ __forceinline__ void relation_fwd_body(const twog_relation_t& A, float* sV, float* sS, float* sW) {
    const int inst = blockIdx.x;
    const int R = A.R, S = A.S, hid = A.hidden;
    const RowSet q = rowset(A.q, inst, R), k = rowset(A.k, inst, S);
    const RowSet msg = rowset(A.msg, inst, S), pr = rowset(A.p_r, inst, R), ps = rowset(A.p_s, inst, S);
    const RowSet out = rowset(A.out, inst, R);
    pair_scores(A, inst, q, k, sV, sS, nullptr);
    pair_weights(A, sV, sS, sW);
    if (A.att)
        for (int i = threadIdx.x; i < R * S; i += blockDim.x) A.att[(int64_t)inst * R * S + i] = sW[i];
    for (int idx = threadIdx.x; idx < R * hid; idx += blockDim.x) {
        const int r = idx / hid, j = idx - r * hid;
        float acc = 0.f;
        if (A.msg_mode == TWOG_REL_MSG_SENDER) {
            for (int s = 0; s < S; ++s) acc = fmaf(sW[r * S + s], msg.row(s)[j], acc);
        } else {
            const float a = pr.row(r)[j];
            for (int s = 0; s < S; ++s) acc = fmaf(sW[r * S + s], fmaxf(a + ps.row(s)[j], 0.f), acc);
        }
        out.row(r)[j] = acc * recv_scale(A, inst, r);
    }
}

__global__ __launch_bounds__(256) void relation_fwd_kernel(const twog_relation_t A) {
    __shared__ float sV[MAXE * MAXE], sS[MAXE * MAXE], sW[MAXE * MAXE];
    relation_fwd_body(A, sV, sS, sW);
}

// Several relations in one launch (blockIdx.y = descriptor): the host-composed segment loop issues every relation of
// both directions of a chain step together. Descriptors travel as kernel arguments (8 x 320 B / 6 x 552 B < 4 KB).
constexpr int MAXREL_F = 8, MAXREL_B = 6;
struct RelFwdBatch { twog_relation_t d[MAXREL_F]; };
__global__ __launch_bounds__(256) void relation_fwd_n_kernel(const RelFwdBatch G) {
    __shared__ float sV[MAXE * MAXE], sS[MAXE * MAXE], sW[MAXE * MAXE];
    const twog_relation_t& A = G.d[blockIdx.y];
    if ((int)blockIdx.x >= A.n_inst || A.R == 0) return;
    relation_fwd_body(A, sV, sS, sW);
}

__device__ __forceinline__ void relation_bwd_body(const twog_relation_bwd_t& B, float* sV, float* sS, float* sW,
                                                  float* sRaw, float* sD) {
    const twog_relation_t& A = B.f;
    const int inst = blockIdx.x;
    const int R = A.R, S = A.S, hid = A.hidden, D = A.D;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const RowSet q = rowset(A.q, inst, R), k = rowset(A.k, inst, S);
    const RowSet msg = rowset(A.msg, inst, S), pr = rowset(A.p_r, inst, R), ps = rowset(A.p_s, inst, S);
    const RowSet dout = rowset(B.dout, inst, R);
    const RowSet dmsg = rowset(B.dmsg, inst, S), dpr = rowset(B.dp_r, inst, R), dps = rowset(B.dp_s, inst, S);
    const RowSet dq = rowset(B.dq, inst, R), dk = rowset(B.dk, inst, S);
    pair_scores(A, inst, q, k, sV, sS, sRaw);
    pair_weights(A, sV, sS, sW);
    const bool scored = A.score_mode == TWOG_REL_DOT || A.score_mode == TWOG_REL_ADDITIVE;
    // ---- gradient wrt the messages
    if (A.msg_mode == TWOG_REL_MSG_SENDER) {
        if (dmsg.on())
            for (int idx = threadIdx.x; idx < S * hid; idx += blockDim.x) {
                const int s = idx / hid, j = idx - s * hid;
                float acc = 0.f;
                for (int r = 0; r < R; ++r) acc = fmaf(sW[r * S + s] * recv_scale(A, inst, r), dout.row(r)[j], acc);
                if (B.relu_mask_dmsg && !(msg.row(s)[j] > 0.f)) acc = 0.f;
                dmsg.row(s)[j] = acc;
            }
    } else {
        for (int idx = threadIdx.x; idx < R * hid; idx += blockDim.x) {
            const int r = idx / hid, j = idx - r * hid;
            const float a = pr.row(r)[j], g = dout.row(r)[j] * recv_scale(A, inst, r);
            float acc = 0.f;
            for (int s = 0; s < S; ++s)
                if (a + ps.row(s)[j] > 0.f) acc = fmaf(sW[r * S + s], g, acc);
            dpr.row(r)[j] = acc;
        }
        for (int idx = threadIdx.x; idx < S * hid; idx += blockDim.x) {
            const int s = idx / hid, j = idx - s * hid;
            const float c = ps.row(s)[j];
            float acc = 0.f;
            for (int r = 0; r < R; ++r)
                if (pr.row(r)[j] + c > 0.f) acc = fmaf(sW[r * S + s] * recv_scale(A, inst, r), dout.row(r)[j], acc);
            dps.row(s)[j] = acc;
        }
    }
    if (!scored) {
        // constant weights: nothing reaches the scores; requested feature gradients that are not accumulated are zero
        if (dq.on() && !B.dq_accumulate)
            for (int idx = threadIdx.x; idx < R * D; idx += blockDim.x) dq.row(idx / D)[idx % D] = 0.f;
        if (dk.on() && !B.dk_accumulate)
            for (int idx = threadIdx.x; idx < S * D; idx += blockDim.x) dk.row(idx / D)[idx % D] = 0.f;
        return;
    }
    // ---- gradient wrt the weights, one wave per pair: dw[r][s] = recv_scale_r <dout[r], M(r, s)>
    for (int p = wv; p < R * S; p += nw) {
        const int r = p / S, s = p - r * S;
        float acc = 0.f;
        if (sW[p] != 0.f) {
            const float* g = dout.row(r);
            if (A.msg_mode == TWOG_REL_MSG_SENDER) {
                const float* m = msg.row(s);
                for (int j = lane; j < hid; j += 64) acc = fmaf(g[j], m[j], acc);
            } else {
                const float* a = pr.row(r);
                const float* c = ps.row(s);
                for (int j = lane; j < hid; j += 64) acc = fmaf(g[j], fmaxf(a[j] + c[j], 0.f), acc);
            }
            acc = wave_sum(acc) * recv_scale(A, inst, r);
        }
        if (lane == 0) sD[p] = acc;
    }
    __syncthreads();
    // softmax backward per receiver, then through the score's own ReLU: sD <- d raw score
    if (threadIdx.x < R) {
        const int r = threadIdx.x;
        float t = 0.f;
        for (int s = 0; s < S; ++s) t = fmaf(sW[r * S + s], sD[r * S + s], t);
        for (int s = 0; s < S; ++s) {
            float d = sW[r * S + s] * (sD[r * S + s] - t);
            const bool relu = A.score_mode == TWOG_REL_ADDITIVE || A.relu_scores;
            if (relu && !(sRaw[r * S + s] > 0.f)) d = 0.f;
            sD[r * S + s] = d;
        }
    }
    __syncthreads();
    if (A.score_mode == TWOG_REL_ADDITIVE) {
        if (threadIdx.x < R) {
            float t = 0.f;
            for (int s = 0; s < S; ++s) t += sD[threadIdx.x * S + s];
            B.da_r[(int64_t)inst * R + threadIdx.x] = t;
        } else if (threadIdx.x >= 64 && threadIdx.x < 64 + S) {
            const int s = threadIdx.x - 64;
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += sD[r * S + s];
            B.dc_s[(int64_t)inst * S + s] = t;
        }
        return;
    }
    if (B.dscore_sum && threadIdx.x == 0) {
        float t = 0.f;
        for (int p = 0; p < R * S; ++p) t += sD[p];
        B.dscore_sum[inst] = t;
    }
    // dot scores: dq[r] (+)= scale sum_s dscore k[s] ; dk[s] (+)= scale sum_r dscore q[r].
    // q and k may be views of the same rows (self relations: humans -> human): the two passes are separated by a barrier
    // and each adds its own term, so an accumulating target sees both.
    if (dq.on())
        for (int idx = threadIdx.x; idx < R * D; idx += blockDim.x) {
            const int r = idx / D, j = idx - r * D;
            float acc = 0.f;
            for (int s = 0; s < S; ++s) acc = fmaf(sD[r * S + s], k.row(s)[j], acc);
            acc *= A.scale;
            float* d = dq.row(r) + j;
            *d = B.dq_accumulate ? *d + acc : acc;
        }
    __syncthreads();
    if (dk.on())
        for (int idx = threadIdx.x; idx < S * D; idx += blockDim.x) {
            const int s = idx / D, j = idx - s * D;
            float acc = 0.f;
            for (int r = 0; r < R; ++r) acc = fmaf(sD[r * S + s], q.row(r)[j], acc);
            acc *= A.scale;
            float* d = dk.row(s) + j;
            *d = B.dk_accumulate ? *d + acc : acc;
        }
}

__global__ __launch_bounds__(256) void relation_bwd_kernel(const twog_relation_bwd_t B) {
    __shared__ float sV[MAXE * MAXE], sS[MAXE * MAXE], sW[MAXE * MAXE], sRaw[MAXE * MAXE], sD[MAXE * MAXE];
    relation_bwd_body(B, sV, sS, sW, sRaw, sD);
}

struct RelBwdBatch { twog_relation_bwd_t d[MAXREL_B]; };
__global__ __launch_bounds__(256) void relation_bwd_n_kernel(const RelBwdBatch G) {
    __shared__ float sV[MAXE * MAXE], sS[MAXE * MAXE], sW[MAXE * MAXE], sRaw[MAXE * MAXE], sD[MAXE * MAXE];
    const twog_relation_bwd_t& B = G.d[blockIdx.y];
    if ((int)blockIdx.x >= B.f.n_inst || B.f.R == 0) return;
    relation_bwd_body(B, sV, sS, sW, sRaw, sD);
}

inline bool rows_ok(const twog_rows_t& m, int n) { return !m.ptr || m.inner <= 1 || m.inner == n; }
inline int check(const twog_relation_t& a) {
    if (a.R < 0 || a.S < 0 || a.R > MAXE || a.S > MAXE || a.hidden <= 0 || a.inst_per_clip <= 0) return -2;
    if (a.score_mode < TWOG_REL_SUM || a.score_mode > TWOG_REL_MEAN) return -2;
    if (a.msg_mode != TWOG_REL_MSG_SENDER && a.msg_mode != TWOG_REL_MSG_PAIR) return -2;
    if (a.score_mode == TWOG_REL_DOT && (!a.q.ptr || !a.k.ptr || a.D <= 0)) return -2;
    if (a.score_mode == TWOG_REL_ADDITIVE && (!a.a_r || !a.c_s)) return -2;
    if (a.score_mode == TWOG_REL_DISTANCE && !a.dist) return -2;
    if (a.msg_mode == TWOG_REL_MSG_SENDER ? !a.msg.ptr : (!a.p_r.ptr || !a.p_s.ptr)) return -2;
    if (!a.out.ptr) return -2;
    if (!rows_ok(a.q, a.R) || !rows_ok(a.k, a.S) || !rows_ok(a.msg, a.S) || !rows_ok(a.p_r, a.R) || !rows_ok(a.p_s, a.S) ||
        !rows_ok(a.out, a.R))
        return -2;
    return 0;
}

}  // namespace

extern "C" int twog_relation_limits(void) { return MAXE; }

extern "C" int twog_relation_fwd(const twog_relation_t* a, void* stream) {
    if (int rc = check(*a)) return rc;
    if (a->n_inst <= 0 || a->R == 0) return 0;
    hipLaunchKernelGGL(relation_fwd_kernel, dim3(a->n_inst), dim3(256), 0, (hipStream_t)stream, *a);
    TWOG_CHECK_LAUNCH();
    return 0;
}

static int check_bwd(const twog_relation_bwd_t* b) {
    const twog_relation_t& a = b->f;
    if (int rc = check(a)) return rc;
    if (!b->dout.ptr || !rows_ok(b->dout, a.R) || !rows_ok(b->dmsg, a.S) || !rows_ok(b->dp_r, a.R) ||
        !rows_ok(b->dp_s, a.S) || !rows_ok(b->dq, a.R) || !rows_ok(b->dk, a.S))
        return -2;
    if (a.msg_mode == TWOG_REL_MSG_PAIR && (!b->dp_r.ptr || !b->dp_s.ptr)) return -2;
    if (a.score_mode == TWOG_REL_ADDITIVE && (!b->da_r || !b->dc_s)) return -2;
    return 0;
}

extern "C" int twog_relation_bwd(const twog_relation_bwd_t* b, void* stream) {
    const twog_relation_t& a = b->f;
    if (int rc = check_bwd(b)) return rc;
    if (a.n_inst <= 0 || a.R == 0) return 0;
    hipLaunchKernelGGL(relation_bwd_kernel, dim3(a.n_inst), dim3(256), 0, (hipStream_t)stream, *b);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// n relations per call, MAXREL_F / MAXREL_B per launch. The descriptors of one call must not ACCUMULATE into the same
// rows (dq / dk with *_accumulate): the launches of a chunk run concurrently.
extern "C" int twog_relation_fwd_n(const twog_relation_t* rels, int n, void* stream) {
    for (int i = 0; i < n; ++i)
        if (int rc = check(rels[i])) return rc;
    for (int done = 0; done < n; done += MAXREL_F) {
        RelFwdBatch G;
        const int m = n - done < MAXREL_F ? n - done : MAXREL_F;
        int max_inst = 0;
        for (int i = 0; i < m; ++i) {
            G.d[i] = rels[done + i];
            if (G.d[i].R > 0 && G.d[i].n_inst > max_inst) max_inst = G.d[i].n_inst;
        }
        if (max_inst <= 0) continue;
        hipLaunchKernelGGL(relation_fwd_n_kernel, dim3(max_inst, m), dim3(256), 0, (hipStream_t)stream, G);
        TWOG_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int twog_relation_bwd_n(const twog_relation_bwd_t* rels, int n, void* stream) {
    for (int i = 0; i < n; ++i)
        if (int rc = check_bwd(rels + i)) return rc;
    for (int done = 0; done < n; done += MAXREL_B) {
        RelBwdBatch G;
        const int m = n - done < MAXREL_B ? n - done : MAXREL_B;
        int max_inst = 0;
        for (int i = 0; i < m; ++i) {
            G.d[i] = rels[done + i];
            if (G.d[i].f.R > 0 && G.d[i].f.n_inst > max_inst) max_inst = G.d[i].f.n_inst;
        }
        if (max_inst <= 0) continue;
        hipLaunchKernelGGL(relation_bwd_n_kernel, dim3(max_inst, m), dim3(256), 0, (hipStream_t)stream, G);
        TWOG_CHECK_LAUNCH();
    }
    return 0;
}
