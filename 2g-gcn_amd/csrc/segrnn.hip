// Segment-level gated bidirectional recurrence with message passing between entities.
//
// Reference: the segment loop of TGGCN.forward (vhoi/models.py:785-880): for every (tf, tb) pair each human / object
// gathers attention-weighted messages computed from the *previously committed* segment states of the other entities
// (_humans_to_human_segment_message :1051, _humans_to_object_segment_message :1145,
// _objects_to_human_segment_message :1239, _objects_to_object_segment_message :1334), concatenates them to its
// frame-level input and takes one gated GRUCell step (_bidirectional_step :1535-1564).
//
// MI355X design: entities are batched as rows ((clip, entity)), the forward and the backward chains advance in the
// same launches, and the frame-level part of W_ih x (the bulk of the FLOPs) is hoisted out of the chain into one
// large GEMM by the caller. Per step the chain costs three grouped launches of MFMA GEMMs / fused kernels:
//   (1) sender MLPs on the previous states      relu(H_prev W_s^T + b)          grouped GEMM (humans, objects) x dir
//   (2) attention over senders + weighted sums  twog_attn_fwd                   one workgroup per clip x dir
//   (3) W_ih[:, msg] m + W_hh h_prev            grouped GEMM, then the fused gate kernel (u*GRU + (1-u)*h_prev)
// Everything the backward pass needs (gate activations, sender messages, aggregated messages, attention weights) is
// written once into (clip, time, entity)-ordered buffers so the weight gradients become large GEMMs after the loop.
#include "twog_common.h"
#include "graph_cache.h"

namespace {

inline twog_rows_t rows_be(const float* base, int E, int64_t row_w, int T) {
    twog_rows_t r;
    r.ptr = const_cast<float*>(base); r.inner = E; r.ld_inner = row_w; r.ld_outer = (int64_t)T * E * row_w; r.pad_ = 0;
    return r;
}
inline twog_rows_t rows_plain(const float* base, int64_t ld) {
    twog_rows_t r;
    r.ptr = const_cast<float*>(base); r.inner = 1; r.ld_inner = ld; r.ld_outer = ld; r.pad_ = 0;
    return r;
}
inline twog_rows_t rows_null() {
    twog_rows_t r;
    r.ptr = nullptr; r.inner = 1; r.ld_inner = 0; r.ld_outer = 0; r.pad_ = 0;
    return r;
}
inline void gemm_set(twog_gemm_t& G, twog_rows_t A, twog_rows_t B, twog_rows_t C, const float* bias, int M, int N,
                     int K, int act, int acc) {
    G.A = A; G.B = B; G.C = C; G.bias = bias; G.M = M; G.N = N; G.K = K; G.act = act; G.accumulate = acc; G.batch = 1;
    G.a_batch_stride = G.b_batch_stride = G.c_batch_stride = 0;
    G.a_colsum = nullptr; G.a_colsum_accumulate = 0; G.pad2_ = 0;
}

struct Dims {
    int bs, T, H, O, h, nsh, nso, nmh, nmo, natt;
};
inline Dims dims_of(const twog_segrnn_t& S) {
    Dims d;
    d.bs = S.bs; d.T = S.T; d.H = S.H; d.O = S.O; d.h = S.hidden;
    d.nsh = (S.rel_hh ? 1 : 0) + (S.rel_ho ? 1 : 0);  // sender MLPs applied to human states  (hh | ho)
    d.nso = (S.rel_oh ? 1 : 0) + (S.rel_oo ? 1 : 0);  // to object states                     (oh | oo)
    d.nmh = (S.rel_hh ? 1 : 0) + (S.rel_oh ? 1 : 0);  // message blocks received by a human   (hh | oh)
    d.nmo = (S.rel_ho ? 1 : 0) + (S.rel_oo ? 1 : 0);  // by an object                         (ho | oo)
    d.natt = S.H * S.H + 2 * S.H * S.O + S.O * S.O;
    return d;
}

// attention descriptor of one direction at time t (previous states at tp, or zeros at the chain start)
inline void fill_attn(twog_attn_t& A, const twog_segrnn_t& S, const Dims& d, int dir, int t, bool first, int tp) {
    const int h = d.h;
    A.feat_h = first ? rows_plain(S.zeros, h) : rows_be(S.hs_h + (int64_t)tp * d.H * 2 * h + dir * h, d.H, 2 * h, d.T);
    A.feat_o = first ? rows_plain(S.zeros, h) : rows_be(S.hs_o + (int64_t)tp * d.O * 2 * h + dir * h, d.O, 2 * h, d.T);
    const float* msh = S.msrc_h + ((int64_t)dir * d.bs * d.T * d.H + (int64_t)t * d.H) * d.nsh * h;
    const float* mso = S.msrc_o + ((int64_t)dir * d.bs * d.T * d.O + (int64_t)t * d.O) * d.nso * h;
    float* mgh = S.mg_h + ((int64_t)dir * d.bs * d.T * d.H + (int64_t)t * d.H) * d.nmh * h;
    float* mgo = S.mg_o + ((int64_t)dir * d.bs * d.T * d.O + (int64_t)t * d.O) * d.nmo * h;
    const twog_rows_t nul = rows_null();
    A.msg_hh = S.rel_hh ? rows_be(msh, d.H, d.nsh * h, d.T) : nul;
    A.msg_ho = S.rel_ho ? rows_be(msh + (S.rel_hh ? h : 0), d.H, d.nsh * h, d.T) : nul;
    A.msg_oh = S.rel_oh ? rows_be(mso, d.O, d.nso * h, d.T) : nul;
    A.msg_oo = S.rel_oo ? rows_be(mso + (S.rel_oh ? h : 0), d.O, d.nso * h, d.T) : nul;
    A.msg_so = nul; A.msg_sh = nul;
    A.out_hh = S.rel_hh ? rows_be(mgh, d.H, d.nmh * h, d.T) : nul;
    A.out_oh = S.rel_oh ? rows_be(mgh + (S.rel_hh ? h : 0), d.H, d.nmh * h, d.T) : nul;
    A.out_ho = S.rel_ho ? rows_be(mgo, d.O, d.nmo * h, d.T) : nul;
    A.out_oo = S.rel_oo ? rows_be(mgo + (S.rel_ho ? h : 0), d.O, d.nmo * h, d.T) : nul;
    A.out_sh = nul; A.out_so = nul;
    A.obj_mask = S.obj_mask;
    A.att = S.att + ((int64_t)dir * d.T + t) * d.bs * d.natt;
    A.n_inst = d.bs; A.inst_per_clip = 1; A.H = d.H; A.O = d.O; A.D = h; A.hidden = h;
    A.scale = S.att_scale;
    A.recv_mask_ho = 0;  // the segment-level human->object message is NOT masked by the receiver (models.py:841-843)
}

}  // namespace

static int segrnn_fwd_impl(const twog_segrnn_t* desc, void* chain_ws, size_t chain_ws_bytes, void* stream) {
    const twog_segrnn_t& S = *desc;
    const Dims d = dims_of(S);
    const int h = d.h, T = d.T;
    const bool msg = S.msg_segment && (d.nmh + d.nmo) > 0;
    for (int s = 0; s < T; ++s) {
        const bool first = (s == 0);
        twog_gemm_t gm[8];
        int n = 0, rc;
        if (msg) {
            // (1) sender MLPs on the previous states
            for (int dir = 0; dir < 2; ++dir) {
                const int t = dir == 0 ? s : T - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
                if (d.nsh && d.H) {
                    twog_rows_t A = first ? rows_plain(S.zeros, h)
                                          : rows_be(S.hs_h + (int64_t)tp * d.H * 2 * h + dir * h, d.H, 2 * h, T);
                    float* C = S.msrc_h + ((int64_t)dir * d.bs * T * d.H + (int64_t)t * d.H) * d.nsh * h;
                    gemm_set(gm[n++], A, rows_plain(S.w_smsg_h, h), rows_be(C, d.H, d.nsh * h, T), S.b_smsg_h,
                             d.bs * d.H, d.nsh * h, h, 1, 0);
                }
                if (d.nso && d.O) {
                    twog_rows_t A = first ? rows_plain(S.zeros, h)
                                          : rows_be(S.hs_o + (int64_t)tp * d.O * 2 * h + dir * h, d.O, 2 * h, T);
                    float* C = S.msrc_o + ((int64_t)dir * d.bs * T * d.O + (int64_t)t * d.O) * d.nso * h;
                    gemm_set(gm[n++], A, rows_plain(S.w_smsg_o, h), rows_be(C, d.O, d.nso * h, T), S.b_smsg_o,
                             d.bs * d.O, d.nso * h, h, 1, 0);
                }
            }
            rc = twog_gemm_f32_chain(gm, n, 0, 0, chain_ws, chain_ws_bytes, stream);
            if (rc) return rc;
            // (2) attention + weighted sums
            twog_attn_t at[2];
            for (int dir = 0; dir < 2; ++dir) {
                const int t = dir == 0 ? s : T - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
                fill_attn(at[dir], S, d, dir, t, first, tp);
            }
            rc = twog_attn_fwd(at, 2, stream);
            if (rc) return rc;
        }
        // (3) projections of messages and previous state, then the gates
        n = 0;
        twog_gru_step_t st[4];
        twog_gemm_t ghp[4], gimp[4];
        int ns = 0;
        for (int dir = 0; dir < 2; ++dir) {
            const int t = dir == 0 ? s : T - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
            for (int kind = 0; kind < 2; ++kind) {  // 0 humans, 1 objects
                const int E = kind == 0 ? d.H : d.O;
                if (E == 0) continue;
                const int rows = d.bs * E, nm = kind == 0 ? d.nmh : d.nmo;
                float* hs = kind == 0 ? S.hs_h : S.hs_o;
                twog_rows_t hprev = first ? rows_plain(S.zeros, h)
                                          : rows_be(hs + (int64_t)tp * E * 2 * h + dir * h, E, 2 * h, T);
                float* tmp_gh = (kind == 0 ? S.tmp_gh_h : S.tmp_gh_o) + (int64_t)dir * rows * 3 * h;
                float* tmp_gim = (kind == 0 ? S.tmp_gim_h : S.tmp_gim_o) + (int64_t)dir * rows * 3 * h;
                const float* whh = kind == 0 ? S.w_hh_h[dir] : S.w_hh_o[dir];
                const float* bhh = kind == 0 ? S.b_hh_h[dir] : S.b_hh_o[dir];
                gemm_set(gm[n], hprev, rows_plain(whh, h), rows_plain(tmp_gh, 3 * h), bhh, rows, 3 * h, h, 0, 0);
                ghp[ns] = gm[n++];
                gimp[ns].M = 0;
                if (msg && nm) {
                    const float* mg = (kind == 0 ? S.mg_h : S.mg_o) + ((int64_t)dir * d.bs * T * E + (int64_t)t * E) * nm * h;
                    const float* wim = kind == 0 ? S.w_ihm_h[dir] : S.w_ihm_o[dir];
                    const int64_t ldw = kind == 0 ? S.ld_ih_h : S.ld_ih_o;
                    gemm_set(gm[n], rows_be(mg, E, nm * h, T), rows_plain(wim, ldw), rows_plain(tmp_gim, 3 * h),
                             nullptr, rows, 3 * h, nm * h, 0, 0);
                    gimp[ns] = gm[n++];
                }
                twog_gru_step_t& G = st[ns++];
                const float* gi = kind == 0 ? S.gi_h : S.gi_o;
                G.gi = rows_be(gi + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T);
                G.gi2 = (msg && nm) ? rows_plain(tmp_gim, 3 * h) : rows_null();
                G.gh = rows_plain(tmp_gh, 3 * h);
                G.h_prev = first ? rows_null() : hprev;
                G.h_out = rows_be(hs + (int64_t)t * E * 2 * h + dir * h, E, 2 * h, T);
                float* sv = kind == 0 ? S.save_h : S.save_o;
                G.save = rows_be(sv + ((int64_t)dir * d.bs * T * E + (int64_t)t * E) * 4 * h, E, 4 * h, T);
                G.u = (kind == 0 ? S.u_h : S.u_o) + (int64_t)t * E;
                G.u_ld_outer = (int64_t)T * E; G.u_ld_inner = 1; G.u_inner = E;
                G.rows = rows; G.hidden = h;
            }
        }
        // both products and the gates in one launch where the shapes allow it (gemm_gru_fwd_kernel, gemm_f32.hip)
        rc = twog_internal_gemm_gru_fwd(ghp, gimp, st, ns, 0, stream);
        if (rc < 0) return rc;
        if (rc == 0) continue;
        rc = twog_gemm_f32_chain(gm, n, 0, 0, chain_ws, chain_ws_bytes, stream);
        if (rc) return rc;
        rc = twog_gru_step_fwd(st, ns, stream);
        if (rc) return rc;
    }
    return 0;
}

// adds the parked d_u partials (fused gate epilogues) into d_u in fixed order: one thread per (clip, time, entity)
__global__ __launch_bounds__(256) void du_reduce_kernel(const float* part, float* du, int bs, int T, int E, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * T * E) return;
    const int e = i % E, t = (i / E) % T, b = i / (E * T);
    const int rows = bs * E, row = b * E + e;
    float acc = 0.f;
    for (int dir = 0; dir < 2; ++dir) {
        // the first step of each chain (t = T-1 forward in time, t = 0 backward) ran as a separate launch
        if ((dir == 0 && t == T - 1) || (dir == 1 && t == 0)) continue;
        const float* p = part + ((int64_t)(dir * T + t) * 16) * rows + row;
        for (int k = 0; k < P; ++k) acc += p[(int64_t)k * rows];
    }
    du[i] += acc;
}

static int segrnn_bwd_impl(const twog_segrnn_t* desc, const twog_segrnn_bwd_t* bdesc, void* chain_ws, size_t chain_ws_bytes,
                           void* stream) {
    const twog_segrnn_t& S = *desc;
    const twog_segrnn_bwd_t& B = *bdesc;
    const Dims d = dims_of(S);
    const int h = d.h, T = d.T;
    const bool msg = S.msg_segment && (d.nmh + d.nmo) > 0;
    // gate backward descriptors of chain step s, order (dir, kind) like the GEMM problem lists below
    auto make_gates = [&](int s, twog_gru_step_bwd_t* st, float** du_part) {
        const bool first = (s == 0), last = (s == T - 1);
        int ns = 0;
        for (int dir = 0; dir < 2; ++dir) {
            const int t = dir == 0 ? s : T - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
            for (int kind = 0; kind < 2; ++kind) {
                const int E = kind == 0 ? d.H : d.O;
                if (E == 0) continue;
                const int rows = d.bs * E;
                const float* hs = kind == 0 ? S.hs_h : S.hs_o;
                const float* dhs = kind == 0 ? B.d_hs_h : B.d_hs_o;
                float* carry = (kind == 0 ? B.carry_h : B.carry_o) + (int64_t)dir * rows * h;
                twog_gru_step_bwd_t& G = st[ns];
                G.dh = rows_be(dhs + (int64_t)t * E * 2 * h + dir * h, E, 2 * h, T);
                G.dh2 = last ? rows_null() : rows_plain(carry, h);
                const float* sv = kind == 0 ? S.save_h : S.save_o;
                G.save = rows_be(sv + ((int64_t)dir * d.bs * T * E + (int64_t)t * E) * 4 * h, E, 4 * h, T);
                G.h_prev = first ? rows_null() : rows_be(hs + (int64_t)tp * E * 2 * h + dir * h, E, 2 * h, T);
                float* dgi = kind == 0 ? B.d_gi_h : B.d_gi_o;
                float* dgh = kind == 0 ? B.d_gh_h : B.d_gh_o;
                G.dgi = rows_be(dgi + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T);
                G.dgh = rows_be(dgh + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T);
                G.dh_prev = rows_plain(carry, h);
                G.u = (kind == 0 ? S.u_h : S.u_o) + (int64_t)t * E;
                G.du = (kind == 0 ? B.d_u_h : B.d_u_o) + (int64_t)t * E;
                G.u_ld_outer = (int64_t)T * E; G.u_ld_inner = 1; G.u_inner = E;
                G.rows = rows; G.hidden = h; G.dh_prev_accumulate = 0;
                if (du_part) {
                    float* base = kind == 0 ? B.du_part_h : B.du_part_o;
                    du_part[ns] = base ? base + ((int64_t)(dir * T + t) * 16) * rows : nullptr;
                }
                ++ns;
            }
        }
        return ns;
    };
    // Which launch adds the LAST contribution to the carried state gradients of a step? With messages from both kinds
    // of senders it is the sender-MLP GEMM (e); without segment-level messages the W_hh GEMM (b). In those two layouts
    // the gate backward of the next chain step is fused into that launch's epilogue (gemm_f32.hip, GateArgs); any other
    // layout (a sender kind missing) keeps the separate gate launches.
    const bool both = d.H > 0 && d.O > 0;
    const bool fuse_e = msg && both && d.nsh > 0 && d.nso > 0;
    const bool fuse_b = !msg && both;
    bool fuse = (fuse_e || fuse_b) && B.du_part_h && B.du_part_o && T > 1 && (h + 63) / 64 <= 8 &&
                getenv("TWOG_NO_GATE_FUSION") == nullptr;
    for (int s = T - 1; s >= 0; --s) {
        const bool first = (s == 0);
        int rc;
        // (a) gate backward (separate launch: always for the first step of the chain, else only when not fused)
        twog_gru_step_bwd_t st[4];
        if (!fuse || s == T - 1) {
            const int ns = make_gates(s, st, nullptr);
            rc = twog_gru_step_bwd(st, ns, stream);
            if (rc) return rc;
        }
        twog_gru_step_bwd_t nxt[4];  // gates of step s-1, aligned with the problems of the fused launch
        float* nxt_part[4];
        if (fuse && !first) make_gates(s - 1, nxt, nxt_part);
        // (b) through the projections: d_mg = d_gi W_ih[:, msg] ; carry += d_gh W_hh
        twog_gemm_t gm[8];
        int n = 0;
        for (int dir = 0; dir < 2; ++dir) {
            const int t = dir == 0 ? s : T - 1 - s;
            for (int kind = 0; kind < 2; ++kind) {
                const int E = kind == 0 ? d.H : d.O;
                if (E == 0) continue;
                const int rows = d.bs * E, nm = kind == 0 ? d.nmh : d.nmo;
                float* dgi = kind == 0 ? B.d_gi_h : B.d_gi_o;
                float* dgh = kind == 0 ? B.d_gh_h : B.d_gh_o;
                float* carry = (kind == 0 ? B.carry_h : B.carry_o) + (int64_t)dir * rows * h;
                if (!first) {
                    const float* whh = kind == 0 ? S.w_hh_h[dir] : S.w_hh_o[dir];
                    gemm_set(gm[n++], rows_be(dgh + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T),
                             rows_plain(whh, h), rows_plain(carry, h), nullptr, rows, h, 3 * h, 0, 1);
                }
                if (msg && nm) {
                    const float* wim = kind == 0 ? S.w_ihm_h[dir] : S.w_ihm_o[dir];
                    const int64_t ldw = kind == 0 ? S.ld_ih_h : S.ld_ih_o;
                    float* dmg = (kind == 0 ? B.tmp_dmg_h : B.tmp_dmg_o) + (int64_t)dir * rows * nm * h;
                    gemm_set(gm[n++], rows_be(dgi + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T),
                             rows_plain(wim, ldw), rows_plain(dmg, nm * h), nullptr, rows, nm * h, 3 * h, 0, 0);
                }
            }
        }
        if (n) {
            rc = 1;
            if (fuse && fuse_b && !first) {  // problems are exactly the four W_hh GEMMs, in (dir, kind) order
                rc = twog_internal_gemm_gate_bwd(gm, n, nxt, nxt_part, 0, chain_ws, chain_ws_bytes, stream);
                if (rc < 0) return rc;
                if (rc == 1) {  // shape not served by the fused kernel: decided at the first step, cannot change later
                    if (s != T - 1) return -120;
                    fuse = false;
                }
            }
            if (rc == 1) rc = twog_gemm_f32_chain(gm, n, 0, 1, chain_ws, chain_ws_bytes, stream);
            if (rc) return rc;
        }
        if (!msg) continue;
        // (c) attention backward: d_mg -> gradient wrt sender messages (ReLU-masked) and wrt the previous states
        twog_attn_bwd_t ab[2];
        for (int dir = 0; dir < 2; ++dir) {
            const int t = dir == 0 ? s : T - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
            twog_attn_bwd_t& X = ab[dir];
            fill_attn(X.f, S, d, dir, t, first, tp);
            const twog_rows_t nul = rows_null();
            float* dmgh = B.tmp_dmg_h + (int64_t)dir * d.bs * d.H * d.nmh * h;
            float* dmgo = B.tmp_dmg_o + (int64_t)dir * d.bs * d.O * d.nmo * h;
            X.dout_hh = S.rel_hh ? rows_plain(dmgh, d.nmh * h) : nul;
            X.dout_oh = S.rel_oh ? rows_plain(dmgh + (S.rel_hh ? h : 0), d.nmh * h) : nul;
            X.dout_ho = S.rel_ho ? rows_plain(dmgo, d.nmo * h) : nul;
            X.dout_oo = S.rel_oo ? rows_plain(dmgo + (S.rel_ho ? h : 0), d.nmo * h) : nul;
            X.dout_sh = nul; X.dout_so = nul;
            float* dph = B.d_pre_h + ((int64_t)dir * d.bs * T * d.H + (int64_t)t * d.H) * d.nsh * h;
            float* dpo = B.d_pre_o + ((int64_t)dir * d.bs * T * d.O + (int64_t)t * d.O) * d.nso * h;
            X.dmsg_hh = S.rel_hh ? rows_be(dph, d.H, d.nsh * h, T) : nul;
            X.dmsg_ho = S.rel_ho ? rows_be(dph + (S.rel_hh ? h : 0), d.H, d.nsh * h, T) : nul;
            X.dmsg_oh = S.rel_oh ? rows_be(dpo, d.O, d.nso * h, T) : nul;
            X.dmsg_oo = S.rel_oo ? rows_be(dpo + (S.rel_oh ? h : 0), d.O, d.nso * h, T) : nul;
            X.dmsg_so = nul; X.dmsg_sh = nul;
            // at the chain start the features are the constant zero state: its gradient is dropped
            X.dfeat_h = rows_plain((first ? B.trash : B.carry_h + (int64_t)dir * d.bs * d.H * h), h);
            X.dfeat_o = rows_plain((first ? B.trash : B.carry_o + (int64_t)dir * d.bs * d.O * h), h);
            X.dw_extra = nullptr;
            X.dfeat_accumulate = first ? 0 : 1;
            X.relu_mask_dmsg = 1;
        }
        rc = twog_attn_bwd(ab, 2, stream);
        if (rc) return rc;
        if (first) continue;
        // (e) through the sender MLPs: carry += d_pre W_s
        n = 0;
        for (int dir = 0; dir < 2; ++dir) {
            const int t = dir == 0 ? s : T - 1 - s;
            if (d.nsh && d.H) {
                float* dph = B.d_pre_h + ((int64_t)dir * d.bs * T * d.H + (int64_t)t * d.H) * d.nsh * h;
                gemm_set(gm[n++], rows_be(dph, d.H, d.nsh * h, T), rows_plain(S.w_smsg_h, h),
                         rows_plain(B.carry_h + (int64_t)dir * d.bs * d.H * h, h), nullptr, d.bs * d.H, h, d.nsh * h, 0, 1);
            }
            if (d.nso && d.O) {
                float* dpo = B.d_pre_o + ((int64_t)dir * d.bs * T * d.O + (int64_t)t * d.O) * d.nso * h;
                gemm_set(gm[n++], rows_be(dpo, d.O, d.nso * h, T), rows_plain(S.w_smsg_o, h),
                         rows_plain(B.carry_o + (int64_t)dir * d.bs * d.O * h, h), nullptr, d.bs * d.O, h, d.nso * h, 0, 1);
            }
        }
        rc = 1;
        if (fuse && fuse_e) {  // problems are exactly (dir, kind) = the order of make_gates
            rc = twog_internal_gemm_gate_bwd(gm, n, nxt, nxt_part, 0, chain_ws, chain_ws_bytes, stream);
            if (rc < 0) return rc;
            if (rc == 1) {
                if (s != T - 1) return -122;
                fuse = false;
            }
        }
        if (rc == 1) rc = twog_gemm_f32_chain(gm, n, 0, 1, chain_ws, chain_ws_bytes, stream);
        if (rc) return rc;
    }
    if (fuse) {
        const int P = 2 * ((h + 63) / 64);
        if (d.H) hipLaunchKernelGGL(du_reduce_kernel, dim3((d.bs * T * d.H + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                    B.du_part_h, B.d_u_h, d.bs, T, d.H, P);
        if (d.O) hipLaunchKernelGGL(du_reduce_kernel, dim3((d.bs * T * d.O + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                    B.du_part_o, B.d_u_o, d.bs, T, d.O, P);
        TWOG_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int twog_segrnn_fwd(const twog_segrnn_t* desc, void* chain_ws, size_t chain_ws_bytes, void* stream) {
    const int tag[2] = {0x33, twog_internal_gru_fwd_mode()};
    twog_graph::Desc key;
    key.pod(tag).pod(*desc).pod(chain_ws).pod(chain_ws_bytes);
    return twog_graph::run(key, (hipStream_t)stream,
                           [&](hipStream_t st) { return segrnn_fwd_impl(desc, chain_ws, chain_ws_bytes, st); });
}

extern "C" int twog_segrnn_bwd(const twog_segrnn_t* desc, const twog_segrnn_bwd_t* bdesc, void* chain_ws,
                               size_t chain_ws_bytes, void* stream) {
    const int tag = 0x44;
    twog_graph::Desc key;
    key.pod(tag).pod(*desc).pod(*bdesc).pod(chain_ws).pod(chain_ws_bytes);
    return twog_graph::run(key, (hipStream_t)stream,
                           [&](hipStream_t st) { return segrnn_bwd_impl(desc, bdesc, chain_ws, chain_ws_bytes, st); });
}

extern "C" int twog_graph_cache_stats(int64_t* entries, int64_t* collisions) {
    twog_graph::State& S = twog_graph::state();
    std::lock_guard<std::mutex> lock(S.mu);
    if (entries) *entries = (int64_t)S.cache.size();
    if (collisions) *collisions = (int64_t)S.collisions;
    return 0;
}
