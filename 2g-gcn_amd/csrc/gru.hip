// GRU gate math + the frame-level bidirectional GRU recurrence.
//
// Reference: torch.nn.GRU (bidirectional, 1 layer) applied per entity slice in a Python loop
// (vhoi/models.py:983-1002, modules :267-269, :274-276, :299-301) and the gated GRUCell step of the segment level,
// h_t = u * GRUCell(x_t, h_{t-1}) + (1 - u) * h_{t-1} (vhoi/models.py:1535-1564). Gate order r, z, n.
//
// MI355X design: the input projection W_ih x + b_ih of ALL time steps is one large MFMA GEMM done by the caller; only
// the truly sequential part runs here: per step one grouped MFMA GEMM gh = h_{t-1} W_hh^T + b_hh for every
// (entity type, direction) at once -- all entities of a type are batched as rows, both directions advance in the same
// launch -- followed by one fused gate kernel (one workgroup per row, lane-contiguous along the hidden dimension).
// r, z, n and W_hn h are saved for the backward pass, which walks the chain in reverse with the transposed GEMM.
#include "twog_common.h"
#include "graph_cache.h"

namespace {

constexpr int MAXS = 8;
struct FwdGroup { twog_gru_step_t s[MAXS]; };
struct BwdGroup { twog_gru_step_bwd_t s[MAXS]; };

__device__ __forceinline__ float gate_u(const float* u, int64_t ldo, int64_t ldi, int inner, int r) {
    if (!u) return 1.0f;
    if (inner <= 1) return u[(int64_t)r * ldo];
    const int o = r / inner;
    return u[(int64_t)o * ldo + (int64_t)(r - o * inner) * ldi];
}

__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const FwdGroup g) {
    const twog_gru_step_t& S = g.s[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= S.rows) return;
    const int H = S.hidden;
    const float* gi = twog_row_ptr(S.gi, r);
    const float* gi2 = S.gi2.ptr ? twog_row_ptr(S.gi2, r) : nullptr;
    const float* gh = twog_row_ptr(S.gh, r);
    const float* hp = S.h_prev.ptr ? twog_row_ptr(S.h_prev, r) : nullptr;
    float* ho = twog_row_ptr(S.h_out, r);
    float* sv = S.save.ptr ? twog_row_ptr(S.save, r) : nullptr;
    const float uu = gate_u(S.u, S.u_ld_outer, S.u_ld_inner, S.u_inner, r);
#pragma unroll 2
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float ir = gi[j], iz = gi[H + j], in_ = gi[2 * H + j];
        if (gi2) {
            ir += gi2[j];
            iz += gi2[H + j];
            in_ += gi2[2 * H + j];
        }
        const float hr = gh[j], hz = gh[H + j], hn = gh[2 * H + j];
        const float rg = 1.0f / (1.0f + expf(-(ir + hr)));
        const float z = 1.0f / (1.0f + expf(-(iz + hz)));
        const float n = tanhf(in_ + rg * hn);
        const float h0 = hp ? hp[j] : 0.f;
        const float gnew = (1.0f - z) * n + z * h0;
        ho[j] = S.u ? uu * gnew + (1.0f - uu) * h0 : gnew;
        if (sv) {
            sv[j] = rg;
            sv[H + j] = z;
            sv[2 * H + j] = n;
            sv[3 * H + j] = hn;
        }
    }
}

// The same step, four hidden units per thread (16-byte accesses; round 6): the launch of the segment level's forward chain
// at a real batch moves 39 MB per step and was at 3.4 TB/s with scalar accesses. Same arithmetic per element: bit-identical.
// Host-side condition: hidden % 4 == 0, hidden / 4 <= 256, every row pointer 16-byte aligned (gru_step_vec_ok).
__global__ __launch_bounds__(256) void gru_step_fwd_vec_kernel(const FwdGroup g) {
    const twog_gru_step_t& S = g.s[blockIdx.y];
    const int H = S.hidden, lpr = H >> 2, rpb = (int)blockDim.x / lpr;
    const int rl = (int)threadIdx.x / lpr, q = (int)threadIdx.x - rl * lpr;
    const int r = (int)blockIdx.x * rpb + rl;
    if (rl >= rpb || r >= S.rows) return;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int j = 4 * q;
    const float* gi = twog_row_ptr(S.gi, r);
    const float* gh = twog_row_ptr(S.gh, r);
    f4 ir = *reinterpret_cast<const f4*>(gi + j), iz = *reinterpret_cast<const f4*>(gi + H + j), in_ = *reinterpret_cast<const f4*>(gi + 2 * H + j);
    if (S.gi2.ptr) {
        const float* gi2 = twog_row_ptr(S.gi2, r);
        ir += *reinterpret_cast<const f4*>(gi2 + j);
        iz += *reinterpret_cast<const f4*>(gi2 + H + j);
        in_ += *reinterpret_cast<const f4*>(gi2 + 2 * H + j);
    }
    const f4 hr = *reinterpret_cast<const f4*>(gh + j), hz = *reinterpret_cast<const f4*>(gh + H + j), hn = *reinterpret_cast<const f4*>(gh + 2 * H + j);
    f4 h0 = {0.f, 0.f, 0.f, 0.f};
    if (S.h_prev.ptr) h0 = *reinterpret_cast<const f4*>(twog_row_ptr(S.h_prev, r) + j);
    const float uu = gate_u(S.u, S.u_ld_outer, S.u_ld_inner, S.u_inner, r);
    f4 rg, z, n, ho;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rg[k] = 1.0f / (1.0f + expf(-(ir[k] + hr[k])));
        z[k] = 1.0f / (1.0f + expf(-(iz[k] + hz[k])));
        n[k] = tanhf(in_[k] + rg[k] * hn[k]);
        const float gnew = (1.0f - z[k]) * n[k] + z[k] * h0[k];
        ho[k] = S.u ? uu * gnew + (1.0f - uu) * h0[k] : gnew;
    }
    *reinterpret_cast<f4*>(twog_row_ptr(S.h_out, r) + j) = ho;
    if (S.save.ptr) {
        float* sv = twog_row_ptr(S.save, r);
        *reinterpret_cast<f4*>(sv + j) = rg;
        *reinterpret_cast<f4*>(sv + H + j) = z;
        *reinterpret_cast<f4*>(sv + 2 * H + j) = n;
        *reinterpret_cast<f4*>(sv + 3 * H + j) = hn;
    }
}

inline bool rows_vec_ok(const twog_rows_t& m) {
    if (!m.ptr) return true;
    return (reinterpret_cast<uintptr_t>(m.ptr) & 15) == 0 && (m.ld_outer & 3) == 0 && (m.inner <= 1 || (m.ld_inner & 3) == 0);
}
inline bool gru_step_vec_ok(const twog_gru_step_t& S) {
    return (S.hidden & 3) == 0 && S.hidden >= 64 && (S.hidden >> 2) <= 256 && 256 % (S.hidden >> 2) == 0 && rows_vec_ok(S.gi) &&
           rows_vec_ok(S.gi2) && rows_vec_ok(S.gh) && rows_vec_ok(S.h_prev) && rows_vec_ok(S.h_out) && rows_vec_ok(S.save);
}

__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const BwdGroup g) {
    __shared__ float red[16];
    const twog_gru_step_bwd_t& S = g.s[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= S.rows) return;
    const int H = S.hidden;
    const float* dh = twog_row_ptr(S.dh, r);
    const float* dh2 = S.dh2.ptr ? twog_row_ptr(S.dh2, r) : nullptr;
    const float* sv = twog_row_ptr(S.save, r);
    const float* hp = S.h_prev.ptr ? twog_row_ptr(S.h_prev, r) : nullptr;
    float* dgi = twog_row_ptr(S.dgi, r);
    float* dgh = twog_row_ptr(S.dgh, r);
    float* dhp = twog_row_ptr(S.dh_prev, r);
    const float uu = gate_u(S.u, S.u_ld_outer, S.u_ld_inner, S.u_inner, r);
    float du = 0.f;
#pragma unroll 2
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float d = dh[j];
        if (dh2) d += dh2[j];
        const float rg = sv[j], z = sv[H + j], n = sv[2 * H + j], hn = sv[3 * H + j];
        const float h0 = hp ? hp[j] : 0.f;
        const float gnew = (1.0f - z) * n + z * h0;
        du += d * (gnew - h0);
        const float dg = S.u ? uu * d : d;
        float dprev = S.u ? (1.0f - uu) * d : 0.f;
        const float dn = dg * (1.0f - z);
        const float dz = dg * (h0 - n);
        dprev += dg * z;
        const float dn_pre = dn * (1.0f - n * n);
        const float dr_pre = dn_pre * hn * rg * (1.0f - rg);
        const float dz_pre = dz * z * (1.0f - z);
        dgi[j] = dr_pre;
        dgi[H + j] = dz_pre;
        dgi[2 * H + j] = dn_pre;
        dgh[j] = dr_pre;
        dgh[H + j] = dz_pre;
        dgh[2 * H + j] = dn_pre * rg;
        dhp[j] = S.dh_prev_accumulate ? dhp[j] + dprev : dprev;
    }
    if (S.du) {  // uniform per block
        const float t = block_sum(du, red);
        if (threadIdx.x == 0) {
            int64_t off;
            if (S.u_inner <= 1) off = (int64_t)r * S.u_ld_outer;
            else { const int o = r / S.u_inner; off = (int64_t)o * S.u_ld_outer + (int64_t)(r - o * S.u_inner) * S.u_ld_inner; }
            atomicAdd(&S.du[off], t);
        }
    }
}

inline twog_rows_t rows_be(const float* cbase, int E, int64_t row_w, int T) {
    float* base = const_cast<float*>(cbase);
    // rows (b, e) of a [bs][T][E][row_w] tensor at a fixed t (base already offset to t and column)
    twog_rows_t r;
    r.ptr = base;
    r.inner = E;
    r.ld_inner = row_w;
    r.ld_outer = (int64_t)T * E * row_w;
    r.pad_ = 0;
    return r;
}
inline twog_rows_t rows_plain(const float* cbase, int64_t ld) {
    float* base = const_cast<float*>(cbase);
    twog_rows_t r;
    r.ptr = base; r.inner = 1; r.ld_inner = ld; r.ld_outer = ld; r.pad_ = 0;
    return r;
}

}  // namespace

extern "C" int twog_gru_step_fwd(const twog_gru_step_t* steps, int n_steps, void* stream) {
    int done = 0;
    while (done < n_steps) {
        const int n = (n_steps - done) < MAXS ? (n_steps - done) : MAXS;
        FwdGroup g;
        int maxrows = 0, maxh = 0;
        for (int i = 0; i < n; ++i) {
            g.s[i] = steps[done + i];
            if (g.s[i].rows > maxrows) maxrows = g.s[i].rows;
            if (g.s[i].hidden > maxh) maxh = g.s[i].hidden;
        }
        if (maxrows > 0) {
            static const int vec_on = getenv("TWOG_GRU_STEP_VEC") ? atoi(getenv("TWOG_GRU_STEP_VEC")) : 1;
            bool vec = vec_on != 0;
            for (int i = 0; i < n; ++i) vec = vec && gru_step_vec_ok(g.s[i]) && g.s[i].hidden == maxh;
            if (vec) {
                const int rpb = 256 / (maxh >> 2);
                hipLaunchKernelGGL(gru_step_fwd_vec_kernel, dim3((maxrows + rpb - 1) / rpb, n), dim3(256), 0, (hipStream_t)stream, g);
            } else {
                int bt = maxh >= 256 ? 256 : ((maxh + 63) / 64) * 64;
                hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(maxrows, n), dim3(bt), 0, (hipStream_t)stream, g);
            }
            TWOG_CHECK_LAUNCH();
        }
        done += n;
    }
    return 0;
}

extern "C" int twog_gru_step_bwd(const twog_gru_step_bwd_t* steps, int n_steps, void* stream) {
    int done = 0;
    while (done < n_steps) {
        const int n = (n_steps - done) < MAXS ? (n_steps - done) : MAXS;
        BwdGroup g;
        int maxrows = 0, maxh = 0;
        for (int i = 0; i < n; ++i) {
            g.s[i] = steps[done + i];
            if (g.s[i].rows > maxrows) maxrows = g.s[i].rows;
            if (g.s[i].hidden > maxh) maxh = g.s[i].hidden;
        }
        if (maxrows > 0) {
            int bt = maxh >= 256 ? 256 : ((maxh + 63) / 64) * 64;
            hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(maxrows, n), dim3(bt), 0, (hipStream_t)stream, g);
            TWOG_CHECK_LAUNCH();
        }
        done += n;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Frame-level BiGRU recurrence for up to 4 entity types at once (humans, objects, geometry).
// ---------------------------------------------------------------------------------------------------------------
static int bigru_fwd_impl(const twog_bigru_t* types, int n_types, int bs, int T, int hidden, void* chain_ws,
                          size_t chain_ws_bytes, void* stream) {
    if (n_types > 4) return -1;
    const int h = hidden;
    for (int s = 0; s < T; ++s) {
        twog_gemm_t gm[8];
        twog_gru_step_t st[8];
        int n = 0;
        for (int k = 0; k < n_types; ++k) {
            const twog_bigru_t& Y = types[k];
            const int E = Y.E, rows = bs * E;
            for (int dir = 0; dir < 2; ++dir) {
                const int t = dir == 0 ? s : T - 1 - s;
                const int tp = dir == 0 ? t - 1 : t + 1;
                float* tmp = Y.tmp_gh + (int64_t)dir * rows * 3 * h;
                twog_rows_t hprev = s == 0 ? rows_plain(Y.zeros, h)
                                           : rows_be(Y.out + (int64_t)tp * E * 2 * h + dir * h, E, 2 * h, T);
                twog_gemm_t& G = gm[n];
                G.A = hprev;
                G.B = rows_plain((dir == 0 ? Y.w_hh_f : Y.w_hh_r), h);
                G.C = rows_plain(tmp, 3 * h);
                G.bias = dir == 0 ? Y.b_hh_f : Y.b_hh_r;
                G.M = rows; G.N = 3 * h; G.K = h; G.act = 0; G.accumulate = 0; G.batch = 1;
                G.a_batch_stride = G.b_batch_stride = G.c_batch_stride = 0; G.a_colsum = nullptr; G.a_colsum_accumulate = 0; G.pad2_ = 0;
                twog_gru_step_t& S = st[n];
                S.gi = rows_be(Y.gi + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T);
                S.gi2.ptr = nullptr; S.gi2.inner = 1; S.gi2.ld_inner = S.gi2.ld_outer = 0;
                S.gh = rows_plain(tmp, 3 * h);
                S.h_prev = hprev;
                if (s == 0) S.h_prev.ptr = nullptr;
                S.h_out = rows_be(Y.out + (int64_t)t * E * 2 * h + dir * h, E, 2 * h, T);
                S.save = rows_be(Y.save + ((int64_t)dir * bs * T * E + (int64_t)t * E) * 4 * h, E, 4 * h, T);
                S.u = nullptr; S.u_ld_outer = S.u_ld_inner = 0; S.u_inner = 1;
                S.rows = rows; S.hidden = h;
                ++n;
            }
        }
        // one launch: the W_hh products on gate-aware tiles with the gate math in the epilogue (gemm_f32.hip); shapes it
        // does not serve (hidden size not a multiple of the k-tile) take the GEMM + gate kernel pair
        int rc = twog_internal_gemm_gru_fwd(gm, nullptr, st, n, 0, stream);
        if (rc < 0) return rc;
        if (rc == 0) continue;
        rc = twog_gemm_f32_chain(gm, n, 0, 0, chain_ws, chain_ws_bytes, stream);
        if (rc) return rc;
        rc = twog_gru_step_fwd(st, n, stream);
        if (rc) return rc;
    }
    return 0;
}

// Backward through time. d_out: gradient wrt `out` [bs][T][E][2h]; writes d_gi [bs][T][E][6h] and
// d_gh [bs][T][E][6h] (the caller turns them into dX / dW_ih / dW_hh / biases with large GEMMs).
static int bigru_bwd_impl(const twog_bigru_bwd_t* types, int n_types, int bs, int T, int hidden, void* chain_ws,
                          size_t chain_ws_bytes, void* stream) {
    if (n_types > 4) return -1;
    const int h = hidden;
    // gate descriptors + the carry GEMM of chain step s, one pair per (type, direction)
    auto make_step = [&](int s, twog_gru_step_bwd_t* st, twog_gemm_t* gm) {
        int n = 0;
        for (int k = 0; k < n_types; ++k) {
            const twog_bigru_bwd_t& Y = types[k];
            const int E = Y.E, rows = bs * E;
            for (int dir = 0; dir < 2; ++dir) {
                const int t = dir == 0 ? s : T - 1 - s;
                const int tp = dir == 0 ? t - 1 : t + 1;
                float* carry = Y.carry + (int64_t)dir * rows * h;
                twog_gru_step_bwd_t& S = st[n];
                S.dh = rows_be(Y.d_out + (int64_t)t * E * 2 * h + dir * h, E, 2 * h, T);
                if (s == T - 1) { S.dh2.ptr = nullptr; S.dh2.inner = 1; S.dh2.ld_inner = S.dh2.ld_outer = 0; }
                else S.dh2 = rows_plain(carry, h);
                S.save = rows_be(Y.save + ((int64_t)dir * bs * T * E + (int64_t)t * E) * 4 * h, E,
                                 4 * h, T);
                if (s == 0) { S.h_prev.ptr = nullptr; S.h_prev.inner = 1; S.h_prev.ld_inner = S.h_prev.ld_outer = 0; }
                else S.h_prev = rows_be(Y.out + (int64_t)tp * E * 2 * h + dir * h, E, 2 * h, T);
                S.dgi = rows_be(Y.d_gi + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T);
                S.dgh = rows_be(Y.d_gh + (int64_t)t * E * 6 * h + dir * 3 * h, E, 6 * h, T);
                S.dh_prev = rows_plain(carry, h);
                S.u = nullptr; S.du = nullptr; S.u_ld_outer = S.u_ld_inner = 0; S.u_inner = 1;
                S.rows = rows; S.hidden = h; S.dh_prev_accumulate = 0;
                // carried gradient: carry += d_gh W_hh   (reduction over the 3h gate rows of W_hh: k-major B)
                if (gm) {
                    twog_gemm_t& G = gm[n];
                    G.A = S.dgh;
                    G.B = rows_plain((dir == 0 ? Y.w_hh_f : Y.w_hh_r), h);
                    G.C = rows_plain(carry, h);
                    G.bias = nullptr;
                    G.M = rows; G.N = h; G.K = 3 * h; G.act = 0; G.accumulate = 1; G.batch = 1;
                    G.a_batch_stride = G.b_batch_stride = G.c_batch_stride = 0; G.a_colsum = nullptr; G.a_colsum_accumulate = 0; G.pad2_ = 0;
                }
                ++n;
            }
        }
        return n;
    };
    // The carry GEMM of step s is the only writer of the carried gradient, so the gate backward of step s-1 runs in its
    // epilogue (gemm_f32.hip, GateArgs) instead of in a launch of its own; the first step has no GEMM before it.
    bool fuse = T > 1 && (h + 63) / 64 <= 8 && getenv("TWOG_NO_GATE_FUSION") == nullptr;
    for (int s = T - 1; s >= 0; --s) {
        twog_gemm_t gm[8];
        twog_gru_step_bwd_t st[8], nxt[8];
        const int n = make_step(s, st, gm);
        int rc;
        if (!fuse || s == T - 1) {
            rc = twog_gru_step_bwd(st, n, stream);
            if (rc) return rc;
        }
        if (s > 0) {
            rc = 1;
            if (fuse) {
                make_step(s - 1, nxt, nullptr);
                rc = twog_internal_gemm_gate_bwd(gm, n, nxt, nullptr, 0, chain_ws, chain_ws_bytes, stream);
                if (rc < 0) return rc;
                if (rc == 1) {  // shape not served by the fused kernel: decided at the first step
                    if (s != T - 1) return -120;
                    fuse = false;
                }
            }
            if (rc == 1) rc = twog_gemm_f32_chain(gm, n, 0, 1, chain_ws, chain_ws_bytes, stream);
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" int twog_bigru_fwd(const twog_bigru_t* types, int n_types, int bs, int T, int hidden, void* chain_ws,
                              size_t chain_ws_bytes, void* stream) {
    if (n_types > 4 || n_types < 0) return -1;
    const int dims[6] = {0x11, n_types, bs, T, hidden, twog_internal_gru_fwd_mode()};
    twog_graph::Desc key;
    key.pod(dims).add(types, sizeof(twog_bigru_t) * n_types).pod(chain_ws).pod(chain_ws_bytes);
    return twog_graph::run(key, (hipStream_t)stream, [&](hipStream_t st) {
        return bigru_fwd_impl(types, n_types, bs, T, hidden, chain_ws, chain_ws_bytes, st);
    });
}

extern "C" int twog_bigru_bwd(const twog_bigru_bwd_t* types, int n_types, int bs, int T, int hidden, void* chain_ws,
                              size_t chain_ws_bytes, void* stream) {
    if (n_types > 4 || n_types < 0) return -1;
    const int dims[5] = {0x22, n_types, bs, T, hidden};
    twog_graph::Desc key;
    key.pod(dims).add(types, sizeof(twog_bigru_bwd_t) * n_types).pod(chain_ws).pod(chain_ws_bytes);
    return twog_graph::run(key, (hipStream_t)stream, [&](hipStream_t st) {
        return bigru_bwd_impl(types, n_types, bs, T, hidden, chain_ws, chain_ws_bytes, st);
    });
}
