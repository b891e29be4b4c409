// Fusion-level attention message passing between entities (humans, objects, geometry node).
//
// Reference: the ten *_message methods (vhoi/models.py:1004-1475) + compute_non_relational_message (:1693-1718) +
// compute_attention_weights (:1721-1754) for message_type 'v2', granularity 'v1' (the sender's message does not
// depend on the receiver), aggregation 'att', attention style 'v2'/'v3' (dot / scaled dot product).
// The reference runs this per time step, per receiver, per sender in Python; here ONE workgroup handles ONE instance
// (a (clip, frame) at frame level, a clip at segment level): all entity feature vectors of the instance are staged
// in LDS once, every needed pairwise score is a wave-level dot product (shuffle reduction), the masked softmax
// (-inf on virtual senders, NaN -> 0 when every sender is virtual, :1750-1753) runs on a handful of lanes, and the
// weighted message sums are written lane-contiguously straight into the caller's concatenated entity rows.
// Sender messages are computed once per (sender, relation) by the MFMA GEMM, not once per receiver.
#include "twog_common.h"

namespace {

constexpr int MAX_H = 4, MAX_O = 12, MAX_E = MAX_H + MAX_O;
constexpr int MAXG = 4;
struct FwdGroup { twog_attn_t a[MAXG]; int staged; };
struct BwdGroup { twog_attn_bwd_t a[MAXG]; int staged; };

// Latency-bound calls (few instances, e.g. one per clip inside the segment-level time loop) first copy every message /
// gradient row of the instance into LDS with one burst of 16-byte loads, so the rest of the kernel never waits on HBM/L2
// again; the relation descriptor is re-pointed at the LDS copy. Throughput-bound calls (one instance per (clip, frame))
// keep streaming from global memory at high occupancy.
__device__ __forceinline__ void stage_rows_lds(twog_rows_t& rel, int first_row, int n_rows, int width, float*& cursor) {
    if (!rel.ptr) return;
    const int w4 = width >> 2;
    for (int i = threadIdx.x; i < n_rows * w4; i += blockDim.x) {
        const int r = i / w4, c = (i - r * w4) * 4;
        *reinterpret_cast<float4*>(cursor + r * width + c) =
            *reinterpret_cast<const float4*>(twog_row_ptr(rel, first_row + r) + c);
    }
    rel.ptr = cursor; rel.inner = 1; rel.ld_outer = width; rel.ld_inner = width;
    cursor += n_rows * width;
}

// layout of the saved attention weights of one instance
__device__ __forceinline__ int att_off_hh(int, int) { return 0; }
__device__ __forceinline__ int att_off_oh(int H, int) { return H * H; }
__device__ __forceinline__ int att_off_ho(int H, int O) { return H * H + H * O; }
__device__ __forceinline__ int att_off_oo(int H, int O) { return H * H + 2 * H * O; }

// softmax over senders for one receiver: score[s] valid where ok(s). NaN->0 semantics when nothing is valid.
__device__ __forceinline__ void masked_softmax(const float* score, const bool* ok, int S, float* w) {
    float m = -INFINITY;
    for (int s = 0; s < S; ++s)
        if (ok[s]) m = fmaxf(m, score[s]);
    float sum = 0.f;
    for (int s = 0; s < S; ++s) {
        w[s] = ok[s] ? expf(score[s] - m) : 0.f;
        sum += w[s];
    }
    for (int s = 0; s < S; ++s) w[s] = ok[s] ? w[s] / sum : 0.f;
}

__device__ void compute_weights(const twog_attn_t& A, const float* sF, int inst, float* sG, float* sW, float* sMask) {
    const int H = A.H, O = A.O, E = H + O, D = A.D;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int clip = inst / A.inst_per_clip;
    if (threadIdx.x < O) sMask[threadIdx.x] = A.obj_mask ? A.obj_mask[(int64_t)clip * O + threadIdx.x] : 1.f;
    // Gram matrix of the entity features (upper triangle incl. diagonal unused entries are cheap to skip)
    for (int p = wv; p < E * E; p += nw) {
        const int a = p / E, b = p - a * E;
        if (b < a) continue;
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc = fmaf(sF[a * D + d], sF[b * D + d], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            sG[a * E + b] = acc * A.scale;
            sG[b * E + a] = acc * A.scale;
        }
    }
    __syncthreads();
    // one thread per (relation, receiver)
    const int nrec = 2 * H + 2 * O;
    for (int i = threadIdx.x; i < nrec; i += blockDim.x) {
        float sc[MAX_E], w[MAX_E];
        bool ok[MAX_E];
        if (i < H) {  // hh: receiver human i, senders humans != i
            const int h = i;
            for (int s = 0; s < H; ++s) { sc[s] = sG[h * E + s]; ok[s] = (s != h); }
            masked_softmax(sc, ok, H, w);
            for (int s = 0; s < H; ++s) sW[att_off_hh(H, O) + h * H + s] = A.msg_hh.ptr ? w[s] : 0.f;
        } else if (i < 2 * H) {  // oh: receiver human, senders objects (masked)
            const int h = i - H;
            for (int s = 0; s < O; ++s) { sc[s] = sG[h * E + H + s]; ok[s] = sMask[s] != 0.f; }
            masked_softmax(sc, ok, O, w);
            for (int s = 0; s < O; ++s) sW[att_off_oh(H, O) + h * O + s] = A.msg_oh.ptr ? w[s] : 0.f;
        } else if (i < 2 * H + O) {  // ho: receiver object, senders humans
            const int k = i - 2 * H;
            for (int s = 0; s < H; ++s) { sc[s] = sG[(H + k) * E + s]; ok[s] = true; }
            masked_softmax(sc, ok, H, w);
            for (int s = 0; s < H; ++s) sW[att_off_ho(H, O) + k * H + s] = A.msg_ho.ptr ? w[s] : 0.f;
        } else {  // oo: receiver object k, senders objects != k (masked)
            const int k = i - 2 * H - O;
            for (int s = 0; s < O; ++s) { sc[s] = sG[(H + k) * E + H + s]; ok[s] = (s != k) && sMask[s] != 0.f; }
            masked_softmax(sc, ok, O, w);
            for (int s = 0; s < O; ++s) sW[att_off_oo(H, O) + k * O + s] = A.msg_oo.ptr ? w[s] : 0.f;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void stage_features(const twog_attn_t& A, int inst, float* sF) {
    const int H = A.H, O = A.O, D = A.D, d4 = D >> 2;
    for (int i = threadIdx.x; i < (H + O) * d4; i += blockDim.x) {
        const int e = i / d4, c = (i - e * d4) * 4;
        const float* src = e < H ? twog_row_ptr(A.feat_h, inst * H + e) : twog_row_ptr(A.feat_o, inst * O + (e - H));
        *reinterpret_cast<float4*>(sF + e * D + c) = *reinterpret_cast<const float4*>(src + c);
    }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const FwdGroup g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    twog_attn_t A = g.a[blockIdx.y];
    const int inst = blockIdx.x;
    if (inst >= A.n_inst) return;
    const int H = A.H, O = A.O, E = H + O, D = A.D, hid = A.hidden;
    float* sF = sm;                 // [E][D]
    float* sG = sF + E * D;         // [E][E]
    float* sW = sG + MAX_E * MAX_E; // [H*H + 2*H*O + O*O]
    float* sMask = sW + (MAX_H * MAX_H + 2 * MAX_H * MAX_O + MAX_O * MAX_O);
    int ih = inst * H, io = inst * O, ig = inst;  // first message row of this instance per sender type
    stage_features(A, inst, sF);
    if (g.staged) {
        float* cur = sMask + MAX_O + 4;
        cur = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(cur) + 15) & ~uintptr_t(15));
        stage_rows_lds(A.msg_hh, ih, H, hid, cur);
        stage_rows_lds(A.msg_ho, ih, H, hid, cur);
        stage_rows_lds(A.msg_oh, io, O, hid, cur);
        stage_rows_lds(A.msg_oo, io, O, hid, cur);
        stage_rows_lds(A.msg_so, ig, 1, hid, cur);
        stage_rows_lds(A.msg_sh, ig, 1, hid, cur);
        ih = io = ig = 0;
    }
    __syncthreads();
    compute_weights(A, sF, inst, sG, sW, sMask);
    const int natt = H * H + 2 * H * O + O * O;
    if (A.att)
        for (int i = threadIdx.x; i < natt; i += blockDim.x) A.att[(int64_t)inst * natt + i] = sW[i];
    // weighted sums, one output column per thread (lane-contiguous loads/stores)
    for (int j = threadIdx.x; j < hid; j += blockDim.x) {
        float m[MAX_O];
        if (A.msg_hh.ptr) {
            for (int s = 0; s < H; ++s) m[s] = twog_row_ptr(A.msg_hh, ih + s)[j];
            for (int h = 0; h < H; ++h) {
                float acc = 0.f;
                for (int s = 0; s < H; ++s) acc = fmaf(sW[att_off_hh(H, O) + h * H + s], m[s], acc);
                twog_row_ptr(A.out_hh, inst * H + h)[j] = acc;
            }
        }
        if (A.msg_oh.ptr) {
            for (int s = 0; s < O; ++s) m[s] = twog_row_ptr(A.msg_oh, io + s)[j];
            for (int h = 0; h < H; ++h) {
                float acc = 0.f;
                for (int s = 0; s < O; ++s) acc = fmaf(sW[att_off_oh(H, O) + h * O + s], m[s], acc);
                twog_row_ptr(A.out_oh, inst * H + h)[j] = acc;
            }
        }
        if (A.msg_sh.ptr) {
            const float v = twog_row_ptr(A.msg_sh, ig)[j];
            for (int h = 0; h < H; ++h) twog_row_ptr(A.out_sh, inst * H + h)[j] = v;
        }
        if (A.msg_ho.ptr) {
            for (int s = 0; s < H; ++s) m[s] = twog_row_ptr(A.msg_ho, ih + s)[j];
            for (int k = 0; k < O; ++k) {
                float acc = 0.f;
                for (int s = 0; s < H; ++s) acc = fmaf(sW[att_off_ho(H, O) + k * H + s], m[s], acc);
                twog_row_ptr(A.out_ho, inst * O + k)[j] = A.recv_mask_ho ? acc * sMask[k] : acc;
            }
        }
        if (A.msg_so.ptr) {
            const float v = twog_row_ptr(A.msg_so, ig)[j];
            for (int k = 0; k < O; ++k) twog_row_ptr(A.out_so, inst * O + k)[j] = A.recv_mask_ho ? v * sMask[k] : v;
        }
        if (A.msg_oo.ptr) {
            for (int s = 0; s < O; ++s) m[s] = twog_row_ptr(A.msg_oo, io + s)[j];
            for (int k = 0; k < O; ++k) {
                float acc = 0.f;
                for (int s = 0; s < O; ++s) acc = fmaf(sW[att_off_oo(H, O) + k * O + s], m[s], acc);
                twog_row_ptr(A.out_oo, inst * O + k)[j] = acc;
            }
        }
    }
}

// wave-level dot product of two global rows of length n
__device__ __forceinline__ float wave_dot(const float* a, const float* b, int n, int lane) {
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) acc = fmaf(a[j], b[j], acc);
    return wave_sum(acc);
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const BwdGroup g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    twog_attn_bwd_t B = g.a[blockIdx.y];
    twog_attn_t& A = B.f;
    const int inst = blockIdx.x;
    if (inst >= A.n_inst) return;
    const int H = A.H, O = A.O, E = H + O, D = A.D, hid = A.hidden;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int natt = H * H + 2 * H * O + O * O;
    constexpr int NATT_MAX = MAX_H * MAX_H + 2 * MAX_H * MAX_O + MAX_O * MAX_O;
    float* sW = sm;               // saved weights [natt]
    float* sdW = sW + NATT_MAX;   // dL/dw then dscore [natt]
    float* sC = sdW + NATT_MAX;   // [E][E] coefficients of dF
    float* sMask = sC + MAX_E * MAX_E;
    const int clip = inst / A.inst_per_clip;
    // first rows of this instance: sender messages (mh/mo/mg) and incoming gradients (rh/ro)
    int mh = inst * H, mo = inst * O, mg = inst, rh = inst * H, ro = inst * O;
    for (int i = threadIdx.x; i < natt; i += blockDim.x) sW[i] = A.att[(int64_t)inst * natt + i];
    if (threadIdx.x < O) sMask[threadIdx.x] = A.obj_mask ? A.obj_mask[(int64_t)clip * O + threadIdx.x] : 1.f;
    for (int i = threadIdx.x; i < MAX_E * MAX_E; i += blockDim.x) sC[i] = 0.f;
    if (g.staged) {
        float* cur = sMask + MAX_O + 4;
        cur = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(cur) + 15) & ~uintptr_t(15));
        stage_rows_lds(A.msg_hh, mh, H, hid, cur);
        stage_rows_lds(A.msg_ho, mh, H, hid, cur);
        stage_rows_lds(A.msg_oh, mo, O, hid, cur);
        stage_rows_lds(A.msg_oo, mo, O, hid, cur);
        stage_rows_lds(A.msg_so, mg, 1, hid, cur);
        stage_rows_lds(A.msg_sh, mg, 1, hid, cur);
        if (A.msg_hh.ptr) stage_rows_lds(B.dout_hh, rh, H, hid, cur);
        if (A.msg_oh.ptr) stage_rows_lds(B.dout_oh, rh, H, hid, cur);
        if (A.msg_sh.ptr) stage_rows_lds(B.dout_sh, rh, H, hid, cur);
        if (A.msg_ho.ptr) stage_rows_lds(B.dout_ho, ro, O, hid, cur);
        if (A.msg_so.ptr) stage_rows_lds(B.dout_so, ro, O, hid, cur);
        if (A.msg_oo.ptr) stage_rows_lds(B.dout_oo, ro, O, hid, cur);
        mh = mo = mg = rh = ro = 0;
    }
    __syncthreads();
    // dL/dw[r][s] = recv_mask_r * <dout[r], msg[s]>   (one wave per pair)
    for (int p = wv; p < natt; p += nw) {
        float v = 0.f;
        if (p < H * H) {
            const int r = p / H, s = p - r * H;
            if (A.msg_hh.ptr && r != s)
                v = wave_dot(twog_row_ptr(B.dout_hh, rh + r), twog_row_ptr(A.msg_hh, mh + s), hid, lane);
        } else if (p < H * H + H * O) {
            const int q = p - H * H, r = q / O, s = q - r * O;
            if (A.msg_oh.ptr)
                v = wave_dot(twog_row_ptr(B.dout_oh, rh + r), twog_row_ptr(A.msg_oh, mo + s), hid, lane);
        } else if (p < H * H + 2 * H * O) {
            const int q = p - H * H - H * O, r = q / H, s = q - r * H;
            if (A.msg_ho.ptr)
                v = (A.recv_mask_ho ? sMask[r] : 1.f) *
                    wave_dot(twog_row_ptr(B.dout_ho, ro + r), twog_row_ptr(A.msg_ho, mh + s), hid, lane);
        } else {
            const int q = p - H * H - 2 * H * O, r = q / O, s = q - r * O;
            if (A.msg_oo.ptr && r != s)
                v = wave_dot(twog_row_ptr(B.dout_oo, ro + r), twog_row_ptr(A.msg_oo, mo + s), hid, lane);
        }
        if (lane == 0) sdW[p] = v;
    }
    __syncthreads();
    // softmax backward per receiver: dscore = w * (dw - sum_s w dw) * scale ; scatter into the dF coefficient matrix
    if (threadIdx.x < 2 * H + 2 * O) {
        const int i = threadIdx.x;
        int off, S;
        if (i < H) { off = att_off_hh(H, O) + i * H; S = H; }
        else if (i < 2 * H) { off = att_off_oh(H, O) + (i - H) * O; S = O; }
        else if (i < 2 * H + O) { off = att_off_ho(H, O) + (i - 2 * H) * H; S = H; }
        else { off = att_off_oo(H, O) + (i - 2 * H - O) * O; S = O; }
        float t = 0.f;
        for (int s = 0; s < S; ++s) t = fmaf(sW[off + s], sdW[off + s], t);
        for (int s = 0; s < S; ++s) sdW[off + s] = sW[off + s] * (sdW[off + s] - t) * A.scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // tiny (<= ~100 entries): serial, deterministic
        for (int r = 0; r < H; ++r)
            for (int s = 0; s < H; ++s) {
                const float v = sdW[att_off_hh(H, O) + r * H + s];
                sC[r * MAX_E + s] += v;
                sC[s * MAX_E + r] += v;
            }
        for (int r = 0; r < H; ++r)
            for (int s = 0; s < O; ++s) {
                const float v = sdW[att_off_oh(H, O) + r * O + s];
                sC[r * MAX_E + H + s] += v;
                sC[(H + s) * MAX_E + r] += v;
            }
        for (int r = 0; r < O; ++r)
            for (int s = 0; s < H; ++s) {
                const float v = sdW[att_off_ho(H, O) + r * H + s];
                sC[(H + r) * MAX_E + s] += v;
                sC[s * MAX_E + H + r] += v;
            }
        for (int r = 0; r < O; ++r)
            for (int s = 0; s < O; ++s) {
                const float v = sdW[att_off_oo(H, O) + r * O + s];
                sC[(H + r) * MAX_E + H + s] += v;
                sC[(H + s) * MAX_E + H + r] += v;
            }
    }
    __syncthreads();
    // gradient wrt sender messages: dmsg[s] = sum_r w[r][s] * recv_mask_r * dout[r]  (optionally times ReLU'(msg))
    for (int j = threadIdx.x; j < hid; j += blockDim.x) {
        float gr[MAX_O];
        if (A.msg_hh.ptr) {
            for (int r = 0; r < H; ++r) gr[r] = twog_row_ptr(B.dout_hh, rh + r)[j];
            for (int s = 0; s < H; ++s) {
                float acc = 0.f;
                for (int r = 0; r < H; ++r) acc = fmaf(sW[att_off_hh(H, O) + r * H + s], gr[r], acc);
                if (B.relu_mask_dmsg && !(twog_row_ptr(A.msg_hh, mh + s)[j] > 0.f)) acc = 0.f;
                twog_row_ptr(B.dmsg_hh, inst * H + s)[j] = acc;
            }
        }
        if (A.msg_oh.ptr) {
            for (int r = 0; r < H; ++r) gr[r] = twog_row_ptr(B.dout_oh, rh + r)[j];
            for (int s = 0; s < O; ++s) {
                float acc = 0.f;
                for (int r = 0; r < H; ++r) acc = fmaf(sW[att_off_oh(H, O) + r * O + s], gr[r], acc);
                if (B.relu_mask_dmsg && !(twog_row_ptr(A.msg_oh, mo + s)[j] > 0.f)) acc = 0.f;
                twog_row_ptr(B.dmsg_oh, inst * O + s)[j] = acc;
            }
        }
        if (A.msg_sh.ptr) {
            float acc = 0.f;
            for (int r = 0; r < H; ++r) acc += twog_row_ptr(B.dout_sh, rh + r)[j];
            if (B.relu_mask_dmsg && !(twog_row_ptr(A.msg_sh, mg)[j] > 0.f)) acc = 0.f;
            twog_row_ptr(B.dmsg_sh, inst)[j] = acc;
        }
        if (A.msg_ho.ptr) {
            for (int r = 0; r < O; ++r) gr[r] = (A.recv_mask_ho ? sMask[r] : 1.f) * twog_row_ptr(B.dout_ho, ro + r)[j];
            for (int s = 0; s < H; ++s) {
                float acc = 0.f;
                for (int r = 0; r < O; ++r) acc = fmaf(sW[att_off_ho(H, O) + r * H + s], gr[r], acc);
                if (B.relu_mask_dmsg && !(twog_row_ptr(A.msg_ho, mh + s)[j] > 0.f)) acc = 0.f;
                twog_row_ptr(B.dmsg_ho, inst * H + s)[j] = acc;
            }
        }
        if (A.msg_so.ptr) {
            float acc = 0.f;
            for (int r = 0; r < O; ++r) acc = fmaf(A.recv_mask_ho ? sMask[r] : 1.f, twog_row_ptr(B.dout_so, ro + r)[j], acc);
            if (B.relu_mask_dmsg && !(twog_row_ptr(A.msg_so, mg)[j] > 0.f)) acc = 0.f;
            twog_row_ptr(B.dmsg_so, inst)[j] = acc;
        }
        if (A.msg_oo.ptr) {
            for (int r = 0; r < O; ++r) gr[r] = twog_row_ptr(B.dout_oo, ro + r)[j];
            for (int s = 0; s < O; ++s) {
                float acc = 0.f;
                for (int r = 0; r < O; ++r) acc = fmaf(sW[att_off_oo(H, O) + r * O + s], gr[r], acc);
                if (B.relu_mask_dmsg && !(twog_row_ptr(A.msg_oo, mo + s)[j] > 0.f)) acc = 0.f;
                twog_row_ptr(B.dmsg_oo, inst * O + s)[j] = acc;
            }
        }
    }
    // gradient wrt the features: dF[a] = sum_b C[a][b] F[b]; F is read straight from global memory (one burst of E
    // independent lane-contiguous loads per column), so the backward kernel keeps no LDS copy of the features
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float f[MAX_E];
        for (int b = 0; b < E; ++b)
            f[b] = (b < H ? twog_row_ptr(A.feat_h, inst * H + b) : twog_row_ptr(A.feat_o, inst * O + (b - H)))[d];
        for (int a = 0; a < E; ++a) {
            float acc = 0.f;
            for (int b = 0; b < E; ++b) acc = fmaf(sC[a * MAX_E + b], f[b], acc);
            float* dst = a < H ? twog_row_ptr(B.dfeat_h, inst * H + a) : twog_row_ptr(B.dfeat_o, inst * O + (a - H));
            dst[d] = B.dfeat_accumulate ? dst[d] + acc : acc;
        }
    }
}

constexpr int STAGE_MAX_INST = 1024;  // calls with at most this many instances use the LDS-staged (latency) path
inline size_t n_msg_rows(const twog_attn_t& a) {
    return (size_t)(a.msg_hh.ptr ? a.H : 0) + (a.msg_ho.ptr ? a.H : 0) + (a.msg_oh.ptr ? a.O : 0) +
           (a.msg_oo.ptr ? a.O : 0) + (a.msg_so.ptr ? 1 : 0) + (a.msg_sh.ptr ? 1 : 0);
}
inline size_t n_dout_rows(const twog_attn_t& a) {
    return (size_t)(a.msg_hh.ptr ? a.H : 0) + (a.msg_oh.ptr ? a.H : 0) + (a.msg_sh.ptr ? a.H : 0) +
           (a.msg_ho.ptr ? a.O : 0) + (a.msg_so.ptr ? a.O : 0) + (a.msg_oo.ptr ? a.O : 0);
}
constexpr int NATT_MAX_H = MAX_H * MAX_H + 2 * MAX_H * MAX_O + MAX_O * MAX_O;
inline size_t lds_fwd(const twog_attn_t& a, bool staged) {
    size_t f = (size_t)(a.H + a.O) * a.D + MAX_E * MAX_E + NATT_MAX_H + MAX_O + 8;
    if (staged) f += n_msg_rows(a) * a.hidden;
    return sizeof(float) * f;
}
inline size_t lds_bwd(const twog_attn_t& a, bool staged) {
    size_t f = 2 * (size_t)NATT_MAX_H + MAX_E * MAX_E + MAX_O + 8;
    if (staged) f += (n_msg_rows(a) + n_dout_rows(a)) * a.hidden;
    return sizeof(float) * f;
}
constexpr size_t LDS_LIMIT = 160 * 1024;

}  // namespace

extern "C" int twog_attn_limits(int* max_h, int* max_o) {
    *max_h = MAX_H;
    *max_o = MAX_O;
    return 0;
}

extern "C" int twog_attn_fwd(const twog_attn_t* a, int n, void* stream) {
    if (n > MAXG) return -1;
    FwdGroup g;
    int maxinst = 0;
    for (int i = 0; i < n; ++i) {
        g.a[i] = a[i];
        if (a[i].H > MAX_H || a[i].O > MAX_O || a[i].H < 0 || a[i].O < 0 || (a[i].D & 3) || (a[i].hidden & 3)) return -2;
        if (a[i].n_inst > maxinst) maxinst = a[i].n_inst;
    }
    if (maxinst == 0) return 0;
    bool staged = maxinst <= STAGE_MAX_INST;
    size_t lds = 0;
    for (int pass = 0; pass < 2; ++pass) {
        lds = 0;
        for (int i = 0; i < n; ++i) lds = lds_fwd(a[i], staged) > lds ? lds_fwd(a[i], staged) : lds;
        if (lds <= LDS_LIMIT) break;
        staged = false;
    }
    g.staged = staged ? 1 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(maxinst, n), dim3(256), lds, (hipStream_t)stream, g);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_attn_bwd(const twog_attn_bwd_t* a, int n, void* stream) {
    if (n > MAXG) return -1;
    BwdGroup g;
    int maxinst = 0;
    for (int i = 0; i < n; ++i) {
        g.a[i] = a[i];
        if (a[i].f.H > MAX_H || a[i].f.O > MAX_O || (a[i].f.D & 3) || (a[i].f.hidden & 3)) return -2;
        if (a[i].f.n_inst > maxinst) maxinst = a[i].f.n_inst;
    }
    if (maxinst == 0) return 0;
    bool staged = maxinst <= STAGE_MAX_INST;
    size_t lds = 0;
    for (int pass = 0; pass < 2; ++pass) {
        lds = 0;
        for (int i = 0; i < n; ++i) lds = lds_bwd(a[i].f, staged) > lds ? lds_bwd(a[i].f, staged) : lds;
        if (lds <= LDS_LIMIT) break;
        staged = false;
    }
    g.staged = staged ? 1 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(maxinst, n), dim3(256), lds, (hipStream_t)stream, g);
    TWOG_CHECK_LAUNCH();
    return 0;
}
