// Fusion-level attention message passing between entities (humans, objects, geometry node).
//
// Reference: the ten *_message methods (vhoi/models.py:1004-1475) + compute_non_relational_message (:1693-1718) +
// compute_attention_weights (:1721-1754) for message_type 'v2', granularity 'v1' (the sender's message does not
// depend on the receiver), aggregation 'att', attention style 'v2'/'v3' (dot / scaled dot product).
// The reference runs this per time step, per receiver, per sender in Python; here ONE workgroup handles ONE instance
// (a (clip, frame) at frame level, a clip at segment level): the entity feature vectors are staged in LDS once, every
// pairwise score is a wave-level dot product (shuffle reduction), the masked softmax (-inf on virtual senders, NaN -> 0
// when every sender is virtual, :1750-1753) is done in place in LDS, and the weighted message sums are written
// lane-contiguously straight into the caller's concatenated entity rows. Sender messages are computed once per
// (sender, relation) by the MFMA GEMM, not once per receiver.
//
// Two launch regimes: throughput (frame level: thousands of instances, 256-thread workgroups streaming from global
// memory at high occupancy) and latency (segment level: one instance per clip inside the time loop; 1024-thread
// workgroups so 4 waves per SIMD interleave the dependent chains, all message / gradient rows staged in LDS up front).
// The rows of one instance are always equally strided, so every row set is resolved ONCE to (base, step): no integer
// division and no descriptor re-read inside the loops (that instruction overhead dominated the first version).
#include "twog_common.h"

namespace {

constexpr int MAX_H = 4, MAX_O = 12, MAX_E = MAX_H + MAX_O;
constexpr int NATT_MAX = MAX_H * MAX_H + 2 * MAX_H * MAX_O + MAX_O * MAX_O;
constexpr int MAXG = 4;
struct FwdGroup { twog_attn_t a[MAXG]; int staged; int columns; };
struct BwdGroup { twog_attn_bwd_t a[MAXG]; int staged; int wave_groups; };

// the n rows (inst*n .. inst*n+n-1) of a twog_rows_t resolved to base + e*step  (host guarantees inner <= 1 or == n)
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

// copies the n x width floats of a row set into LDS (16-byte accesses) and re-points the set at the copy
// (rows are laid out width + 4 floats apart: consecutive rows start 4 LDS banks apart, so lanes that read different rows
// at the same column -- the thread-level partial dot products of the backward kernel -- do not collide)
__device__ __forceinline__ void stage_rowset(RowSet& rs, int n, int width, float*& cursor) {
    if (!rs.on()) return;
    const int w4 = width >> 2, ld = width + 4;
    for (int i = threadIdx.x; i < n * w4; i += blockDim.x) {
        const int r = i / w4, c = (i - r * w4) * 4;
        *reinterpret_cast<float4*>(cursor + r * ld + c) = *reinterpret_cast<const float4*>(rs.row(r) + c);
    }
    rs.base = cursor;
    rs.step = ld;
    cursor += n * ld;
}

// layout of the saved attention weights of one instance
__device__ __forceinline__ int att_off_hh(int, int) { return 0; }
__device__ __forceinline__ int att_off_oh(int H, int) { return H * H; }
__device__ __forceinline__ int att_off_ho(int H, int O) { return H * H + H * O; }
__device__ __forceinline__ int att_off_oo(int H, int O) { return H * H + 2 * H * O; }

// masked softmax of one receiver's scores: every thread of the (relation, receiver) rows runs the SAME code -- the row is
// described by (base, S, excluded sender, use the object mask) -- with its <= MAX_O scores pulled into registers first, so
// the wave that holds these rows neither diverges four ways nor pays a dependent LDS round trip per sender and pass.
// w[0..S) <- softmax over the valid senders, 0 elsewhere (no valid sender -> all zeros: the reference's NaN -> 0).
__device__ __forceinline__ void softmax_row(const float* score, float* w, int S, bool relation_on, int excluded,
                                            bool use_mask, const float* sMask) {
    float sc[MAX_O];
    bool ok[MAX_O];
    float m = -INFINITY;
#pragma unroll
    for (int s = 0; s < MAX_O; ++s) {
        ok[s] = s < S && s != excluded && (!use_mask || sMask[s < S ? s : 0] != 0.f);
        sc[s] = ok[s] ? score[s < S ? s : 0] : -INFINITY;
        m = fmaxf(m, sc[s]);
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < MAX_O; ++s) {
        sc[s] = ok[s] ? expf(sc[s] - m) : 0.f;
        sum += sc[s];
    }
#pragma unroll
    for (int s = 0; s < MAX_O; ++s)
        if (s < S) w[s] = (relation_on && ok[s]) ? sc[s] / sum : 0.f;
}

constexpr int GRAM_PART = 1024;   // floats of LDS scratch for the partial dot products
__device__ __forceinline__ void weights_from_gram(const twog_attn_t& A, const float* sG, float* sW, const float* sMask);

// Gram matrix of the throughput regime (frame level: thousands of instances, 2h-wide features), column-parallel: every
// thread owns feature columns (two floats at a time), loads the E values of a column from global memory ONCE, and keeps
// the E(E-1)/2 pair products in registers -- no staging of the feature rows in LDS, no re-reading them once per pair
// partner (10 rows x 1028 floats staged and 451 KB read back per instance in the row-parallel form). The per-thread
// partials are summed over the 64 lanes with a reduce-scatter butterfly (every exchange step halves the number of live
// values: PP - 1 exchanges per lane instead of 6 per value), the waves' results are added in fixed order through sP.
// N live values per lane, exchange distance M: the lane with bit M set keeps the upper half of its N values and receives
// its partner's contribution to them, the other lane the lower half; with one value left the steps are plain all-reduces
template <int PP, int N, int M>
__device__ __forceinline__ void lane_reduce_scatter(float (&acc)[PP], int lane) {
    if constexpr (M >= 1) {
        if constexpr (N > 1) {
            constexpr int half = N / 2;
            const bool upper = (lane & M) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float send = upper ? acc[i] : acc[i + half];
                const float keep = upper ? acc[i + half] : acc[i];
                acc[i] = keep + __shfl_xor(send, M, 64);
            }
            lane_reduce_scatter<PP, half, M / 2>(acc, lane);
        } else {
            acc[0] += __shfl_xor(acc[0], M, 64);
            lane_reduce_scatter<PP, 1, M / 2>(acc, lane);
        }
    }
}

template <int EMAX, int PP>
__device__ __forceinline__ void gram_columns(const twog_attn_t& A, const RowSet& fh, const RowSet& fo, float* sG, float* sP) {
    constexpr int P = EMAX * (EMAX - 1) / 2;
    static_assert(P <= PP && (PP & (PP - 1)) == 0 && PP <= 64, "pairs padded to a power of two <= 64");
    const int H = A.H, O = A.O, E = H + O, D = A.D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float acc[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) acc[i] = 0.f;
    const float* rows[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) rows[e] = e < H ? fh.row(e) : (e < E ? fo.row(e - H) : fh.row(0));
    for (int c2 = threadIdx.x; c2 < (D >> 1); c2 += blockDim.x) {
        float2 x[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) {
            x[e] = *reinterpret_cast<const float2*>(rows[e] + 2 * c2);
            if (e >= E) x[e] = make_float2(0.f, 0.f);
        }
        int p = 0;
#pragma unroll
        for (int a = 0; a < EMAX; ++a)
#pragma unroll
            for (int b = a + 1; b < EMAX; ++b) {
                acc[p] = fmaf(x[a].x, x[b].x, fmaf(x[a].y, x[b].y, acc[p]));
                ++p;
            }
    }
    // reduce-scatter over the lanes: after the halving steps lane L holds pair (L >> shift), summed over all 64 lanes
    lane_reduce_scatter<PP, PP, 32>(acc, lane);
    constexpr int LANES_PER_VALUE = 64 / PP;
    if ((lane & (LANES_PER_VALUE - 1)) == 0) sP[wave * PP + lane / LANES_PER_VALUE] = acc[0];
    __syncthreads();
    const int t = threadIdx.x;
    if (t < P) {
        int p = t, a = 0;
        while (p >= EMAX - 1 - a) { p -= EMAX - 1 - a; ++a; }
        const int b = a + 1 + p;
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += sP[w * PP + t];   // fixed order: deterministic
        v *= A.scale;
        if (a < E && b < E) { sG[a * E + b] = v; sG[b * E + a] = v; }
    }
    if (t < E) sG[t * E + t] = 0.f;   // never read (a receiver is excluded from its own senders)
    __syncthreads();
}

// pairwise scores + the four masked softmaxes. sF: [E][ldf] features in LDS (ldf = D + 4: consecutive rows start 4 banks
// apart, so lanes that read different rows at the same column never collide); sG: [E][E] scratch; sP: GRAM_PART floats of
// scratch; sW: weights out.
// Gram matrix: the P = E(E+1)/2 pairs x Q column chunks are spread over the threads (thread -> pair p, chunk q); every
// thread accumulates its strided share of one dot product from LDS, the Q partials of a pair are added by one thread.
// (One wave per pair with a 6-step cross-lane reduction each -- the first version -- made this phase a chain of dependent
// LDS / permute latencies: 10 of the 19 us of a segment-level launch, measured with cycle stamps.)
__device__ __forceinline__ void compute_weights(const twog_attn_t& A, const float* sF, int ldf, float* sG, float* sP,
                                                float* sW, const float* sMask) {
    const int H = A.H, O = A.O, E = H + O, D = A.D;
    const int P = E * (E + 1) / 2;
    int Q = (int)blockDim.x / (P > 0 ? P : 1);
    if (Q > GRAM_PART / (P > 0 ? P : 1)) Q = GRAM_PART / (P > 0 ? P : 1);
    if (Q < 1) Q = 1;
    const int t = threadIdx.x;
    if (t < P * Q) {
        int p = t % P;
        const int q = t / P;
        int a = 0;
        while (p >= E - a) { p -= E - a; ++a; }
        const int b = a + p;
        const float4* fa = reinterpret_cast<const float4*>(sF + a * ldf);
        const float4* fb = reinterpret_cast<const float4*>(sF + b * ldf);
        float acc = 0.f;
        for (int d = q; d < (D >> 2); d += Q) {
            const float4 x = fa[d], y = fb[d];
            acc = fmaf(x.x, y.x, acc);
            acc = fmaf(x.y, y.y, acc);
            acc = fmaf(x.z, y.z, acc);
            acc = fmaf(x.w, y.w, acc);
        }
        sP[q * P + (t % P)] = acc;
    }
    __syncthreads();
    if (t < P) {
        int p = t, a = 0;
        while (p >= E - a) { p -= E - a; ++a; }
        const int b = a + p;
        float acc = 0.f;
        for (int q = 0; q < Q; ++q) acc += sP[q * P + t];   // fixed order: deterministic
        acc *= A.scale;
        sG[a * E + b] = acc;
        sG[b * E + a] = acc;
    }
    __syncthreads();
    weights_from_gram(A, sG, sW, sMask);
}

// the four masked softmaxes over rows of the scaled Gram matrix sG [E][E]: one thread per (relation, receiver); ends with
// a barrier
__device__ __forceinline__ void weights_from_gram(const twog_attn_t& A, const float* sG, float* sW, const float* sMask) {
    const int H = A.H, O = A.O, E = H + O;
    const int i = threadIdx.x;
    if (i < 2 * H + 2 * O) {
        const float* sc;
        float* w;
        int S, excl = -1;
        bool on, use_mask = false;
        if (i < H) {                      // hh: receiver human i, senders humans != i
            sc = sG + i * E; w = sW + att_off_hh(H, O) + i * H; S = H; on = A.msg_hh.ptr != nullptr; excl = i;
        } else if (i < 2 * H) {           // oh: receiver human, senders objects (masked)
            const int h = i - H;
            sc = sG + h * E + H; w = sW + att_off_oh(H, O) + h * O; S = O; on = A.msg_oh.ptr != nullptr; use_mask = true;
        } else if (i < 2 * H + O) {       // ho: receiver object, senders humans
            const int k = i - 2 * H;
            sc = sG + (H + k) * E; w = sW + att_off_ho(H, O) + k * H; S = H; on = A.msg_ho.ptr != nullptr;
        } else {                          // oo: receiver object k, senders objects != k (masked)
            const int k = i - 2 * H - O;
            sc = sG + (H + k) * E + H; w = sW + att_off_oo(H, O) + k * O; S = O; on = A.msg_oo.ptr != nullptr;
            excl = k; use_mask = true;
        }
        softmax_row(sc, w, S, on, excl, use_mask, sMask);
    }
    __syncthreads();
}

// weighted sums of one instance from the weights in sW (phase 3 of both forward kernels)
struct FwdRows {
    RowSet m_hh, m_ho, m_oh, m_oo, m_so, m_sh, o_hh, o_oh, o_sh, o_ho, o_so, o_oo;
};
__device__ __forceinline__ void attn_outputs(const twog_attn_t& A, const float* sW, const float* sMask, const FwdRows& R,
                                             bool do01, bool do23) {
    const int H = A.H, O = A.O, hid = A.hidden;
    const RowSet &m_hh = R.m_hh, &m_ho = R.m_ho, &m_oh = R.m_oh, &m_oo = R.m_oo, &m_so = R.m_so, &m_sh = R.m_sh;
    const RowSet &o_hh = R.o_hh, &o_oh = R.o_oh, &o_sh = R.o_sh, &o_ho = R.o_ho, &o_so = R.o_so, &o_oo = R.o_oo;
    // weighted sums: one (group, column) item per thread, lane-contiguous loads/stores. Groups are independent pieces of
    // work of similar size -- 0: messages to humans (hh, oh, sh); 1: human/geometry messages to objects (ho, so);
    // 2, 3: object->object messages for the first / second half of the receivers.
    const int o_half = (O + 1) / 2;
    const bool rmask = A.recv_mask_ho != 0;
    const int idx_lo = do01 ? 0 : 2 * hid, idx_hi = do23 ? 4 * hid : 2 * hid;
    for (int idx = idx_lo + threadIdx.x; idx < idx_hi; idx += blockDim.x) {
        const int grp = idx / hid, j = idx - grp * hid;
        float m[MAX_O];
        if (grp == 0) {
            if (m_hh.on()) {
#pragma unroll
                for (int s = 0; s < MAX_H; ++s) m[s] = s < H ? m_hh.row(s)[j] : 0.f;
                for (int h = 0; h < H; ++h) {
                    const float* w = sW + att_off_hh(H, O) + h * H;
                    float acc = 0.f;
#pragma unroll
                    for (int s = 0; s < MAX_H; ++s)
                        if (s < H) acc = fmaf(w[s], m[s], acc);
                    o_hh.row(h)[j] = acc;
                }
            }
            if (m_oh.on()) {
#pragma unroll
                for (int s = 0; s < MAX_O; ++s) m[s] = s < O ? m_oh.row(s)[j] : 0.f;
                for (int h = 0; h < H; ++h) {
                    const float* w = sW + att_off_oh(H, O) + h * O;
                    float acc = 0.f;
#pragma unroll
                    for (int s = 0; s < MAX_O; ++s)
                        if (s < O) acc = fmaf(w[s], m[s], acc);
                    o_oh.row(h)[j] = acc;
                }
            }
            if (m_sh.on()) {
                const float v = m_sh.row(0)[j];
                for (int h = 0; h < H; ++h) o_sh.row(h)[j] = v;
            }
        } else if (grp == 1) {
            if (m_ho.on()) {
#pragma unroll
                for (int s = 0; s < MAX_H; ++s) m[s] = s < H ? m_ho.row(s)[j] : 0.f;
                for (int k = 0; k < O; ++k) {
                    const float* w = sW + att_off_ho(H, O) + k * H;
                    float acc = 0.f;
#pragma unroll
                    for (int s = 0; s < MAX_H; ++s)
                        if (s < H) acc = fmaf(w[s], m[s], acc);
                    o_ho.row(k)[j] = rmask ? acc * sMask[k] : acc;
                }
            }
            if (m_so.on()) {
                const float v = m_so.row(0)[j];
                for (int k = 0; k < O; ++k) o_so.row(k)[j] = rmask ? v * sMask[k] : v;
            }
        } else if (m_oo.on()) {
            const int k0 = grp == 2 ? 0 : o_half, k1 = grp == 2 ? o_half : O;
#pragma unroll
            for (int s = 0; s < MAX_O; ++s) m[s] = s < O ? m_oo.row(s)[j] : 0.f;
            for (int k = k0; k < k1; ++k) {
                const float* w = sW + att_off_oo(H, O) + k * O;
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < MAX_O; ++s)
                    if (s < O) acc = fmaf(w[s], m[s], acc);
                o_oo.row(k)[j] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void attn_fwd_kernel(const FwdGroup g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const twog_attn_t& A = g.a[blockIdx.y];
    const int inst = blockIdx.x;
    if (inst >= A.n_inst) return;
    const int H = A.H, O = A.O, E = H + O, D = A.D, hid = A.hidden;
    const int ldf = D + 4;     // padded feature rows (see compute_weights)
    float* sG = sm;            // [E][E]
    float* sW = sG + MAX_E * MAX_E;
    float* sMask = sW + NATT_MAX;
    float* sP = sMask + MAX_O + 4;   // [GRAM_PART]
    float* sF = sP + GRAM_PART;      // [E][ldf] (row-parallel Gram only), then the staged message rows
    const int clip = inst / A.inst_per_clip;
    const RowSet fh = rowset(A.feat_h, inst, H), fo = rowset(A.feat_o, inst, O);
    RowSet m_hh = rowset(A.msg_hh, inst, H), m_ho = rowset(A.msg_ho, inst, H);
    RowSet m_oh = rowset(A.msg_oh, inst, O), m_oo = rowset(A.msg_oo, inst, O);
    RowSet m_so = rowset(A.msg_so, inst, 1), m_sh = rowset(A.msg_sh, inst, 1);
    const RowSet o_hh = rowset(A.out_hh, inst, H), o_oh = rowset(A.out_oh, inst, H), o_sh = rowset(A.out_sh, inst, H);
    const RowSet o_ho = rowset(A.out_ho, inst, O), o_so = rowset(A.out_so, inst, O), o_oo = rowset(A.out_oo, inst, O);
    if (threadIdx.x < O) sMask[threadIdx.x] = A.obj_mask ? A.obj_mask[(int64_t)clip * O + threadIdx.x] : 1.f;
    {   // features -> LDS
        const int d4 = D >> 2;
        for (int i = threadIdx.x; i < E * d4; i += blockDim.x) {
            const int e = i / d4, c = (i - e * d4) * 4;
            const float* src = e < H ? fh.row(e) : fo.row(e - H);
            *reinterpret_cast<float4*>(sF + e * ldf + c) = *reinterpret_cast<const float4*>(src + c);
        }
    }
    // latency regime: the instance is shared by gridDim.z = 2 workgroups -- both compute the (cheap) weights, half 0
    // produces the messages to humans and the human / geometry messages to objects, half 1 the object -> object ones
    const int half = blockIdx.z, nhalf = gridDim.z;
    const bool do01 = nhalf == 1 || half == 0, do23 = nhalf == 1 || half == 1;
    if (g.staged) {
        float* cur = sF + E * ldf;
        cur = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(cur) + 15) & ~uintptr_t(15));
        if (do01) {
            stage_rowset(m_hh, H, hid, cur);
            stage_rowset(m_ho, H, hid, cur);
            stage_rowset(m_oh, O, hid, cur);
            stage_rowset(m_so, 1, hid, cur);
            stage_rowset(m_sh, 1, hid, cur);
        }
        if (do23) stage_rowset(m_oo, O, hid, cur);
    }
    __syncthreads();
    compute_weights(A, sF, ldf, sG, sP, sW, sMask);
    const int natt = H * H + 2 * H * O + O * O;
    if (A.att && half == 0)
        for (int i = threadIdx.x; i < natt; i += blockDim.x) A.att[(int64_t)inst * natt + i] = sW[i];
    FwdRows R{m_hh, m_ho, m_oh, m_oo, m_so, m_sh, o_hh, o_oh, o_sh, o_ho, o_so, o_oo};
    attn_outputs(A, sW, sMask, R, do01, do23);
}

// Throughput regime with at most ten entities per instance: the Gram matrix comes column-parallel from global memory
// (gram_columns), no feature rows in LDS -- the workgroup needs 6 KB of LDS instead of 47 KB. 80 VGPRs keep three
// 512-thread workgroups on a CU (6 waves per SIMD).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) void attn_fwd_cols_kernel(const FwdGroup g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const twog_attn_t& A = g.a[blockIdx.y];
    const int inst = blockIdx.x;
    if (inst >= A.n_inst) return;
    const int H = A.H, O = A.O, E = H + O;
    float* sG = sm;
    float* sW = sG + MAX_E * MAX_E;
    float* sMask = sW + NATT_MAX;
    float* sP = sMask + MAX_O + 4;
    const int clip = inst / A.inst_per_clip;
    const RowSet fh = rowset(A.feat_h, inst, H), fo = rowset(A.feat_o, inst, O);
    FwdRows R{rowset(A.msg_hh, inst, H), rowset(A.msg_ho, inst, H), rowset(A.msg_oh, inst, O), rowset(A.msg_oo, inst, O),
              rowset(A.msg_so, inst, 1), rowset(A.msg_sh, inst, 1), rowset(A.out_hh, inst, H), rowset(A.out_oh, inst, H),
              rowset(A.out_sh, inst, H), rowset(A.out_ho, inst, O), rowset(A.out_so, inst, O), rowset(A.out_oo, inst, O)};
    if (threadIdx.x < O) sMask[threadIdx.x] = A.obj_mask ? A.obj_mask[(int64_t)clip * O + threadIdx.x] : 1.f;
    if (E <= 6) gram_columns<6, 16>(A, fh, fo, sG, sP);
    else gram_columns<10, 64>(A, fh, fo, sG, sP);
    weights_from_gram(A, sG, sW, sMask);
    const int natt = H * H + 2 * H * O + O * O;
    if (A.att)
        for (int i = threadIdx.x; i < natt; i += blockDim.x) A.att[(int64_t)inst * natt + i] = sW[i];
    attn_outputs(A, sW, sMask, R, true, true);
}

// wave-level dot product of two rows of length n (n % 4 == 0, 16-byte aligned)
__device__ __forceinline__ float wave_dot(const float* a, const float* b, int n, int lane) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float acc = 0.f;
    for (int j = lane; j < (n >> 2); j += 64) {
        const float4 x = a4[j], y = b4[j];
        acc = fmaf(x.x, y.x, acc);
        acc = fmaf(x.y, y.y, acc);
        acc = fmaf(x.z, y.z, acc);
        acc = fmaf(x.w, y.w, acc);
    }
    return wave_sum(acc);
}

// dL/dw of one relation, column-parallel (the throughput regime of the backward kernel): the thread's columns of the R
// gradient rows and the S sender-message rows are loaded once, the R x S products stay in registers, the lanes add
// them with the reduce-scatter butterfly, the waves' sums meet in sP [wave][PP]; the caller adds them in fixed order.
// (tid, nthreads: this thread's index in and the size of the thread group that shares the relation -- the workgroup, or
// one of its wave groups in the latency regime)
// gw != nullptr: the same pass also produces the gradient of the SENDER messages of this relation from the values it has
// in registers anyway -- dmsg[s][c] = sum_r w[r][s] (mask_r) dout[r][c], times ReLU'(msg[s][c]) -- so the backward kernel
// reads every dout / message row once instead of twice (the separate message-gradient phase re-read 58 rows of 2 KB per
// instance at the BASELINE shape: 30 % of the kernel's traffic). Needs ALL receivers of the relation in this call.
// gw: the relation's saved weights w[r * ldw + s] for this call's senders (LDS); gmask: receiver mask or nullptr.
template <int RMAX, int SMAX, int PP>
__device__ __forceinline__ void dw_columns(const RowSet& dr, const RowSet& mr, int R, int S, int hid, float* sP,
                                           int tid, int nthreads, const float* gw = nullptr, int ldw = 0,
                                           const RowSet* gr = nullptr, const float* gmask = nullptr, bool relu = false) {
    static_assert(RMAX * SMAX <= PP && PP <= 64, "one slot per (receiver, sender)");
    const int lane = tid & 63, wave = tid >> 6;
    float acc[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) acc[i] = 0.f;
    for (int c2 = tid; c2 < (hid >> 1); c2 += nthreads) {
        float2 d[RMAX], m[SMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) d[r] = r < R ? *reinterpret_cast<const float2*>(dr.row(r) + 2 * c2) : make_float2(0.f, 0.f);
#pragma unroll
        for (int s = 0; s < SMAX; ++s) m[s] = s < S ? *reinterpret_cast<const float2*>(mr.row(s) + 2 * c2) : make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
#pragma unroll
            for (int s = 0; s < SMAX; ++s)
                acc[r * SMAX + s] = fmaf(d[r].x, m[s].x, fmaf(d[r].y, m[s].y, acc[r * SMAX + s]));
        if (gw) {
            if (gmask) {
#pragma unroll
                for (int r = 0; r < RMAX; ++r)
                    if (r < R) { const float k = gmask[r]; d[r].x *= k; d[r].y *= k; }
            }
#pragma unroll
            for (int s = 0; s < SMAX; ++s) {
                if (s < S) {
                    float gx = 0.f, gy = 0.f;
#pragma unroll
                    for (int r = 0; r < RMAX; ++r)
                        if (r < R) { const float w = gw[r * ldw + s]; gx = fmaf(w, d[r].x, gx); gy = fmaf(w, d[r].y, gy); }
                    if (relu) { if (!(m[s].x > 0.f)) gx = 0.f; if (!(m[s].y > 0.f)) gy = 0.f; }
                    *reinterpret_cast<float2*>(gr->row(s) + 2 * c2) = make_float2(gx, gy);
                }
            }
        }
    }
    lane_reduce_scatter<PP, PP, 32>(acc, lane);
    constexpr int LANES_PER_VALUE = 64 / PP;
    if ((lane & (LANES_PER_VALUE - 1)) == 0) sP[wave * PP + lane / LANES_PER_VALUE] = acc[0];
}

template <bool COLS>
__device__ __forceinline__ void attn_bwd_body(const BwdGroup& g) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const twog_attn_bwd_t& B = g.a[blockIdx.y];
    const twog_attn_t& A = B.f;
    const int inst = blockIdx.x;
    if (inst >= A.n_inst) return;
    const int H = A.H, O = A.O, E = H + O, D = A.D, hid = A.hidden;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int natt = H * H + 2 * H * O + O * O;
    float* sW = sm;               // saved weights [natt]
    float* sdW = sW + NATT_MAX;   // dL/dw then dscore [natt]
    float* sC = sdW + NATT_MAX;   // [E][E] coefficients of dF
    float* sMask = sC + MAX_E * MAX_E;
    const int clip = inst / A.inst_per_clip;
    const RowSet fh = rowset(A.feat_h, inst, H), fo = rowset(A.feat_o, inst, O);
    RowSet m_hh = rowset(A.msg_hh, inst, H), m_ho = rowset(A.msg_ho, inst, H);
    RowSet m_oh = rowset(A.msg_oh, inst, O), m_oo = rowset(A.msg_oo, inst, O);
    RowSet m_so = rowset(A.msg_so, inst, 1), m_sh = rowset(A.msg_sh, inst, 1);
    RowSet d_hh = rowset(B.dout_hh, inst, H), d_oh = rowset(B.dout_oh, inst, H), d_sh = rowset(B.dout_sh, inst, H);
    RowSet d_ho = rowset(B.dout_ho, inst, O), d_so = rowset(B.dout_so, inst, O), d_oo = rowset(B.dout_oo, inst, O);
    const RowSet g_hh = rowset(B.dmsg_hh, inst, H), g_ho = rowset(B.dmsg_ho, inst, H);
    const RowSet g_oh = rowset(B.dmsg_oh, inst, O), g_oo = rowset(B.dmsg_oo, inst, O);
    const RowSet g_so = rowset(B.dmsg_so, inst, 1), g_sh = rowset(B.dmsg_sh, inst, 1);
    const RowSet df_h = rowset(B.dfeat_h, inst, H), df_o = rowset(B.dfeat_o, inst, O);
    const bool rmask = A.recv_mask_ho != 0, relu_mask = B.relu_mask_dmsg != 0;
    for (int i = threadIdx.x; i < natt; i += blockDim.x) sW[i] = A.att[(int64_t)inst * natt + i];
    if (threadIdx.x < O) sMask[threadIdx.x] = A.obj_mask ? A.obj_mask[(int64_t)clip * O + threadIdx.x] : 1.f;
    float* sP = sMask + MAX_O + 4;   // [GRAM_PART] partial dot products (latency regime)
    // latency regime (gridDim.z = 2): half 0 produces the sender-message gradients, half 1 the feature gradients
    const int half = blockIdx.z, nhalf = gridDim.z;
    const bool do_feat = nhalf == 1 || half == 1, do_msg = nhalf == 1 || half == 0;
    // Half 1 with at most 2 humans and 8 objects computes dL/dw column-parallel straight from global memory, four wave
    // groups of 256 threads taking the relations side by side: it stages nothing (the 12 row sets are ~100 KB per
    // instance: 7 of the 20 us of this launch, cycle stamps) and its critical path loses the LDS round trip.
    const bool wg_dw = g.staged && nhalf == 2 && half == 1 && blockDim.x == 1024 && H <= 2 && O <= 8 && (hid & 1) == 0 &&
                       g.wave_groups != 0;
    if (g.staged && !wg_dw) {
        float* cur = sP + GRAM_PART;
        cur = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(cur) + 15) & ~uintptr_t(15));
        stage_rowset(m_hh, H, hid, cur);
        stage_rowset(m_ho, H, hid, cur);
        stage_rowset(m_oh, O, hid, cur);
        stage_rowset(m_oo, O, hid, cur);
        stage_rowset(m_so, 1, hid, cur);
        stage_rowset(m_sh, 1, hid, cur);
        if (m_hh.on()) stage_rowset(d_hh, H, hid, cur);
        if (m_oh.on()) stage_rowset(d_oh, H, hid, cur);
        if (m_sh.on()) stage_rowset(d_sh, H, hid, cur);
        if (m_ho.on()) stage_rowset(d_ho, O, hid, cur);
        if (m_so.on()) stage_rowset(d_so, O, hid, cur);
        if (m_oo.on()) stage_rowset(d_oo, O, hid, cur);
    }
    __syncthreads();
    // dL/dw[r][s] = recv_mask_r * <dout[r], msg[s]>
    // latency regime (rows staged in LDS): thread -> (pair p, column chunk q) partial dot products, then one ordered add
    // per pair -- no chain of per-pair cross-lane reductions (see compute_weights)
    auto pair_rows = [&](int p, const float*& dr, const float*& mr, float& rs) {
        dr = mr = nullptr;
        rs = 1.f;
        if (p < H * H) {
            const int r = p / H, s_ = p - r * H;
            if (m_hh.on() && r != s_) { dr = d_hh.row(r); mr = m_hh.row(s_); }
        } else if (p < H * H + H * O) {
            const int q = p - H * H, r = q / O, s_ = q - r * O;
            if (m_oh.on()) { dr = d_oh.row(r); mr = m_oh.row(s_); }
        } else if (p < H * H + 2 * H * O) {
            const int q = p - H * H - H * O, r = q / H, s_ = q - r * H;
            if (m_ho.on()) { dr = d_ho.row(r); mr = m_ho.row(s_); rs = rmask ? sMask[r] : 1.f; }
        } else {
            const int q = p - H * H - 2 * H * O, r = q / O, s_ = q - r * O;
            if (m_oo.on() && r != s_) { dr = d_oo.row(r); mr = m_oo.row(s_); }
        }
    };
    if (wg_dw) {
        // wave group 0 / 1: object->object, receivers 0..3 / 4..7; group 2: human->object then human->human; group 3:
        // object->human. Partial sums of a group's 4 waves at sP [group][wave][32], slot (r, s) at r * SMAX + s.
        const int grp = threadIdx.x >> 8, tg = threadIdx.x & 255;
        float* sPg = sP + grp * 4 * 32;
        auto shifted = [](RowSet rs, int r0) { if (rs.on()) rs.base += (int64_t)r0 * rs.step; return rs; };
        if (grp == 0) { if (m_oo.on()) dw_columns<4, 8, 32>(d_oo, m_oo, min(O, 4), O, hid, sPg, tg, 256); }
        else if (grp == 1) { if (m_oo.on() && O > 4) dw_columns<4, 8, 32>(shifted(d_oo, 4), m_oo, O - 4, O, hid, sPg, tg, 256); }
        else if (grp == 2) { if (m_ho.on()) dw_columns<8, 2, 16>(d_ho, m_ho, O, H, hid, sPg, tg, 256); }
        else { if (m_oh.on()) dw_columns<2, 8, 16>(d_oh, m_oh, H, O, hid, sPg, tg, 256); }
        if (grp == 2 && m_hh.on()) dw_columns<2, 2, 4>(d_hh, m_hh, H, H, hid, sPg + 4 * 16, tg, 256);
        __syncthreads();
        for (int p = threadIdx.x; p < natt; p += blockDim.x) {
            float v = 0.f;
            if (p < H * H) {
                const int r = p / H, s_ = p - r * H;
                if (m_hh.on() && r != s_)
                    for (int w = 0; w < 4; ++w) v += sP[2 * 128 + 64 + w * 4 + r * 2 + s_];
            } else if (p < H * H + H * O) {
                const int q = p - H * H, r = q / O, s_ = q - r * O;
                if (m_oh.on())
                    for (int w = 0; w < 4; ++w) v += sP[3 * 128 + w * 16 + r * 8 + s_];
            } else if (p < H * H + 2 * H * O) {
                const int q = p - H * H - H * O, r = q / H, s_ = q - r * H;
                if (m_ho.on()) {
                    for (int w = 0; w < 4; ++w) v += sP[2 * 128 + w * 16 + r * 2 + s_];
                    if (rmask) v *= sMask[r];
                }
            } else {
                const int q = p - H * H - 2 * H * O, r = q / O, s_ = q - r * O;
                if (m_oo.on() && r != s_)
                    for (int w = 0; w < 4; ++w) v += sP[(r >> 2) * 128 + w * 32 + (r & 3) * 8 + s_];
            }
            sdW[p] = B.dw_extra ? v + B.dw_extra[(int64_t)inst * natt + p] : v;
        }
    } else if (do_feat && g.staged) {
        int Q = (int)blockDim.x / natt;
        if (Q > GRAM_PART / natt) Q = GRAM_PART / natt;
        if (Q < 1) Q = 1;
        const int t = threadIdx.x;
        if (t < natt * Q) {
            const int p = t % natt, q = t / natt;
            const float *dr, *mr;
            float rs;
            pair_rows(p, dr, mr, rs);
            float acc = 0.f;
            if (dr) {
                const float4* a4 = reinterpret_cast<const float4*>(dr);
                const float4* b4 = reinterpret_cast<const float4*>(mr);
                for (int j = q; j < (hid >> 2); j += Q) {
                    const float4 x = a4[j], y = b4[j];
                    acc = fmaf(x.x, y.x, acc);
                    acc = fmaf(x.y, y.y, acc);
                    acc = fmaf(x.z, y.z, acc);
                    acc = fmaf(x.w, y.w, acc);
                }
            }
            sP[q * natt + p] = acc * rs;
        }
        __syncthreads();
        if (t < natt) {
            float v = 0.f;
            for (int q = 0; q < Q; ++q) v += sP[q * natt + t];
            sdW[t] = B.dw_extra ? v + B.dw_extra[(int64_t)inst * natt + t] : v;
        }
    } else if (COLS && do_feat) {
        // (at most 2 humans and 8 objects, checked by the host) one relation after the other; slot (r, s) of a relation
        // sits at r * SMAX + s of its butterfly, the waves' partial sums at sP [wave][64]
        // receivers [r0, r0 + R) of a relation; skip_diag: receiver r0 + r is excluded from its own senders
        // receivers [0, R) (all of them), senders [s0, s0 + S) of a relation with Sfull senders; skip_diag: receiver r is
        // excluded from its own senders (its saved weight is an exact zero). The pass also writes the relation's
        // sender-message gradients (see dw_columns).
        auto relation = [&](auto tag, const RowSet& dr, RowSet mr, RowSet gr, int R, int s0, int S, int Sfull, int off,
                            bool skip_diag, bool recv_masked) {
            constexpr int RMAX = decltype(tag)::R, SMAX = decltype(tag)::S, PP = decltype(tag)::PP;
            if (S <= 0 || R <= 0) return;   // uniform
            if (mr.on()) mr.base += (int64_t)s0 * mr.step;
            if (gr.on()) gr.base += (int64_t)s0 * gr.step;
            if (mr.on())
                dw_columns<RMAX, SMAX, PP>(dr, mr, R, S, hid, sP, (int)threadIdx.x, (int)blockDim.x,
                                           gr.on() ? sW + off + s0 : nullptr, Sfull, &gr, recv_masked ? sMask : nullptr, relu_mask);
            __syncthreads();
            for (int t = threadIdx.x; t < R * S; t += blockDim.x) {
                const int r = t / S, s_ = t - r * S;
                float v = 0.f;
                if (mr.on() && !(skip_diag && r == s0 + s_)) {
                    for (int w = 0; w < nw; ++w) v += sP[w * PP + r * SMAX + s_];   // fixed order: deterministic
                    if (recv_masked) v *= sMask[r];
                }
                const int slot = off + r * Sfull + s0 + s_;
                sdW[slot] = B.dw_extra ? v + B.dw_extra[(int64_t)inst * natt + slot] : v;
            }
            __syncthreads();
        };
        struct T22 { enum { R = 2, S = 2, PP = 4 }; };
        struct T28 { enum { R = 2, S = 8, PP = 16 }; };
        struct T82 { enum { R = 8, S = 2, PP = 16 }; };
        struct T84 { enum { R = 8, S = 4, PP = 32 }; };
        relation(T22{}, d_hh, m_hh, g_hh, H, 0, H, H, att_off_hh(H, O), true, false);
        relation(T28{}, d_oh, m_oh, g_oh, H, 0, O, O, att_off_oh(H, O), false, false);
        relation(T82{}, d_ho, m_ho, g_ho, O, 0, H, H, att_off_ho(H, O), false, rmask);
        relation(T84{}, d_oo, m_oo, g_oo, O, 0, min(O, 4), O, att_off_oo(H, O), true, false);   // two halves of the SENDERS:
        relation(T84{}, d_oo, m_oo, g_oo, O, 4, O - 4, O, att_off_oo(H, O), true, false);        // 32 products in registers
    } else if (do_feat)
    for (int p = wv; p < natt; p += nw) {
        float v = 0.f;
        if (p < H * H) {
            const int r = p / H, s = p - r * H;
            if (m_hh.on() && r != s) v = wave_dot(d_hh.row(r), m_hh.row(s), hid, lane);
        } else if (p < H * H + H * O) {
            const int q = p - H * H, r = q / O, s = q - r * O;
            if (m_oh.on()) v = wave_dot(d_oh.row(r), m_oh.row(s), hid, lane);
        } else if (p < H * H + 2 * H * O) {
            const int q = p - H * H - H * O, r = q / H, s = q - r * H;
            if (m_ho.on()) v = (rmask ? sMask[r] : 1.f) * wave_dot(d_ho.row(r), m_ho.row(s), hid, lane);
        } else {
            const int q = p - H * H - 2 * H * O, r = q / O, s = q - r * O;
            if (m_oo.on() && r != s) v = wave_dot(d_oo.row(r), m_oo.row(s), hid, lane);
        }
        if (lane == 0) sdW[p] = B.dw_extra ? v + B.dw_extra[(int64_t)inst * natt + p] : v;
    }
    __syncthreads();
    // softmax backward per receiver: dscore = w * (dw - sum_s w dw) * scale
    if (do_feat && threadIdx.x < 2 * H + 2 * O) {
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
    // coefficient of F[b] in dF[a]: every score <F_r, F_s> sends its dscore to both its receiver and its sender.
    // Fixed order of the (at most 4) terms per entry -> deterministic.
    if (do_feat)
    for (int i = threadIdx.x; i < E * E; i += blockDim.x) {
        const int a = i / E, b = i - a * E;
        float v;
        if (a < H && b < H) v = sdW[att_off_hh(H, O) + a * H + b] + sdW[att_off_hh(H, O) + b * H + a];
        else if (a < H) v = sdW[att_off_oh(H, O) + a * O + (b - H)] + sdW[att_off_ho(H, O) + (b - H) * H + a];
        else if (b < H) v = sdW[att_off_ho(H, O) + (a - H) * H + b] + sdW[att_off_oh(H, O) + b * O + (a - H)];
        else v = sdW[att_off_oo(H, O) + (a - H) * O + (b - H)] + sdW[att_off_oo(H, O) + (b - H) * O + (a - H)];
        sC[a * MAX_E + b] = v;
    }
    __syncthreads();
    // gradient wrt sender messages: dmsg[s] = sum_r w[r][s] * recv_mask_r * dout[r]  (optionally times ReLU'(msg)),
    // one (group, column) item per thread; groups: 0: hh, oh, sh; 1: ho, so; 2, 3: oo senders first / second half
    const int o_half = (O + 1) / 2;
    const bool ent_done = COLS && do_feat;   // (column regime: the entity relations' message gradients are already written)
    if (do_msg)
    for (int idx = threadIdx.x; idx < (ent_done ? 2 : 4) * hid; idx += blockDim.x) {
        const int grp = idx / hid, j = idx - grp * hid;
        float gr[MAX_O];
        if (grp == 0) {
            if (m_hh.on() && !ent_done) {
#pragma unroll
                for (int r = 0; r < MAX_H; ++r) gr[r] = r < H ? d_hh.row(r)[j] : 0.f;
                for (int s = 0; s < H; ++s) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 0; r < MAX_H; ++r)
                        if (r < H) acc = fmaf(sW[att_off_hh(H, O) + r * H + s], gr[r], acc);
                    if (relu_mask && !(m_hh.row(s)[j] > 0.f)) acc = 0.f;
                    g_hh.row(s)[j] = acc;
                }
            }
            if (m_oh.on() && !ent_done) {
#pragma unroll
                for (int r = 0; r < MAX_H; ++r) gr[r] = r < H ? d_oh.row(r)[j] : 0.f;
                for (int s = 0; s < O; ++s) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 0; r < MAX_H; ++r)
                        if (r < H) acc = fmaf(sW[att_off_oh(H, O) + r * O + s], gr[r], acc);
                    if (relu_mask && !(m_oh.row(s)[j] > 0.f)) acc = 0.f;
                    g_oh.row(s)[j] = acc;
                }
            }
            if (m_sh.on()) {
                float acc = 0.f;
                for (int r = 0; r < H; ++r) acc += d_sh.row(r)[j];
                if (relu_mask && !(m_sh.row(0)[j] > 0.f)) acc = 0.f;
                g_sh.row(0)[j] = acc;
            }
        } else if (grp == 1) {
            if (m_ho.on() && !ent_done) {
#pragma unroll
                for (int r = 0; r < MAX_O; ++r) gr[r] = r < O ? (rmask ? sMask[r] : 1.f) * d_ho.row(r)[j] : 0.f;
                for (int s = 0; s < H; ++s) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 0; r < MAX_O; ++r)
                        if (r < O) acc = fmaf(sW[att_off_ho(H, O) + r * H + s], gr[r], acc);
                    if (relu_mask && !(m_ho.row(s)[j] > 0.f)) acc = 0.f;
                    g_ho.row(s)[j] = acc;
                }
            }
            if (m_so.on()) {
                float acc = 0.f;
                for (int r = 0; r < O; ++r) acc = fmaf(rmask ? sMask[r] : 1.f, d_so.row(r)[j], acc);
                if (relu_mask && !(m_so.row(0)[j] > 0.f)) acc = 0.f;
                g_so.row(0)[j] = acc;
            }
        } else if (m_oo.on()) {
            const int s0 = grp == 2 ? 0 : o_half, s1 = grp == 2 ? o_half : O;
#pragma unroll
            for (int r = 0; r < MAX_O; ++r) gr[r] = r < O ? d_oo.row(r)[j] : 0.f;
            for (int s = s0; s < s1; ++s) {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < MAX_O; ++r)
                    if (r < O) acc = fmaf(sW[att_off_oo(H, O) + r * O + s], gr[r], acc);
                if (relu_mask && !(m_oo.row(s)[j] > 0.f)) acc = 0.f;
                g_oo.row(s)[j] = acc;
            }
        }
    }
    // gradient wrt the features: dF[a] = sum_b C[a][b] F[b]; F is read straight from global memory (one burst of E
    // independent lane-contiguous loads per column), so the backward kernel keeps no LDS copy of the features.
    // Two (group, column) items per column: first / second half of the entities.
    const int e_half = (E + 1) / 2;
    const bool accum = B.dfeat_accumulate != 0;
    if (do_feat)
    for (int idx = threadIdx.x; idx < 2 * D; idx += blockDim.x) {
        const int grp = idx / D, d = idx - grp * D;
        float f[MAX_E];
#pragma unroll
        for (int b = 0; b < MAX_E; ++b) f[b] = b < E ? (b < H ? fh.row(b) : fo.row(b - H))[d] : 0.f;
        const int a0 = grp == 0 ? 0 : e_half, a1 = grp == 0 ? e_half : E;
        // the previous gradient values of all of this item's rows are requested up front: written as one
        // read-modify-write after the other, every load would wait for the store before it (the compiler cannot rule out
        // that the rows alias) -- five dependent round trips instead of one
        float prev[MAX_E / 2];
#pragma unroll
        for (int i = 0; i < MAX_E / 2; ++i) {
            const int a = a0 + i;
            prev[i] = (accum && a < a1) ? (a < H ? df_h.row(a) : df_o.row(a - H))[d] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MAX_E / 2; ++i) {
            const int a = a0 + i;
            if (a < a1) {
                const float* c = sC + a * MAX_E;
                float acc = prev[i];
#pragma unroll
                for (int b = 0; b < MAX_E; ++b)
                    if (b < E) acc = fmaf(c[b], f[b], acc);
                (a < H ? df_h.row(a) : df_o.row(a - H))[d] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void attn_bwd_kernel(const BwdGroup g) { attn_bwd_body<false>(g); }
// throughput regime, at most 2 humans and 8 objects: dL/dw column-parallel (dw_columns)
__global__ __launch_bounds__(256) void attn_bwd_cols_kernel(const BwdGroup g) { attn_bwd_body<true>(g); }

constexpr int STAGE_MAX_INST = 1024;  // calls with at most this many instances use the LDS-staged (latency) path
inline size_t n_msg_rows(const twog_attn_t& a) {
    return (size_t)(a.msg_hh.ptr ? a.H : 0) + (a.msg_ho.ptr ? a.H : 0) + (a.msg_oh.ptr ? a.O : 0) +
           (a.msg_oo.ptr ? a.O : 0) + (a.msg_so.ptr ? 1 : 0) + (a.msg_sh.ptr ? 1 : 0);
}
inline size_t n_dout_rows(const twog_attn_t& a) {
    return (size_t)(a.msg_hh.ptr ? a.H : 0) + (a.msg_oh.ptr ? a.H : 0) + (a.msg_sh.ptr ? a.H : 0) +
           (a.msg_ho.ptr ? a.O : 0) + (a.msg_so.ptr ? a.O : 0) + (a.msg_oo.ptr ? a.O : 0);
}
inline size_t lds_fwd(const twog_attn_t& a, bool staged, bool columns = false) {
    size_t f = (columns ? 0 : (size_t)(a.H + a.O) * (a.D + 4)) + MAX_E * MAX_E + NATT_MAX + MAX_O + 8 + GRAM_PART;
    if (staged) f += n_msg_rows(a) * (a.hidden + 4);
    return sizeof(float) * f;
}
inline size_t lds_bwd(const twog_attn_t& a, bool staged) {
    size_t f = 2 * (size_t)NATT_MAX + MAX_E * MAX_E + MAX_O + 8 + GRAM_PART;
    if (staged) f += (n_msg_rows(a) + n_dout_rows(a)) * (a.hidden + 4);
    return sizeof(float) * f;
}
constexpr size_t LDS_LIMIT = 160 * 1024;

// the rows of one instance must be equally strided: plain rows (inner <= 1) or exactly one outer group per instance
inline bool rows_ok(const twog_rows_t& m, int n) { return !m.ptr || m.inner <= 1 || m.inner == n; }
inline bool desc_ok(const twog_attn_t& a) {
    if (a.H > MAX_H || a.O > MAX_O || a.H < 0 || a.O < 0 || (a.D & 3) || (a.hidden & 3)) return false;
    return rows_ok(a.feat_h, a.H) && rows_ok(a.feat_o, a.O) && rows_ok(a.msg_hh, a.H) && rows_ok(a.msg_ho, a.H) &&
           rows_ok(a.msg_oh, a.O) && rows_ok(a.msg_oo, a.O) && rows_ok(a.msg_so, 1) && rows_ok(a.msg_sh, 1) &&
           rows_ok(a.out_hh, a.H) && rows_ok(a.out_oh, a.H) && rows_ok(a.out_sh, a.H) && rows_ok(a.out_ho, a.O) &&
           rows_ok(a.out_so, a.O) && rows_ok(a.out_oo, a.O);
}
inline bool bdesc_ok(const twog_attn_bwd_t& b) {
    const twog_attn_t& a = b.f;
    return rows_ok(b.dout_hh, a.H) && rows_ok(b.dout_oh, a.H) && rows_ok(b.dout_sh, a.H) && rows_ok(b.dout_ho, a.O) &&
           rows_ok(b.dout_so, a.O) && rows_ok(b.dout_oo, a.O) && rows_ok(b.dmsg_hh, a.H) && rows_ok(b.dmsg_ho, a.H) &&
           rows_ok(b.dmsg_oh, a.O) && rows_ok(b.dmsg_oo, a.O) && rows_ok(b.dmsg_so, 1) && rows_ok(b.dmsg_sh, 1) &&
           rows_ok(b.dfeat_h, a.H) && rows_ok(b.dfeat_o, a.O);
}

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
        if (!desc_ok(a[i])) return -2;
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
    if (lds > LDS_LIMIT) return -3;
    g.staged = staged ? 1 : 0;
    // throughput regime with at most ten entities per instance: column-parallel Gram, no feature rows in LDS
    static const int cols_on = getenv("TWOG_ATTN_COLUMNS") ? atoi(getenv("TWOG_ATTN_COLUMNS")) : 1;
    bool columns = !staged && cols_on;
    for (int i = 0; i < n; ++i) columns = columns && (a[i].H + a[i].O) <= 10 && (a[i].D & 1) == 0;
    g.columns = columns ? 1 : 0;
    if (columns) {
        lds = 0;
        for (int i = 0; i < n; ++i) lds = lds_fwd(a[i], false, true) > lds ? lds_fwd(a[i], false, true) : lds;
    }
    static std::atomic<uint32_t> lds_attr_done{0};
    twog_allow_dynamic_lds(attn_fwd_kernel, (int)LDS_LIMIT, lds_attr_done);
    // throughput regime: 512 threads (the LDS feature tile limits a CU to 3 workgroups: 24 waves instead of 12)
    if (columns) {
        // 256 threads: the butterfly costs the same per wave whatever the workgroup size, and six small workgroups per
        // CU overlap their phases better than three large ones (measured at the BASELINE shape: 0.262 ms with 512
        // threads, 0.231 with 256, 0.250 with 128; the row-parallel kernel: 0.283)
        static const int cols_threads = getenv("TWOG_ATTN_COLS_THREADS") ? atoi(getenv("TWOG_ATTN_COLS_THREADS")) : 256;
        hipLaunchKernelGGL(attn_fwd_cols_kernel, dim3(maxinst, n, 1), dim3(cols_threads), lds, (hipStream_t)stream, g);
        TWOG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(maxinst, n, staged ? 2 : 1), dim3(staged ? 1024 : 512), lds, (hipStream_t)stream, g);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_attn_bwd(const twog_attn_bwd_t* a, int n, void* stream) {
    if (n > MAXG) return -1;
    BwdGroup g;
    int maxinst = 0;
    for (int i = 0; i < n; ++i) {
        g.a[i] = a[i];
        if (!desc_ok(a[i].f) || !bdesc_ok(a[i])) return -2;
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
    if (lds > LDS_LIMIT) return -3;
    g.staged = staged ? 1 : 0;
    static const int wg_on = getenv("TWOG_ATTN_WAVE_GROUPS") ? atoi(getenv("TWOG_ATTN_WAVE_GROUPS")) : 1;
    g.wave_groups = wg_on;
    static std::atomic<uint32_t> lds_attr_done{0};
    twog_allow_dynamic_lds(attn_bwd_kernel, (int)LDS_LIMIT, lds_attr_done);
    static const int cols_on = getenv("TWOG_ATTN_COLUMNS") ? atoi(getenv("TWOG_ATTN_COLUMNS")) : 1;
    bool columns = !staged && cols_on;
    for (int i = 0; i < n; ++i) columns = columns && a[i].f.H <= 2 && a[i].f.O <= 8 && (a[i].f.hidden & 1) == 0;
    if (columns) {
        hipLaunchKernelGGL(attn_bwd_cols_kernel, dim3(maxinst, n, 1), dim3(256), lds, (hipStream_t)stream, g);
        TWOG_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(maxinst, n, staged ? 2 : 1), dim3(staged ? 1024 : 256), lds, (hipStream_t)stream, g);
    TWOG_CHECK_LAUNCH();
    return 0;
}
