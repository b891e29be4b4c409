// Geometric-level GCN kernels (reference pyrutils/torch/models_gcn.py:6-100, called from vhoi/models.py:640-645).
//
// Geo_gcn.forward is executed as
//   bn_stats/bn_finalize  : BatchNorm1d(4N) statistics folded into a per-channel scale/shift (models_gcn.py:43-50)
//   gcn_embed1_fwd        : x^ = a*x+b, e1 = relu(W1 x^ + b1)            (models_gcn.py:57-59; K = 4, VALU)
//   twog_gemm_f32         : X = relu(e1 W2^T + b2), [Q|K] = X [Ws1;Ws2]^T + b   (MFMA; models_gcn.py:60-63, :95-96)
//   gcn_attn_fwd          : per frame S = softmax_j(Q_i . K_j), Z = S X   (models_gcn.py:97-100, :33-34)
//   twog_gemm_f32         : Y = Z W written straight into the (bs,128,N,T) layout, T fastest (models_gcn.py:35-36)
// Node features of one frame (Q, K: N x 128, X: N x 64) are staged in LDS with padded rows so every ds_read_b128 is
// bank-conflict free; the N x N adjacency row lives on the lanes of one wave and its softmax is done with wave
// shuffles; rows of the adjacency are written with lane-contiguous (coalesced) stores.
// The geometry input is read in place from x_human[b, t, 0, 2048:] (frame stride = H * F_h floats), no split copy.
#include "twog_common.h"

namespace {

constexpr int MAX_NODES = 64;

// channel of BatchNorm1d for node n, feature c: c*N + n (models_gcn.py:47); memory position in a frame: n*4 + c
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* x, int64_t fstride, int n_frames, int N,
                                                       double* partials) {
    const int pos = threadIdx.x;  // n*4 + c
    const int nch = 4 * N;
    const int per = (n_frames + gridDim.x - 1) / gridDim.x;
    const int f0 = blockIdx.x * per, f1 = min(n_frames, f0 + per);
    if (pos >= nch) return;
    // four frames in flight (the loop is bound by the latency of its loads, not by the two fp64 adds); the partial sums
    // are combined in a fixed order: bit-reproducible
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    int f = f0;
    for (; f + 4 <= f1; f += 4) {
        const double v0 = (double)x[(int64_t)f * fstride + pos], v1 = (double)x[(int64_t)(f + 1) * fstride + pos];
        const double v2 = (double)x[(int64_t)(f + 2) * fstride + pos], v3 = (double)x[(int64_t)(f + 3) * fstride + pos];
        s0 += v0; q0 += v0 * v0;
        s1 += v1; q1 += v1 * v1;
        s2 += v2; q2 += v2 * v2;
        s3 += v3; q3 += v3 * v3;
    }
    for (; f < f1; ++f) {
        const double v = (double)x[(int64_t)f * fstride + pos];
        s0 += v; q0 += v * v;
    }
    const double s = (s0 + s1) + (s2 + s3), q = (q0 + q1) + (q2 + q3);
    const int ch = (pos & 3) * N + (pos >> 2);
    partials[((int64_t)blockIdx.x * 2 + 0) * nch + ch] = s;
    partials[((int64_t)blockIdx.x * 2 + 1) * nch + ch] = q;
}

// Workgroup 0 (1024 threads): thread = (channel, one of 4 partial lanes); lane p adds the blocks p, p + 4, ... of its
// channel, the four sums meet in LDS in fixed order. Workgroups 1 .. 65 (md_out != NULL) fold the two similarity
// projections of compute_similarity (models_gcn.py:95-100) for the fused forward kernel (geo_fused.hip):
// md_out[n][k] = sum_o Wk[o][n] Wq[o][k] (n, k < 64), md_out[64][n] = sum_o Wk[o][n] bq[o]  (theta_i . phi_j = x_i^T M x_j
// + d . x_j + terms constant in j) -- 0.5 MFLOP, not worth a launch of its own.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* partials, int n_blocks, int n_frames, int nch,
                                                           const float* gamma, const float* beta, float* rmean,
                                                           float* rvar, long long* nbt, int training, float* ab,
                                                           float* mean_invstd, const float* wq, const float* wk,
                                                           const float* bq, float* md_out) {
    __shared__ double red[2][4][256];
    __shared__ float mred[16][64];
    if (blockIdx.x > 0) {
        // blocks 1 .. 65: row n = blockIdx.x - 1 of md (row 64 = d). Thread = (output k, one of 16 slices of the 128-term sum):
        // eight independent loads per operand in flight, the 16 slice sums added in fixed order through LDS
        const int n = blockIdx.x - 1, k = threadIdx.x & 63, sl = threadIdx.x >> 6;
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = sl * 8 + u;
            a[u] = n < 64 ? wk[o * 64 + n] : bq[o];
            b[u] = n < 64 ? wq[o * 64 + k] : wk[o * 64 + k];
        }
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(a[u], b[u], acc);
        mred[sl][k] = acc;
        __syncthreads();
        if (sl == 0) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) t += mred[j][k];
            md_out[n * 64 + k] = t;
        }
        return;
    }
    const int ch = threadIdx.x & 255, part = threadIdx.x >> 8;
    if (threadIdx.x == 0 && training && nbt) *nbt += 1;
    if (training) {
        double s = 0.0, q = 0.0;
        if (ch < nch) {
            int b = part;
            for (; b + 28 < n_blocks; b += 32) {   // eight blocks' partials in flight (latency-bound loads), added in order
                double ps[8], pq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ps[u] = partials[((int64_t)(b + 4 * u) * 2 + 0) * nch + ch];
                    pq[u] = partials[((int64_t)(b + 4 * u) * 2 + 1) * nch + ch];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s += ps[u]; q += pq[u]; }
            }
            for (; b < n_blocks; b += 4) {
                s += partials[((int64_t)b * 2 + 0) * nch + ch];
                q += partials[((int64_t)b * 2 + 1) * nch + ch];
            }
        }
        red[0][part][ch] = s;
        red[1][part][ch] = q;
    }
    __syncthreads();
    if (part == 0 && ch < nch) {
        float mean, var;
        if (training) {
            const double s = (red[0][0][ch] + red[0][1][ch]) + (red[0][2][ch] + red[0][3][ch]);
            const double q = (red[1][0][ch] + red[1][1][ch]) + (red[1][2][ch] + red[1][3][ch]);
            const double m = s / n_frames;
            double v = q / n_frames - m * m;
            if (v < 0.0) v = 0.0;
            mean = (float)m;
            var = (float)v;
            const double unb = n_frames > 1 ? v * ((double)n_frames / (n_frames - 1)) : v;
            rmean[ch] = 0.9f * rmean[ch] + 0.1f * mean;  // momentum 0.1 (torch default, models_gcn.py:43)
            rvar[ch] = 0.9f * rvar[ch] + 0.1f * (float)unb;
        } else {
            mean = rmean[ch];
            var = rvar[ch];
        }
        const float invstd = 1.0f / sqrtf(var + 1e-5f);
        const float a = gamma[ch] * invstd;
        ab[ch] = a;
        ab[nch + ch] = beta[ch] - mean * a;
        mean_invstd[ch] = mean;
        mean_invstd[nch + ch] = invstd;
    }
}

// e1[(f,n)][j] = relu(sum_c W1[j][c] * (a[ch]*x[f][n][c] + b[ch]) + b1[j]); block = 4 rows x 64 outputs
__global__ __launch_bounds__(256) void embed1_fwd_kernel(const float* x, int64_t fstride, int n_frames, int N,
                                                         const float* ab, const float* w1, const float* b1,
                                                         float* e1) {
    const int j = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const float4 w = *reinterpret_cast<const float4*>(w1 + j * 4);
    const float bj = b1[j];
    const int nch = 4 * N;
    const int64_t rows = (int64_t)n_frames * N;
    for (int64_t r = (int64_t)blockIdx.x * 4 + rl; r < rows; r += (int64_t)gridDim.x * 4) {
        const int f = (int)(r / N), n = (int)(r - (int64_t)f * N);
        const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)f * fstride + n * 4);
        // (explicit fused multiply-adds: bit-identical to the copy of this arithmetic inside gcn_fused_fwd_kernel,
        // geo_fused.hip -- the backward pass recomputes e1 here and masks with its signs)
        const float x0 = fmaf(ab[n], v.x, ab[nch + n]);
        const float x1 = fmaf(ab[N + n], v.y, ab[nch + N + n]);
        const float x2 = fmaf(ab[2 * N + n], v.z, ab[nch + 2 * N + n]);
        const float x3 = fmaf(ab[3 * N + n], v.w, ab[nch + 3 * N + n]);
        float acc = bj;
        acc = fmaf(w.x, x0, acc);
        acc = fmaf(w.y, x1, acc);
        acc = fmaf(w.z, x2, acc);
        acc = fmaf(w.w, x3, acc);
        e1[r * 64 + j] = fmaxf(acc, 0.f);
    }
}

// Backward of embed1 given de1 = dL/d(pre-activation) (already ReLU-masked). Per block partial sums of
// dW1[64][4], db1[64], da[4N], db[4N] in `partials` (row = block, layout [256 | 64 | 4N | 4N]).
__global__ __launch_bounds__(256) void embed1_bwd_kernel(const float* x, int64_t fstride, int n_frames, int N,
                                                         const float* ab, const float* w1, const float* de1,
                                                         float* partials) {
    __shared__ float s_dw[4][64][5];           // per wave: dW1[j][0..3], db1[j]
    __shared__ float s_ab[4][2][4 * MAX_NODES];  // per wave: da[ch], db[ch]
    const int j = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nch = 4 * N;
    const float4 w = *reinterpret_cast<const float4*>(w1 + j * 4);
    for (int i = j; i < 2 * 4 * MAX_NODES; i += 64) (&s_ab[wv][0][0])[i] = 0.f;
    float dw0 = 0.f, dw1 = 0.f, dw2 = 0.f, dw3 = 0.f, dbj = 0.f;
    const int64_t rows = (int64_t)n_frames * N;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < rows; r += (int64_t)gridDim.x * 4) {
        const int f = (int)(r / N), n = (int)(r - (int64_t)f * N);
        const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)f * fstride + n * 4);
        const float x0 = ab[n] * v.x + ab[nch + n];
        const float x1 = ab[N + n] * v.y + ab[nch + N + n];
        const float x2 = ab[2 * N + n] * v.z + ab[nch + 2 * N + n];
        const float x3 = ab[3 * N + n] * v.w + ab[nch + 3 * N + n];
        const float d = de1[r * 64 + j];
        dw0 = fmaf(d, x0, dw0);
        dw1 = fmaf(d, x1, dw1);
        dw2 = fmaf(d, x2, dw2);
        dw3 = fmaf(d, x3, dw3);
        dbj += d;
        // dx^[c] = sum_j d_j W1[j][c]  (wave reduction over the 64 outputs)
        const float g0 = wave_sum(d * w.x), g1 = wave_sum(d * w.y), g2 = wave_sum(d * w.z), g3 = wave_sum(d * w.w);
        if (j < 4) {
            const float g = j == 0 ? g0 : j == 1 ? g1 : j == 2 ? g2 : g3;
            const float xv = j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
            s_ab[wv][0][j * N + n] += g * xv;  // d a[ch]
            s_ab[wv][1][j * N + n] += g;       // d b[ch]
        }
    }
    s_dw[wv][j][0] = dw0; s_dw[wv][j][1] = dw1; s_dw[wv][j][2] = dw2; s_dw[wv][j][3] = dw3; s_dw[wv][j][4] = dbj;
    __syncthreads();
    float* out = partials + (int64_t)blockIdx.x * (320 + 2 * nch);
    for (int i = threadIdx.x; i < 320; i += 256) {
        const int jj = i < 256 ? i >> 2 : i - 256, k = i < 256 ? (i & 3) : 4;
        out[i] = s_dw[0][jj][k] + s_dw[1][jj][k] + s_dw[2][jj][k] + s_dw[3][jj][k];
    }
    for (int i = threadIdx.x; i < 2 * nch; i += 256) {
        const int which = i / nch, ch = i - which * nch;
        out[320 + i] = s_ab[0][which][ch] + s_ab[1][which][ch] + s_ab[2][which][ch] + s_ab[3][which][ch];
    }
}

// reduce embed1 partials over blocks; convert (da, db) to (dgamma, dbeta):
//   dgamma = invstd * (da - mean * db') ... with x^ = a x + b, a = gamma*invstd, b = beta - mean*a:
//   dL/dgamma = invstd * (dL/da - mean * dL/db), dL/dbeta = dL/db        (mean/invstd constant wrt the parameters)
__global__ __launch_bounds__(1024) void embed1_bwd_final_kernel(const float* partials, int n_blocks, int nch,
                                                                const float* mean_invstd, float* dw1, float* db1,
                                                                float* dgamma, float* dbeta) {
    // 64 outputs x 16 lanes over the per-block partial rows; ordered LDS reduction (deterministic)
    __shared__ float red[2][16][64];
    const int stride = 320 + 2 * nch;
    const int ol = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + ol;
    float sa = 0.f, sb = 0.f;
    if (i < 320) {
        for (int b = rl; b < n_blocks; b += 16) sa += partials[(int64_t)b * stride + i];
    } else if (i < 320 + nch) {
        const int ch = i - 320;
        for (int b = rl; b < n_blocks; b += 16) {
            sa += partials[(int64_t)b * stride + 320 + ch];
            sb += partials[(int64_t)b * stride + 320 + nch + ch];
        }
    }
    red[0][rl][ol] = sa;
    red[1][rl][ol] = sb;
    __syncthreads();
    if (rl == 0 && i < 320 + nch) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { ta += red[0][k][ol]; tb += red[1][k][ol]; }
        if (i < 256) dw1[i] = ta;
        else if (i < 320) db1[i - 256] = ta;
        else {
            const int ch = i - 320;
            dgamma[ch] = mean_invstd[nch + ch] * (ta - mean_invstd[ch] * tb);
            dbeta[ch] = tb;
        }
    }
}

constexpr int LDQ = 132;  // 128 + 4: lane j's float4 reads land on banks 4j (mod 64) -> conflict-free b128
constexpr int LDX = 68;   // 64 + 4

__device__ __forceinline__ void stage_rows(float* dst, int ld, const float* src, int src_ld, int rows, int cols) {
    const int c4 = cols >> 2;
    for (int i = threadIdx.x; i < rows * c4; i += blockDim.x) {
        const int r = i / c4, c = (i - r * c4) * 4;
        *reinterpret_cast<float4*>(dst + r * ld + c) = *reinterpret_cast<const float4*>(src + (int64_t)r * src_ld + c);
    }
}

// per frame: S = softmax_j(Q_i . K_j) (no 1/sqrt(d), models_gcn.py:97-100), Z = S X (models_gcn.py:33-34)
__global__ __launch_bounds__(256) void gcn_attn_fwd_kernel(const float* qk, const float* xin, int n_frames, int N,
                                                           float* s_out, float* z) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sQ = sm;
    float* sK = sQ + N * LDQ;
    float* sX = sK + N * LDQ;
    float* sS = sX + N * LDX;  // [N][N+1]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int LDS_S = N + 1;
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const int64_t r0 = (int64_t)f * N;
        __syncthreads();
        stage_rows(sQ, LDQ, qk + r0 * 256, 256, N, 128);
        stage_rows(sK, LDQ, qk + r0 * 256 + 128, 256, N, 128);
        stage_rows(sX, LDX, xin + r0 * 64, 64, N, 64);
        __syncthreads();
        for (int i = wv; i < N; i += 4) {
            float p = -INFINITY;
            if (lane < N) {
                float acc = 0.f;
                const float4* qa = reinterpret_cast<const float4*>(sQ + i * LDQ);
                const float4* kb = reinterpret_cast<const float4*>(sK + lane * LDQ);
#pragma unroll 8
                for (int d = 0; d < 32; ++d) {
                    const float4 a = qa[d], b = kb[d];
                    acc = fmaf(a.x, b.x, acc);
                    acc = fmaf(a.y, b.y, acc);
                    acc = fmaf(a.z, b.z, acc);
                    acc = fmaf(a.w, b.w, acc);
                }
                p = acc;
            }
            const float m = wave_max(p);
            const float e = lane < N ? expf(p - m) : 0.f;
            const float s = wave_sum(e);
            if (lane < N) {
                const float v = e / s;
                sS[i * LDS_S + lane] = v;
                s_out[(r0 + i) * N + lane] = v;
            }
        }
        __syncthreads();
        for (int i = wv; i < N; i += 4) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(sS[i * LDS_S + j], sX[j * LDX + lane], acc);
            z[(r0 + i) * 64 + lane] = acc;
        }
    }
}

// per frame backward: dS = dZ X^T, dX_att = S^T dZ, dP = S*(dS - rowsum(dS*S)), dQ = dP K, dK = dP^T Q
__global__ __launch_bounds__(256) void gcn_attn_bwd_kernel(const float* qk, const float* xin, const float* s_in,
                                                           const float* dz, int n_frames, int N, float* dx_att,
                                                           float* dqk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sQ = sm;
    float* sK = sQ + N * LDQ;
    float* sX = sK + N * LDQ;
    float* sdZ = sX + N * LDX;
    float* sS = sdZ + N * LDX;      // [N][N+1]
    float* sdP = sS + N * (N + 1);  // [N][N+1]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int LDS_S = N + 1;
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const int64_t r0 = (int64_t)f * N;
        __syncthreads();
        stage_rows(sQ, LDQ, qk + r0 * 256, 256, N, 128);
        stage_rows(sK, LDQ, qk + r0 * 256 + 128, 256, N, 128);
        stage_rows(sX, LDX, xin + r0 * 64, 64, N, 64);
        stage_rows(sdZ, LDX, dz + r0 * 64, 64, N, 64);
        for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
            const int r = i / N, c = i - r * N;
            sS[r * LDS_S + c] = s_in[r0 * N + i];
        }
        __syncthreads();
        for (int i = wv; i < N; i += 4) {
            float ds = 0.f, sv = 0.f;
            if (lane < N) {
                const float4* za = reinterpret_cast<const float4*>(sdZ + i * LDX);
                const float4* xb = reinterpret_cast<const float4*>(sX + lane * LDX);
#pragma unroll 8
                for (int k = 0; k < 16; ++k) {
                    const float4 a = za[k], b = xb[k];
                    ds = fmaf(a.x, b.x, ds);
                    ds = fmaf(a.y, b.y, ds);
                    ds = fmaf(a.z, b.z, ds);
                    ds = fmaf(a.w, b.w, ds);
                }
                sv = sS[i * LDS_S + lane];
            }
            const float t = wave_sum(ds * sv);
            if (lane < N) sdP[i * LDS_S + lane] = sv * (ds - t);
        }
        __syncthreads();
        for (int j = wv; j < N; j += 4) {
            float acc = 0.f;  // dX_att[j][lane] = sum_i S[i][j] dZ[i][lane]
            for (int i = 0; i < N; ++i) acc = fmaf(sS[i * LDS_S + j], sdZ[i * LDX + lane], acc);
            dx_att[(r0 + j) * 64 + lane] = acc;
            float q0 = 0.f, q1 = 0.f, k0 = 0.f, k1 = 0.f;
            for (int i = 0; i < N; ++i) {
                const float pji = sdP[j * LDS_S + i];  // dQ[j][d] = sum_i dP[j][i] K[i][d]
                const float pij = sdP[i * LDS_S + j];  // dK[j][d] = sum_i dP[i][j] Q[i][d]
                q0 = fmaf(pji, sK[i * LDQ + lane], q0);
                q1 = fmaf(pji, sK[i * LDQ + 64 + lane], q1);
                k0 = fmaf(pij, sQ[i * LDQ + lane], k0);
                k1 = fmaf(pij, sQ[i * LDQ + 64 + lane], k1);
            }
            float* o = dqk + (r0 + j) * 256;
            o[lane] = q0;
            o[64 + lane] = q1;
            o[128 + lane] = k0;
            o[192 + lane] = k1;
        }
    }
}

}  // namespace

extern "C" int twog_gcn_max_nodes(void) { return MAX_NODES; }

extern "C" int twog_bn_stats(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes, double* partials,
                             int n_blocks, void* stream) {
    if (n_nodes > MAX_NODES || n_nodes < 1) return -1;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, x_geo, frame_stride,
                       n_frames, n_nodes, partials);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_bn_finalize(const double* partials, int n_blocks, int n_frames, int n_nodes, const float* gamma,
                                const float* beta, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, int training, float* ab, float* mean_invstd,
                                const float* wq, const float* wk, const float* bq, float* md_out, void* stream) {
    if (n_nodes > MAX_NODES || n_nodes < 1) return -1;
    if (md_out && !(wq && wk && bq)) return -2;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(md_out ? 66 : 1), dim3(1024), 0, (hipStream_t)stream, partials, n_blocks, n_frames,
                       4 * n_nodes, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, training,
                       ab, mean_invstd, wq, wk, bq, md_out);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_gcn_embed1_fwd(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes,
                                   const float* ab, const float* w1, const float* b1, float* e1, void* stream) {
    if (n_nodes > MAX_NODES || n_nodes < 1) return -1;
    const int64_t rows = (int64_t)n_frames * n_nodes;
    int grid = (int)((rows + 3) / 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(embed1_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_geo, frame_stride, n_frames,
                       n_nodes, ab, w1, b1, e1);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_gcn_embed1_bwd(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes,
                                   const float* ab, const float* mean_invstd, const float* w1, const float* de1,
                                   float* partials, int n_blocks, float* dw1, float* db1, float* dgamma, float* dbeta,
                                   void* stream) {
    if (n_nodes > MAX_NODES || n_nodes < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(embed1_bwd_kernel, dim3(n_blocks), dim3(256), 0, st, x_geo, frame_stride, n_frames, n_nodes, ab,
                       w1, de1, partials);
    TWOG_CHECK_LAUNCH();
    hipLaunchKernelGGL(embed1_bwd_final_kernel, dim3((320 + 4 * n_nodes + 63) / 64), dim3(1024), 0, st, partials, n_blocks, 4 * n_nodes,
                       mean_invstd, dw1, db1, dgamma, dbeta);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_gcn_attn_fwd(const float* qk, const float* x, int n_frames, int n_nodes, float* s_out, float* z,
                                 void* stream) {
    if (n_nodes > MAX_NODES || n_nodes < 1) return -1;
    const size_t lds = sizeof(float) * (size_t)(2 * n_nodes * LDQ + n_nodes * LDX + n_nodes * (n_nodes + 1));
    int grid = n_frames < 2048 ? n_frames : 2048;
    hipLaunchKernelGGL(gcn_attn_fwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, qk, x, n_frames, n_nodes,
                       s_out, z);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_gcn_attn_bwd(const float* qk, const float* x, const float* s, const float* dz, int n_frames,
                                 int n_nodes, float* dx_att, float* dqk, void* stream) {
    if (n_nodes > MAX_NODES || n_nodes < 1) return -1;
    const size_t lds = sizeof(float) * (size_t)(2 * n_nodes * LDQ + 2 * n_nodes * LDX + 2 * n_nodes * (n_nodes + 1));
    int grid = n_frames < 2048 ? n_frames : 2048;
    hipLaunchKernelGGL(gcn_attn_bwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, qk, x, s, dz, n_frames,
                       n_nodes, dx_att, dqk);
    TWOG_CHECK_LAUNCH();
    return 0;
}
