// Segment-boundary gates ("state change detector").
//
// Reference: _update_human_segment / _update_object_segment (vhoi/models.py:1477-1533) with
// discrete_networks_num_layers == 1: p = sigmoid(Linear([x, h_f, messages...] -> 1)); discrete_estimator (:1620-1627):
//   'gs': y = softmax((log([p, 1-p] + 1e-20) + g) / 1)[0], g ~ Gumbel(0,1) PRE-DRAWN on the host in the reference's call
//         order (t-major, humans then objects; pyrutils/torch/distributions.py:4-36); hard = (y > thr); the value
//         used downstream is (hard - y).detach() + y, i.e. `hard` in the forward pass and d/dy = 1 in the backward pass;
//   'st': hard = (p > thr), soft = p (distributions.py:39-53).
// The hard gate of the last (padded) step is overwritten with 1 in place, which also cuts its gradient (:701-702).
// One wave per entity row: the row's gate-input blocks are read in place from the concatenated entity buffer (the
// weight vector is applied block-wise through seg_col[], so no torch.cat is materialised), reduced with wave shuffles.
#include "twog_common.h"

namespace {

__global__ __launch_bounds__(256) void gate_fwd_kernel(const twog_gate_t G) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t rows = (int64_t)G.bs * G.T * G.E;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    if (row >= rows) return;
    const float* x = twog_row_ptr(G.x, (int)row);
    float acc = 0.f;
    for (int s = 0; s < G.n_seg; ++s) {
        const float* xs = x + G.seg_col[s];
        const float* ws = G.w + s * G.hidden;
        for (int j = lane; j < G.hidden; j += 64) acc = fmaf(xs[j], ws[j], acc);
    }
    acc = wave_sum(acc);
    if (lane != 0) return;
    if (G.b) acc += G.b[0];
    const float p = 1.0f / (1.0f + expf(-acc));
    const int e = (int)(row % G.E);
    const int t = (int)((row / G.E) % G.T);
    const int b = (int)(row / ((int64_t)G.E * G.T));
    float y;
    if (G.noise) {
        const float* g = G.noise + (((int64_t)t * G.noise_entities + G.noise_offset + e) * G.bs + b) * 2;
        const float a0 = logf(p + 1e-20f) + g[0];
        const float a1 = logf((1.0f - p) + 1e-20f) + g[1];
        const float m = fmaxf(a0, a1);
        const float e0 = expf(a0 - m), e1 = expf(a1 - m);
        y = e0 / (e0 + e1);
    } else {
        y = p;
    }
    float hard = y > G.threshold ? 1.f : 0.f;
    if (G.force_last && t == G.T - 1) hard = 1.f;
    G.hard[row] = hard;
    G.soft[row] = y;
    G.p_save[row] = p;
}

// total gradient reaching the soft decision -> gradient wrt the pre-sigmoid logit
__global__ __launch_bounds__(256) void gate_bwd_kernel(const twog_gate_t G, const float* d_hard, const float* d_soft,
                                                       const float* st_mask, float* dlogit) {
    const int64_t rows = (int64_t)G.bs * G.T * G.E;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int t = (int)((row / G.E) % G.T);
    float d = d_soft ? d_soft[row] : 0.f;
    if (d_hard) {
        float m = st_mask ? st_mask[row] : 1.f;
        if (G.force_last && t == G.T - 1) m = 0.f;
        d = fmaf(d_hard[row], m, d);
    }
    const float p = G.p_save[row];
    float dp = d;
    if (G.noise) {
        const float y = G.soft[row];
        dp = d * y * (1.0f - y) * (1.0f / (p + 1e-20f) + 1.0f / ((1.0f - p) + 1e-20f));
    }
    dlogit[row] = dp * p * (1.0f - p);
}

// dst[r][c] += s[r] * v[c]
__global__ __launch_bounds__(256) void rank1_kernel(twog_rows_t dst, const float* s, const float* v, int rows, int cols) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * cols;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        float* d = twog_row_ptr(dst, r) + c;
        *d = fmaf(s[r], v[c], *d);
    }
}

// partial sums of s[r] * x[r][c] over row slabs (s may be NULL -> 1).
// VEC: every lane owns 4 consecutive columns (one 16-byte load per row: 1 KiB per wave instruction) and 4 independent
// row streams are kept in flight; block = 64 column-quads x 4 row lanes.
template <bool VEC>
__device__ __forceinline__ void wcolsum_partial_body(const twog_rows_t& x, const float* s, int rows, int cols, float* partials,
                                                     int col_block, int row_block, int n_row_blocks) {
    constexpr int CW = VEC ? 4 : 1;
    __shared__ float red[4][64 * CW];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = (col_block * 64 + cl) * CW;
    const int per = (rows + n_row_blocks - 1) / n_row_blocks;
    const int r0 = row_block * per, r1 = min(rows, r0 + per);
    float acc[CW];
#pragma unroll
    for (int i = 0; i < CW; ++i) acc[i] = 0.f;
    if (c < cols) {
        if constexpr (VEC) {
            float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0;
            int r = r0 + rl;
            for (; r + 12 < r1; r += 16) {
                const float4 v0 = *reinterpret_cast<const float4*>(twog_row_ptr(x, r) + c);
                const float4 v1 = *reinterpret_cast<const float4*>(twog_row_ptr(x, r + 4) + c);
                const float4 v2 = *reinterpret_cast<const float4*>(twog_row_ptr(x, r + 8) + c);
                const float4 v3 = *reinterpret_cast<const float4*>(twog_row_ptr(x, r + 12) + c);
                const float s0 = s ? s[r] : 1.f, s1 = s ? s[r + 4] : 1.f, s2 = s ? s[r + 8] : 1.f, s3 = s ? s[r + 12] : 1.f;
                a0.x = fmaf(s0, v0.x, a0.x); a0.y = fmaf(s0, v0.y, a0.y); a0.z = fmaf(s0, v0.z, a0.z); a0.w = fmaf(s0, v0.w, a0.w);
                a1.x = fmaf(s1, v1.x, a1.x); a1.y = fmaf(s1, v1.y, a1.y); a1.z = fmaf(s1, v1.z, a1.z); a1.w = fmaf(s1, v1.w, a1.w);
                a2.x = fmaf(s2, v2.x, a2.x); a2.y = fmaf(s2, v2.y, a2.y); a2.z = fmaf(s2, v2.z, a2.z); a2.w = fmaf(s2, v2.w, a2.w);
                a3.x = fmaf(s3, v3.x, a3.x); a3.y = fmaf(s3, v3.y, a3.y); a3.z = fmaf(s3, v3.z, a3.z); a3.w = fmaf(s3, v3.w, a3.w);
            }
            for (; r < r1; r += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(twog_row_ptr(x, r) + c);
                const float s0 = s ? s[r] : 1.f;
                a0.x = fmaf(s0, v0.x, a0.x); a0.y = fmaf(s0, v0.y, a0.y); a0.z = fmaf(s0, v0.z, a0.z); a0.w = fmaf(s0, v0.w, a0.w);
            }
            acc[0] = (a0.x + a1.x) + (a2.x + a3.x);
            acc[1 % CW] = (a0.y + a1.y) + (a2.y + a3.y);
            acc[2 % CW] = (a0.z + a1.z) + (a2.z + a3.z);
            acc[3 % CW] = (a0.w + a1.w) + (a2.w + a3.w);
        } else {
            for (int r = r0 + rl; r < r1; r += 4) acc[0] = fmaf(s ? s[r] : 1.f, twog_row_ptr(x, r)[c], acc[0]);
        }
    }
#pragma unroll
    for (int i = 0; i < CW; ++i) red[rl][cl * CW + i] = acc[i];
    __syncthreads();
    if (rl == 0 && c < cols) {
#pragma unroll
        for (int i = 0; i < CW; ++i)
            partials[(int64_t)row_block * cols + c + i] =
                (red[0][cl * CW + i] + red[1][cl * CW + i]) + (red[2][cl * CW + i] + red[3][cl * CW + i]);
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void wcolsum_partial_kernel(twog_rows_t x, const float* s, int rows, int cols,
                                                              float* partials) {
    wcolsum_partial_body<VEC>(x, s, rows, cols, partials, blockIdx.x, blockIdx.y, gridDim.y);
}

__device__ __forceinline__ void wcolsum_final_body(const float* partials, int n_blocks, int cols, float* out, int accumulate,
                                                   int col_block) {
    // 64 columns x 16 lanes over the partial rows, then an ordered LDS reduction (deterministic)
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = col_block * 64 + cl;
    float acc = 0.f;
    if (c < cols)
        for (int b = rl; b < n_blocks; b += 16) acc += partials[(int64_t)b * cols + c];
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ __launch_bounds__(1024) void wcolsum_final_kernel(const float* partials, int n_blocks, int cols, float* out,
                                                             int accumulate) {
    wcolsum_final_body(partials, n_blocks, cols, out, accumulate, blockIdx.x);
}

// Several column sums in ONE pair of launches (twog_colsum_n): the bias gradients of a backward stage -- 45 column sums per
// step, each a partial + a final launch until round 5 -- are collected by the host and issued together. Workgroup ->
// (problem, column block, row block) through the prefix tables; each problem keeps the arithmetic (and so the results,
// bit for bit) of its own twog_colsum call.
struct ColsumBatch {
    twog_colsum_t op[TWOG_COLSUM_MAX];
    int n;
    int first_block[TWOG_COLSUM_MAX + 1];   // partial launch: blocks of problem i are [first_block[i], first_block[i + 1])
    int first_final[TWOG_COLSUM_MAX + 1];   // final launch
    int col_blocks[TWOG_COLSUM_MAX];        // column blocks of the partial launch (256 columns vectorised, else 64)
    int row_blocks[TWOG_COLSUM_MAX];
    int vec[TWOG_COLSUM_MAX];
    int64_t part_off[TWOG_COLSUM_MAX];      // floats
};
__global__ __launch_bounds__(256) void wcolsum_partial_n_kernel(const ColsumBatch G, float* partials) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < G.n; ++i)
        if ((int)blockIdx.x >= G.first_block[i]) pi = i;
    const twog_colsum_t& o = G.op[pi];
    const int local = (int)blockIdx.x - G.first_block[pi];
    const int cb = local % G.col_blocks[pi], rb = local / G.col_blocks[pi];
    if (G.vec[pi]) wcolsum_partial_body<true>(o.x, o.rowscale, o.rows, o.cols, partials + G.part_off[pi], cb, rb, G.row_blocks[pi]);
    else wcolsum_partial_body<false>(o.x, o.rowscale, o.rows, o.cols, partials + G.part_off[pi], cb, rb, G.row_blocks[pi]);
}
__global__ __launch_bounds__(1024) void wcolsum_final_n_kernel(const ColsumBatch G, const float* partials) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < G.n; ++i)
        if ((int)blockIdx.x >= G.first_final[i]) pi = i;
    const twog_colsum_t& o = G.op[pi];
    wcolsum_final_body(partials + G.part_off[pi], G.row_blocks[pi], o.cols, o.out, o.accumulate, (int)blockIdx.x - G.first_final[pi]);
}

}  // namespace

extern "C" int twog_gate_fwd(const twog_gate_t* g, void* stream) {
    const int64_t rows = (int64_t)g->bs * g->T * g->E;
    if (rows <= 0) return 0;
    if (g->n_seg > 8) return -1;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *g);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_gate_bwd(const twog_gate_t* g, const float* d_hard, const float* d_soft, const float* st_mask,
                             float* dlogit, void* stream) {
    const int64_t rows = (int64_t)g->bs * g->T * g->E;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *g,
                       d_hard, d_soft, st_mask, dlogit);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// the same update with 16-byte accesses (cols % 4 == 0, 16-byte aligned rows): one thread per four columns of a row, the row's
// scale read once, no 64-bit division per element; bit-identical (one fmaf per element)
__global__ __launch_bounds__(256) void rank1_vec_kernel(twog_rows_t dst, const float* s, const float* v, int rows, int qpr) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const unsigned total = (unsigned)rows * (unsigned)qpr;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / (unsigned)qpr, q = i - r * (unsigned)qpr;
        f4* d = reinterpret_cast<f4*>(twog_row_ptr(dst, (int)r) + 4 * q);
        const f4 w = *reinterpret_cast<const f4*>(v + 4 * q);
        const float sr = s[r];
        f4 x = *d;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = fmaf(sr, w[k], x[k]);
        *d = x;
    }
}

extern "C" int twog_rank1_update(twog_rows_t dst, const float* s, const float* v, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    if ((cols & 3) == 0 && (reinterpret_cast<uintptr_t>(dst.ptr) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0 &&
        (dst.ld_outer & 3) == 0 && (dst.inner <= 1 || (dst.ld_inner & 3) == 0) && (int64_t)rows * (cols >> 2) < (int64_t(1) << 31)) {
        const int64_t nq = (int64_t)rows * (cols >> 2);
        const int grid = (int)((nq + 255) / 256 > 8192 ? 8192 : (nq + 255) / 256);
        hipLaunchKernelGGL(rank1_vec_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, s, v, rows, cols >> 2);
        TWOG_CHECK_LAUNCH();
        return 0;
    }
    int64_t n = (int64_t)rows * cols;
    int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(rank1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, s, v, rows, cols);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// row slices of one problem: enough to fill the chip (~1024 workgroups over column blocks x row slices), >= 64 rows per slice
static int colsum_row_blocks(int rows, int cols) {
    const int cb = (cols + 255) / 256;
    int most = 1024 / (cb > 0 ? cb : 1);
    if (most < 1) most = 1;
    int n = rows / 64;
    if (n > most) n = most;
    return n < 1 ? 1 : n;
}

extern "C" size_t twog_colsum_n_partial_floats(const twog_colsum_t* ops, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i)
        if (ops[i].cols > 0 && ops[i].rows >= 0) total += (size_t)colsum_row_blocks(ops[i].rows, ops[i].cols) * ops[i].cols;
    return total;
}

extern "C" int twog_colsum_n(const twog_colsum_t* ops, int n, float* partials, size_t partial_floats, void* stream) {
    if (n <= 0) return 0;
    if (!ops || !partials) return -2;
    hipStream_t st = (hipStream_t)stream;
    for (int done = 0; done < n; done += TWOG_COLSUM_MAX) {
        ColsumBatch G;
        G.n = 0;
        int blocks = 0, finals = 0;
        size_t off = 0;
        for (int i = done; i < n && G.n < TWOG_COLSUM_MAX; ++i) {
            const twog_colsum_t& o = ops[i];
            if (o.cols <= 0) continue;
            if (o.rows < 0 || !o.out || (o.rows > 0 && !o.x.ptr)) return -2;
            const int k = G.n++;
            G.op[k] = o;
            G.vec[k] = (o.cols % 4 == 0) && (reinterpret_cast<uintptr_t>(o.x.ptr) % 16 == 0) && (o.x.ld_outer % 4 == 0) &&
                       (o.x.inner <= 1 || o.x.ld_inner % 4 == 0);
            G.col_blocks[k] = G.vec[k] ? (o.cols + 255) / 256 : (o.cols + 63) / 64;
            G.row_blocks[k] = colsum_row_blocks(o.rows, o.cols);
            G.first_block[k] = blocks;
            G.first_final[k] = finals;
            G.part_off[k] = (int64_t)off;
            blocks += G.col_blocks[k] * G.row_blocks[k];
            finals += (o.cols + 63) / 64;
            off += (size_t)G.row_blocks[k] * o.cols;
        }
        if (G.n == 0) continue;
        if (off > partial_floats) return -2;
        G.first_block[G.n] = blocks;
        G.first_final[G.n] = finals;
        hipLaunchKernelGGL(wcolsum_partial_n_kernel, dim3(blocks), dim3(256), 0, st, G, partials);
        TWOG_CHECK_LAUNCH();
        hipLaunchKernelGGL(wcolsum_final_n_kernel, dim3(finals), dim3(1024), 0, st, G, partials);
        TWOG_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int twog_colsum(twog_rows_t x, const float* rowscale, int rows, int cols, float* out, int accumulate,
                           float* partials, int n_blocks, void* stream) {
    if (cols <= 0) return 0;
    if (n_blocks < 1) n_blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (cols % 4 == 0) && (reinterpret_cast<uintptr_t>(x.ptr) % 16 == 0) && (x.ld_outer % 4 == 0) &&
                     (x.inner <= 1 || x.ld_inner % 4 == 0);
    if (vec)
        hipLaunchKernelGGL(wcolsum_partial_kernel<true>, dim3((cols + 255) / 256, n_blocks), dim3(256), 0, st, x,
                           rowscale, rows, cols, partials);
    else
        hipLaunchKernelGGL(wcolsum_partial_kernel<false>, dim3((cols + 63) / 64, n_blocks), dim3(256), 0, st, x,
                           rowscale, rows, cols, partials);
    TWOG_CHECK_LAUNCH();
    hipLaunchKernelGGL(wcolsum_final_kernel, dim3((cols + 63) / 64), dim3(1024), 0, st, partials, n_blocks, cols, out,
                       accumulate);
    TWOG_CHECK_LAUNCH();
    return 0;
}
