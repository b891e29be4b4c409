// Optional position features of the segment level and small element-wise helpers of the less common gate strategies.
//
// Reference: TGGCN._assemble_time_tensor (vhoi/models.py:935-952), _assemble_segment_length_tensor (:954-981),
// make_periodic_embedding (:1778-1794) and their use at :655-662 (time feature of the gate MLPs, strategy 'u') and
// :754-779 (time / segment-length features appended to the GRUCell inputs); the 'conditional_on_human' object gate
// (:1531-1532). Disabled in every shipped configuration -- streaming one-pass kernels, written for correctness first.
#include "twog_common.h"

namespace {

// out[(b,t,e)][j] = relu(w[j] * s + b[j])          (build_mlp([1, h], ['relu']))
//               or  [sin(s / w_k) | cos(s / w_k)],  w_k = 1e4 ^ (k / (h/2 - 1))        (periodic)
// s = given scalar per row, or the time feature (t + 1) [/ steps[b]]
__global__ __launch_bounds__(256) void pos_embed_fwd_kernel(const float* s_in, const float* steps, int T, int E,
                                                            int divide, const float* w, const float* b, int periodic,
                                                            int hidden, twog_rows_t out, float* s_out, int rows) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    float s;
    if (s_in) s = s_in[r];
    else {
        const int bt = r / E, bi = bt / T, t = bt - bi * T;
        s = (float)(t + 1);
        if (divide) s = s / steps[bi];
    }
    if (s_out && threadIdx.x == 0) s_out[r] = s;
    float* o = twog_row_ptr(out, r);
    const int half = hidden >> 1;
    for (int j = threadIdx.x; j < hidden; j += blockDim.x) {
        float v;
        if (periodic) {
            const int k = j < half ? j : j - half;
            const float e = half > 1 ? (float)k / (float)(half - 1) : 0.f;
            const float wk = powf(1e4f, e);
            v = j < half ? sinf(s / wk) : cosf(s / wk);
        } else {
            v = fmaf(w[j], s, b ? b[j] : 0.f);
            v = v > 0.f ? v : 0.f;
        }
        o[j] = v;
    }
}

// periodic embedding: ds[r] = sum_k ( dout[k] cos(s / w_k) - dout[half + k] sin(s / w_k) ) / w_k
__global__ __launch_bounds__(256) void periodic_bwd_kernel(twog_rows_t dout, const float* s_in, int hidden, float* ds,
                                                           int rows) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    if (r >= rows) return;
    const float s = s_in[r];
    const float* d = twog_row_ptr(dout, r);
    const int half = hidden >> 1;
    float acc = 0.f;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        const float e = half > 1 ? (float)k / (float)(half - 1) : 0.f;
        const float wk = powf(1e4f, e);
        acc += (d[k] * cosf(s / wk) - d[half + k] * sinf(s / wk)) / wk;
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) ds[r] = t;
}

// segment lengths (vhoi/models.py:973-979): per (clip, entity) a scan over time
//   rel_t = u_t * x_t ; if rel_t != 0: rel_t -= acc ; acc += rel_t        x_t = (t + 1) [/ steps[b]]
__global__ __launch_bounds__(256) void seglen_fwd_kernel(const float* u, const float* steps, int bs, int T, int E,
                                                         int divide, float* s_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * E) return;
    const int b = i / E, e = i - b * E;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        const int64_t off = ((int64_t)b * T + t) * E + e;
        const float xt = divide ? (float)(t + 1) / steps[b] : (float)(t + 1);
        float rel = u[off] * xt;
        if (rel != 0.f) rel -= acc;
        acc += rel;
        s_out[off] = rel;
    }
}

// reverse scan: du_t += x_t * (ds_t + da_{t+1}) ; da_t = da_{t+1} - [rel branch taken] * (ds_t + da_{t+1})
__global__ __launch_bounds__(256) void seglen_bwd_kernel(const float* u, const float* steps, int bs, int T, int E,
                                                         int divide, const float* ds, float* du) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * E) return;
    const int b = i / E, e = i - b * E;
    float da = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const int64_t off = ((int64_t)b * T + t) * E + e;
        const float xt = divide ? (float)(t + 1) / steps[b] : (float)(t + 1);
        const float dr = ds[off] + da;
        du[off] += dr * xt;
        if (u[off] * xt != 0.f) da -= dr;
    }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* a, const float* b, float* out, int64_t n, int accumulate) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = accumulate ? out[i] + a[i] * b[i] : a[i] * b[i];
}

// x[r][c] *= s[r]
__global__ __launch_bounds__(256) void scale_rows_kernel(twog_rows_t x, const float* s, int rows, int cols) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * cols;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        twog_row_ptr(x, r)[c] *= s[r];
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int twog_pos_embed_fwd(const float* s, const float* steps, int bs, int T, int E, int divide, const float* w,
                                  const float* b, int periodic, int hidden, twog_rows_t out, float* s_out,
                                  void* stream) {
    const int rows = bs * T * E;
    if (rows <= 0) return 0;
    if (hidden <= 0 || (periodic && (hidden & 1)) || (!periodic && !w) || (!s && divide && !steps)) return -2;
    hipLaunchKernelGGL(pos_embed_fwd_kernel, dim3(rows), dim3(hidden >= 256 ? 256 : 64), 0, (hipStream_t)stream, s,
                       steps, T, E, divide, w, b, periodic, hidden, out, s_out, rows);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_periodic_embed_bwd(twog_rows_t dout, const float* s, int rows, int hidden, float* ds,
                                       void* stream) {
    if (rows <= 0) return 0;
    if (hidden <= 0 || (hidden & 1)) return -2;
    hipLaunchKernelGGL(periodic_bwd_kernel, dim3(rows), dim3(hidden >= 512 ? 256 : 64), 0, (hipStream_t)stream, dout, s,
                       hidden, ds, rows);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_seglen_fwd(const float* u, const float* steps, int bs, int T, int E, int divide, float* s_out,
                               void* stream) {
    if (bs * E <= 0 || T <= 0) return 0;
    if (divide && !steps) return -2;
    hipLaunchKernelGGL(seglen_fwd_kernel, dim3((bs * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, u, steps, bs, T,
                       E, divide, s_out);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_seglen_bwd(const float* u, const float* steps, int bs, int T, int E, int divide, const float* ds,
                               float* du, void* stream) {
    if (bs * E <= 0 || T <= 0) return 0;
    if (divide && !steps) return -2;
    hipLaunchKernelGGL(seglen_bwd_kernel, dim3((bs * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, u, steps, bs, T,
                       E, divide, ds, du);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_mul(const float* a, const float* b, float* out, int64_t n, int accumulate, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, accumulate);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_scale_rows(twog_rows_t x, const float* s, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, x, s,
                       rows, cols);
    TWOG_CHECK_LAUNCH();
    return 0;
}
