// Multi-task loss of the 2G-GCN training step, all terms in one launch.
//
// Reference: vhoi/losses.py:8-70 (select_loss builds the term list: budget, BCE-with-ignore, and 4 or 8 NLL terms),
// pyrutils/torch/losses.py:7-51 (binary_cross_entropy_loss, budget_loss, multi_task_loss), F.nll_loss(ignore_index=-1,
// reduction='mean'). The reference runs 6-12 small torch kernels per term and one host sync per BCE / budget term
// (mask.sum().item()); here every term is a column of a (blocks, terms) grid: lane-contiguous reads, fp64 partial
// sums reduced in a fixed order (bit-reproducible), the per-term {sum, count} kept on the device for the backward
// launch, which writes d(input) of every term in one pass (for NLL: -w/count at the target class, 0 elsewhere).
#include "twog_common.h"

namespace {

struct Terms { twog_loss_t t[TWOG_LOSS_MAX_TERMS]; };

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void loss_partial_kernel(const Terms T, double* partials) {
    const twog_loss_t& L = T.t[blockIdx.y];
    const int64_t n = L.outer * L.inner;
    double sum = 0.0, cnt = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        if (L.kind == 0) {
            const int64_t tgt = reinterpret_cast<const int64_t*>(L.target)[e];
            if (tgt != (int64_t)L.ignore_value && tgt >= 0 && tgt < L.n_classes) {
                const int64_t o = e / L.inner, i = e - o * L.inner;
                sum -= (double)L.input[(o * L.n_classes + tgt) * L.inner + i];
                cnt += 1.0;
            }
        } else {
            const float t = reinterpret_cast<const float*>(L.target)[e];
            if (t != L.ignore_value) {
                const float x = L.input[e];
                if (L.kind == 1) sum -= (double)(t * fmaxf(logf(x), -100.f) + (1.f - t) * fmaxf(logf(1.f - x), -100.f));
                else sum += (double)x;
                cnt += 1.0;
            }
        }
    }
    __shared__ double red[2][4];
    sum = wave_sum_f64(sum);
    cnt = wave_sum_f64(cnt);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = sum; red[1][w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void loss_final_kernel(const Terms T, int n_terms, int n_blocks, const double* partials, double* stats,
                                  float* losses) {
    const int term = blockIdx.x * blockDim.x + threadIdx.x;
    if (term >= n_terms) return;
    double sum = 0.0, cnt = 0.0;
    for (int b = 0; b < n_blocks; ++b) {
        sum += partials[((int64_t)term * n_blocks + b) * 2];
        cnt += partials[((int64_t)term * n_blocks + b) * 2 + 1];
    }
    stats[term * 2] = sum;
    stats[term * 2 + 1] = cnt;
    const twog_loss_t& L = T.t[term];
    double v;
    if (L.kind == 0) v = sum / cnt;  // 0 / 0 -> NaN like torch's mean over an empty selection
    else v = cnt > 0.0 ? sum / cnt : 0.0;
    losses[term] = L.weight * (float)v;
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(const Terms T, const double* stats, const float* dlosses) {
    const twog_loss_t& L = T.t[blockIdx.y];
    if (!L.dinput) return;
    const int64_t n = L.outer * L.inner;
    const double cnt = stats[blockIdx.y * 2 + 1];
    const float g = cnt > 0.0 ? L.weight * dlosses[blockIdx.y] / (float)cnt : 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        if (L.kind == 0) {
            const int64_t tgt = reinterpret_cast<const int64_t*>(L.target)[e];
            const bool valid = tgt != (int64_t)L.ignore_value && tgt >= 0 && tgt < L.n_classes;
            const int64_t o = e / L.inner, i = e - o * L.inner;
            float* d = L.dinput + o * L.n_classes * L.inner + i;
            for (int c = 0; c < L.n_classes; ++c) d[(int64_t)c * L.inner] = (valid && c == tgt) ? -g : 0.f;
        } else {
            const float t = reinterpret_cast<const float*>(L.target)[e];
            float d = 0.f;
            if (t != L.ignore_value) {
                if (L.kind == 1) {
                    const float x = L.input[e];
                    d = g * (x - t) / fmaxf((1.f - x) * x, 1e-12f);  // torch's binary_cross_entropy backward
                } else {
                    d = g;
                }
            }
            L.dinput[e] = d;
        }
    }
}

inline bool terms_ok(const twog_loss_t* t, int n) {
    if (n < 0 || n > TWOG_LOSS_MAX_TERMS) return false;
    for (int i = 0; i < n; ++i) {
        if (t[i].kind < 0 || t[i].kind > 2 || t[i].outer < 0 || t[i].inner < 0) return false;
        if (t[i].kind == 0 && t[i].n_classes < 1) return false;
    }
    return true;
}

}  // namespace

extern "C" int twog_multitask_loss_fwd(const twog_loss_t* terms, int n_terms, double* partials, double* stats,
                                       float* losses, void* stream) {
    if (!terms_ok(terms, n_terms)) return -1;
    if (n_terms == 0) return 0;
    Terms T;
    for (int i = 0; i < n_terms; ++i) T.t[i] = terms[i];
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(TWOG_LOSS_BLOCKS, n_terms), dim3(256), 0, st, T, partials);
    TWOG_CHECK_LAUNCH();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, T, n_terms, TWOG_LOSS_BLOCKS, partials, stats, losses);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_multitask_loss_bwd(const twog_loss_t* terms, int n_terms, const double* stats, const float* dlosses,
                                       void* stream) {
    if (!terms_ok(terms, n_terms)) return -1;
    if (n_terms == 0) return 0;
    Terms T;
    for (int i = 0; i < n_terms; ++i) T.t[i] = terms[i];
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(TWOG_LOSS_BLOCKS * 4, n_terms), dim3(256), 0, (hipStream_t)stream, T, stats,
                       dlosses);
    TWOG_CHECK_LAUNCH();
    return 0;
}
