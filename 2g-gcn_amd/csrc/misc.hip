// Small HBM-bound kernels of the 2G-GCN hot path: ReLU backward, row add, column sums (bias gradients), label-head
// log-softmax + permuted store, reorder_hidden_states, filter_soft_decisions, fused Adam.
// All are one-pass streaming kernels with lane-contiguous (coalesced) access along the feature dimension.
#include "twog_common.h"

namespace {

__global__ __launch_bounds__(256) void relu_bwd_kernel(twog_rows_t dy, twog_rows_t y, twog_rows_t dx, int rows,
                                                       int cols) {
    const int c4 = cols >> 2;  // host guarantees cols % 4 == 0 and 16B alignment when vec != 0
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * c4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4), c = (int)(i - (int64_t)r * c4) * 4;
        const float4 g = *reinterpret_cast<const float4*>(twog_row_ptr(dy, r) + c);
        const float4 v = *reinterpret_cast<const float4*>(twog_row_ptr(y, r) + c);
        float4 o;
        o.x = v.x > 0.f ? g.x : 0.f;
        o.y = v.y > 0.f ? g.y : 0.f;
        o.z = v.z > 0.f ? g.z : 0.f;
        o.w = v.w > 0.f ? g.w : 0.f;
        *reinterpret_cast<float4*>(twog_row_ptr(dx, r) + c) = o;
    }
}

__global__ __launch_bounds__(256) void relu_bwd_scalar_kernel(twog_rows_t dy, twog_rows_t y, twog_rows_t dx, int rows,
                                                              int cols) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * cols;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        twog_row_ptr(dx, r)[c] = twog_row_ptr(y, r)[c] > 0.f ? twog_row_ptr(dy, r)[c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void add_rows_kernel(twog_rows_t src, twog_rows_t dst, int rows, int cols) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)rows * cols;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        twog_row_ptr(dst, r)[c] += twog_row_ptr(src, r)[c];
    }
}

// Several small row-wise operations in one launch (blockIdx.y = operation): what the host-composed general segment loop
// runs between its grouped GEMMs, once per dependency level instead of once per relation and direction.
constexpr int MAXROWOPS = 16;
struct RowOpBatch { twog_rowop_t op[MAXROWOPS]; };
__global__ __launch_bounds__(256) void rowops_kernel(const RowOpBatch G) {
    const twog_rowop_t& o = G.op[blockIdx.y];
    const int cols = o.cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)o.rows * cols;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        float* d = twog_row_ptr(o.dst, r) + c;
        if (o.kind == TWOG_ROWOP_RELU_BWD) *d = twog_row_ptr(o.b, r)[c] > 0.f ? twog_row_ptr(o.a, r)[c] : 0.f;
        else if (o.kind == TWOG_ROWOP_ADD) *d += twog_row_ptr(o.a, r)[c];
        else *d = fmaf(o.s[r], o.v[c], *d);
    }
}

// logits [(b,t,e)][C] -> out [b][C][t][e] = log_softmax over C        (vhoi/models.py:909-917)
__global__ __launch_bounds__(256) void lsm_permute_fwd_kernel(const float* logits, float* out, int bs, int T, int E,
                                                              int C) {
    const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t n_rows = (int64_t)bs * T * E;
    if (row >= n_rows) return;
    const float* l = logits + row * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, l[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(l[c] - m);
    const float lse = m + logf(s);
    const int64_t b = row / ((int64_t)T * E), te = row - b * (int64_t)T * E;
    float* o = out + b * (int64_t)C * T * E + te;
    for (int c = 0; c < C; ++c) o[(int64_t)c * T * E] = l[c] - lse;
}

// dlogits[row][c] = dout[b][c][t][e] - exp(out[b][c][t][e]) * sum_c' dout[b][c'][t][e]
__global__ __launch_bounds__(256) void lsm_permute_bwd_kernel(const float* out, const float* dout, float* dlogits,
                                                              int bs, int T, int E, int C) {
    const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t n_rows = (int64_t)bs * T * E;
    if (row >= n_rows) return;
    const int64_t b = row / ((int64_t)T * E), te = row - b * (int64_t)T * E;
    const int64_t base = b * (int64_t)C * T * E + te, cs = (int64_t)T * E;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dout[base + c * cs];
    float* d = dlogits + row * C;
    for (int c = 0; c < C; ++c) d[c] = dout[base + c * cs] - expf(out[base + c * cs]) * s;
}

// reorder_hidden_states (vhoi/models.py:1567-1586): one block per (clip, entity); idx[t] = first end frame >= t
__global__ __launch_bounds__(256) void reorder_kernel(const float* src, const float* gate, float* dst, int T, int E,
                                                      int cols, int backward) {
    extern __shared__ int idx[];  // [T] end-frame index, preceded in time by the staged gate flags
    const int b = blockIdx.x / E, e = blockIdx.x - b * E;
    // all gate values of this (clip, entity) in one parallel load, then a serial suffix scan over LDS
    for (int t = threadIdx.x; t < T; t += blockDim.x) idx[t] = gate[((int64_t)b * T + t) * E + e] != 0.f ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int nxt = -1;
        for (int t = T - 1; t >= 0; --t) {
            if (idx[t]) nxt = t;
            idx[t] = nxt >= 0 ? nxt : t;
        }
    }
    __syncthreads();
    const int64_t rs = (int64_t)E * cols;  // stride between frames of this (b, e)
    const float* s = src + ((int64_t)b * T * E + e) * cols;
    float* d = dst + ((int64_t)b * T * E + e) * cols;
    // blockIdx.y: a chunk of the columns (whole float4 groups), so that a small batch -- 8 clips x 4 entities = 32 (clip,
    // entity) pairs -- still spreads over the chip (round 5: 76 -> ~20 us per launch at 8 clips; every chunk's block
    // repeats the cheap scan)
    const int nchunk = gridDim.y, chunk = blockIdx.y;
    if (!backward) {
        if ((cols & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
            const int c4 = cols >> 2, per = (c4 + nchunk - 1) / nchunk, q0 = chunk * per, q1 = min(c4, q0 + per), w = max(q1 - q0, 0);
            for (int i = threadIdx.x; i < T * w; i += blockDim.x) {
                const int t = i / w, c = (q0 + i - t * w) * 4;
                *reinterpret_cast<float4*>(d + t * rs + c) = *reinterpret_cast<const float4*>(s + idx[t] * rs + c);
            }
        } else {
            const int per = (cols + nchunk - 1) / nchunk, c0 = chunk * per, c1 = min(cols, c0 + per), w = max(c1 - c0, 0);
            for (int i = threadIdx.x; i < T * w; i += blockDim.x) {
                const int t = i / w, c = c0 + i - t * w;
                d[t * rs + c] = s[idx[t] * rs + c];
            }
        }
    } else {
        // dhx[s] = sum of dout[t] over the frames t mapped to s (a contiguous run ending at s)
        const int per = (cols + nchunk - 1) / nchunk, c0 = chunk * per, c1 = min(cols, c0 + per);
        for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
            float acc = 0.f;
            for (int t = 0; t < T; ++t) {
                acc += s[t * rs + c];
                if (idx[t] == t) {
                    d[t * rs + c] = acc;
                    acc = 0.f;
                } else {
                    d[t * rs + c] = 0.f;
                }
            }
        }
    }
}

// filter_soft_decisions (vhoi/models.py:1637-1664). soft/hard/grad_mask: [bs][T][E]
__global__ __launch_bounds__(256) void filter_kernel(const float* soft, float* hard, float* gmask, int bs, int T, int E,
                                                     float thr) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)bs * T * E) return;
    const int e = (int)(i % E);
    const int t = (int)((i / E) % T);
    (void)e;
    const float u = soft[i];
    const float um1 = t > 0 ? soft[i - E] : 0.f;
    const float up1 = t + 1 < T ? soft[i + E] : 0.f;
    const bool cond = (u > um1) && (u > up1) && (u >= thr);
    // value: cond ? 1 : 0.   gradient wrt soft (straight-through + clamp(max=0) semantics): cond or u < thr -> 1
    hard[i] = cond ? 1.f : 0.f;
    gmask[i] = (cond || !(u >= thr)) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2s, float gscale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        if (wd != 0.f) gi += wd * p[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
    }
}

inline int grid_for(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

inline bool rows_vec_ok(const twog_rows_t& m) {
    return (reinterpret_cast<uintptr_t>(m.ptr) % 16 == 0) && (m.ld_outer % 4 == 0) &&
           (m.inner <= 1 || m.ld_inner % 4 == 0);
}

}  // namespace

extern "C" int twog_relu_bwd(twog_rows_t dy, twog_rows_t y, twog_rows_t dx, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (cols % 4 == 0 && rows_vec_ok(dy) && rows_vec_ok(y) && rows_vec_ok(dx))
        hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((int64_t)rows * cols / 4)), dim3(256), 0, st, dy, y, dx, rows,
                           cols);
    else
        hipLaunchKernelGGL(relu_bwd_scalar_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, st, dy, y, dx,
                           rows, cols);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_rowops(const twog_rowop_t* ops, int n_ops, void* stream) {
    for (int i = 0; i < n_ops; ++i) {
        const twog_rowop_t& o = ops[i];
        if (o.kind < TWOG_ROWOP_RELU_BWD || o.kind > TWOG_ROWOP_RANK1 || o.rows < 0 || o.cols < 0) return -2;
        if (o.rows == 0 || o.cols == 0) continue;
        if (!o.dst.ptr || (o.kind == TWOG_ROWOP_RANK1 ? (!o.s || !o.v) : !o.a.ptr) ||
            (o.kind == TWOG_ROWOP_RELU_BWD && !o.b.ptr))
            return -2;
    }
    for (int done = 0; done < n_ops; done += MAXROWOPS) {
        RowOpBatch G;
        const int m = n_ops - done < MAXROWOPS ? n_ops - done : MAXROWOPS;
        int64_t most = 0;
        for (int i = 0; i < m; ++i) {
            G.op[i] = ops[done + i];
            const int64_t n = (int64_t)G.op[i].rows * G.op[i].cols;
            if (n > most) most = n;
        }
        if (most == 0) continue;
        hipLaunchKernelGGL(rowops_kernel, dim3(grid_for(most, 256, 1024), m), dim3(256), 0, (hipStream_t)stream, G);
        TWOG_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int twog_add_rows(twog_rows_t src, twog_rows_t dst, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, rows, cols);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_logsoftmax_permute_fwd(const float* logits, float* out, int bs, int T, int E, int C,
                                           void* stream) {
    const int64_t n = (int64_t)bs * T * E;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lsm_permute_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, out, bs, T, E, C);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_logsoftmax_permute_bwd(const float* out, const float* dout, float* dlogits, int bs, int T, int E,
                                           int C, void* stream) {
    const int64_t n = (int64_t)bs * T * E;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lsm_permute_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       out, dout, dlogits, bs, T, E, C);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// column chunks per (clip, entity) so that the launch has ~512 blocks; a chunk keeps at least 64 columns
static int reorder_chunks(int pairs, int cols) {
    int n = (512 + pairs - 1) / pairs;
    const int most = cols / 64 > 0 ? cols / 64 : 1;
    if (n > most) n = most;
    return n < 1 ? 1 : n;
}

extern "C" int twog_reorder_fwd(const float* hx, const float* gate, float* out, int bs, int T, int E, int cols,
                                void* stream) {
    if (bs * E <= 0) return 0;
    hipLaunchKernelGGL(reorder_kernel, dim3(bs * E, reorder_chunks(bs * E, cols)), dim3(256), T * sizeof(int), (hipStream_t)stream,
                       hx, gate, out, T, E, cols, 0);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_reorder_bwd(const float* dout, const float* gate, float* dhx, int bs, int T, int E, int cols,
                                void* stream) {
    if (bs * E <= 0) return 0;
    hipLaunchKernelGGL(reorder_kernel, dim3(bs * E, reorder_chunks(bs * E, cols)), dim3(256), T * sizeof(int), (hipStream_t)stream,
                       dout, gate, dhx, T, E, cols, 1);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_filter_fwd(const float* soft, float* hard, float* grad_mask, int bs, int T, int E, float threshold,
                               void* stream) {
    const int64_t n = (int64_t)bs * T * E;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, soft, hard,
                       grad_mask, bs, T, E, threshold);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// Zero fill of caller-owned device memory (16-byte stores; head / tail bytes one by one): what torch.zeros / Tensor.zero_()
// did on the step's path until round 4 -- 19 ATen fill launches per step; the host now clears each group of buffers it
// allocates (one torch.empty for the group) with ONE launch of this kernel.
__global__ __launch_bounds__(256) void fill_zero_kernel(char* p, size_t nbytes) {
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
    const size_t h = head < nbytes ? head : nbytes;
    const size_t body = (nbytes - h) / 16;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < body; i += (size_t)gridDim.x * 256) q[i] = z;
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < h; i += 256) p[i] = 0;
        const size_t t0 = h + body * 16;
        for (size_t i = t0 + threadIdx.x; i < nbytes; i += 256) p[i] = 0;
    }
}

extern "C" int twog_fill_zero(void* p, size_t nbytes, void* stream) {
    if (nbytes == 0) return 0;
    if (!p) return -2;
    const size_t body = nbytes / 16;
    const size_t want = (body + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    hipLaunchKernelGGL(fill_zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<char*>(p), nbytes);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// Copies of contiguous fp32 blocks, all in ONE launch (blockIdx.y = block): the packed forms of weights that the time loops
// read as one operand (the segment level's sender MLPs, w_smsg_* / b_smsg_* of twog_segrnn_t) are rebuilt from the
// parameters by this launch at EVERY forward call -- nothing derived from a weight outlives the call that derived it, so no
// write to a parameter (optimizer, load_state_dict, p.data.mul_(), a kernel on a flat buffer) can leave it stale.
struct CopyBatch { twog_copy_t c[TWOG_COPY_MAX]; };
__global__ __launch_bounds__(256) void copy_blocks_kernel(const CopyBatch G) {
    const twog_copy_t& c = G.c[blockIdx.y];
    const int64_t n = c.n;
    const bool vec = ((reinterpret_cast<uintptr_t>(c.src) | reinterpret_cast<uintptr_t>(c.dst)) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const float4* s4 = reinterpret_cast<const float4*>(c.src);
    float4* d4 = reinterpret_cast<float4*>(c.dst);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) d4[i] = s4[i];
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c.dst[i] = c.src[i];
}

extern "C" int twog_copy_blocks(const twog_copy_t* blocks, int n_blocks, void* stream) {
    if (n_blocks <= 0) return 0;
    if (n_blocks > TWOG_COPY_MAX || !blocks) return -2;
    CopyBatch G;
    int64_t most = 0;
    for (int i = 0; i < n_blocks; ++i) {
        G.c[i] = blocks[i];
        if (blocks[i].n < 0 || (blocks[i].n > 0 && (!blocks[i].src || !blocks[i].dst))) return -2;
        if (blocks[i].n > most) most = blocks[i].n;
    }
    if (most == 0) return 0;
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(grid_for((most + 3) / 4, 256, 512), n_blocks), dim3(256), 0,
                       (hipStream_t)stream, G);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// Diagnostics: holds compute units for a while -- n_blocks workgroups of 256 threads with lds_bytes of LDS each spin until
// `usec` microseconds have passed on the device's wall clock. Tests use it as the "other tenant" a persistent launch
// (gru_persist.hip, seg_persist.hip) must survive: with one block per compute unit and more than half of the LDS each, no
// other workgroup that asks for LDS of its own becomes resident until these have left.
__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, unsigned* sink) {
    extern __shared__ char lds_hold[];
    lds_hold[threadIdx.x] = (char)threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds_hold[(threadIdx.x + 1) & 255] == 77 && sink) sink[0] = 1;   // keeps the LDS request alive
}

extern "C" int twog_debug_occupy(int n_blocks, int lds_bytes, int usec, void* stream) {
    if (n_blocks <= 0 || usec <= 0) return 0;
    if (lds_bytes < 256 || lds_bytes > 160 * 1024) return -2;
    static std::atomic<uint32_t> done{0};
    twog_allow_dynamic_lds(occupy_kernel, 160 * 1024, done);
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    const long long ticks = (long long)usec * khz / 1000;
    hipLaunchKernelGGL(occupy_kernel, dim3(n_blocks), dim3(256), lds_bytes, (hipStream_t)stream, ticks, (unsigned*)nullptr);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// Loud failure without a host round trip: if any of the (host-pinned, device-readable) error words of the pass's persistent
// launches is non-zero, every output tensor of the pass is overwritten with NaN. Enqueued behind the asynchronous copies that
// fill those words, so the caller need not wait for them before it goes on to enqueue the backward pass (kernels.py).
__global__ __launch_bounds__(256) void guard_outputs_kernel(const twog_guard_t g) {
    bool bad = false;
    for (int i = 0; i < g.n_words; ++i) bad = bad || (__builtin_nontemporal_load(g.words[i]) != 0);
    if (!bad) return;
    float* o = g.out[blockIdx.x];
    const float nan = __builtin_nanf("");
    for (int64_t i = threadIdx.x; i < g.n[blockIdx.x]; i += blockDim.x) o[i] = nan;
}
extern "C" int twog_guard_outputs(const twog_guard_t* g, void* stream) {
    if (!g || g->n_words < 0 || g->n_words > TWOG_GUARD_MAX || g->n_out < 0 || g->n_out > TWOG_GUARD_MAX) return -2;
    if (g->n_words == 0 || g->n_out == 0) return 0;
    hipLaunchKernelGGL(guard_outputs_kernel, dim3(g->n_out), dim3(256), 0, (hipStream_t)stream, *g);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// A stream whose kernels may only use n_cus compute units: bit i of the mask is CU i / n_xcd of XCD i % n_xcd (the driver deals
// the bits round-robin over the XCDs), so the low n_cus bits are n_cus / 8 CUs on each of the eight XCDs -- the share a
// launch-per-step recurrence leaves idle (ops.tggcn_backward runs weight-gradient GEMMs there beside the BiGRU backward chain).
// Returns 0 and the stream, or a negative code when the runtime refuses (the caller then uses an ordinary stream).
extern "C" int twog_stream_create_masked(int n_cus, void** stream_out) {
    if (!stream_out || n_cus <= 0) return -2;
    int dev = 0, total = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || total <= 0) return -2;
    if (n_cus > total) n_cus = total;
    uint32_t mask[16] = {0};
    const int words = (total + 31) / 32;
    if (words > 16) return -2;
    for (int i = 0; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); return -3; }
    *stream_out = st;
    return 0;
}
extern "C" int twog_stream_create_low_priority(void** stream_out) {
    if (!stream_out) return -2;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); return -3; }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); return -3; }
    *stream_out = st;
    return 0;
}
extern "C" int twog_stream_destroy(void* stream) {
    return stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess ? -1 : 0;
}

extern "C" int twog_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream) {
    if (n <= 0) return 0;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    TWOG_CHECK_LAUNCH();
    return 0;
}
