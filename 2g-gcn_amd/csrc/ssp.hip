// Sender-side projection of aggregated messages ("project, then aggregate").
//
// The segment-level GRUCells of the objects read xx_os = cat[h_f, m_ho, m_so, m_oo] (vhoi/models.py:748) through
// W_ih, where m_ho[k] = mask_k * sum_h att[k][h] * msg_h (humans -> object k, :1099-1143, :720) and
// m_so[k] = mask_k * msg_s (the geometry node's message, :1384-1429, :729). By linearity
//        m_ho[k] W^T = mask_k * sum_h att[k][h] * (msg_h W^T)          m_so[k] W^T = mask_k * (msg_s W^T)
// so the projection can run over the SENDERS' rows (H humans + 1 geometry node per frame) instead of the O receivers'
// rows: at the BASELINE shape (H = 2, O = 8) that removes 40 % of the largest GEMM of the step, forward and both
// backward forms. These kernels are the cheap, HBM-bound glue: the weighted scatter of the projected sender rows into
// the receivers' pre-activations, and its transpose for the backward pass.
#include "twog_common.h"

namespace {

// gi[(inst, k)][c] += mask[clip][k] * ( sum_h att[inst][att_off + k*H + h] * ph[(inst, h)][c] + ps[inst][c] )
__global__ __launch_bounds__(256) void ssp_fwd_kernel(float* gi, const float* ph, const float* ps, const float* att,
                                                      const float* mask, int n_inst, int inst_per_clip, int H, int O,
                                                      int cols, int natt, int att_off) {
    const int inst = blockIdx.x;
    const int c4 = cols >> 2;
    const int clip = inst / inst_per_clip;
    for (int i = threadIdx.x; i < c4; i += blockDim.x) {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ps) s4 = reinterpret_cast<const float4*>(ps + (int64_t)inst * cols)[i];
        float4 p4[4];
        for (int h = 0; h < H && h < 4; ++h)
            p4[h] = ph ? reinterpret_cast<const float4*>(ph + ((int64_t)inst * H + h) * cols)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < O; ++k) {
            const float m = mask ? mask[(int64_t)clip * O + k] : 1.f;
            if (m == 0.f) continue;
            float4 a = s4;
            if (ph)
                for (int h = 0; h < H && h < 4; ++h) {
                    const float w = att[(int64_t)inst * natt + att_off + k * H + h];
                    a.x = fmaf(w, p4[h].x, a.x); a.y = fmaf(w, p4[h].y, a.y);
                    a.z = fmaf(w, p4[h].z, a.z); a.w = fmaf(w, p4[h].w, a.w);
                }
            float4* g = reinterpret_cast<float4*>(gi + ((int64_t)inst * O + k) * cols) + i;
            float4 v = *g;
            v.x = fmaf(m, a.x, v.x); v.y = fmaf(m, a.y, v.y); v.z = fmaf(m, a.z, v.z); v.w = fmaf(m, a.w, v.w);
            *g = v;
        }
    }
}

// qh[(inst, h)][c] = sum_k mask_k att[k][h] dgi[(inst, k)][c] ;  qs[inst][c] = sum_k mask_k dgi[(inst, k)][c]
// dw[inst][att_off + k*H + h] = mask_k <dgi[(inst, k)], ph[(inst, h)]>     (gradient wrt the attention weights)
__global__ __launch_bounds__(256) void ssp_bwd_kernel(const float* dgi, const float* ph, const float* att,
                                                      const float* mask, float* qh, float* qs, float* dw, int n_inst,
                                                      int inst_per_clip, int H, int O, int cols, int natt, int att_off) {
    __shared__ float red[4][16 * 4];   // [wave][k * H + h] partial dots, O <= 16, H <= 4
    const int inst = blockIdx.x;
    const int c4 = cols >> 2;
    const int clip = inst / inst_per_clip;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float dot[16 * 4];
#pragma unroll
    for (int i = 0; i < 64; ++i) dot[i] = 0.f;
    for (int i = threadIdx.x; i < c4; i += blockDim.x) {
        float4 p4[4], q4[4], s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            q4[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            p4[h] = (ph && h < H) ? reinterpret_cast<const float4*>(ph + ((int64_t)inst * H + h) * cols)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k >= O) break;
            const float m = mask ? mask[(int64_t)clip * O + k] : 1.f;
            if (m == 0.f) continue;
            const float4 g = reinterpret_cast<const float4*>(dgi + ((int64_t)inst * O + k) * cols)[i];
            s4.x += g.x; s4.y += g.y; s4.z += g.z; s4.w += g.w;
            if (ph) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if (h >= H) break;
                    const float w = att[(int64_t)inst * natt + att_off + k * H + h];
                    q4[h].x = fmaf(w, g.x, q4[h].x); q4[h].y = fmaf(w, g.y, q4[h].y);
                    q4[h].z = fmaf(w, g.z, q4[h].z); q4[h].w = fmaf(w, g.w, q4[h].w);
                    dot[k * 4 + h] += g.x * p4[h].x + g.y * p4[h].y + g.z * p4[h].z + g.w * p4[h].w;
                }
            }
        }
        if (qs) reinterpret_cast<float4*>(qs + (int64_t)inst * cols)[i] = s4;
        if (qh)
            for (int h = 0; h < H && h < 4; ++h) reinterpret_cast<float4*>(qh + ((int64_t)inst * H + h) * cols)[i] = q4[h];
    }
    if (!dw || !ph) return;
    // ordered reduction of the per-thread partial dots: wave shuffles, then the four waves in fixed order
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k >= O) break;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (h >= H) break;
            const float v = wave_sum(dot[k * 4 + h]);
            if (lane == 0) red[wv][k * 4 + h] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < O * H) {
        const int k = threadIdx.x / H, h = threadIdx.x - k * H;
        const float m = mask ? mask[(int64_t)clip * O + k] : 1.f;
        const float t = (red[0][k * 4 + h] + red[1][k * 4 + h]) + (red[2][k * 4 + h] + red[3][k * 4 + h]);
        dw[(int64_t)inst * natt + att_off + k * H + h] = m != 0.f ? t : 0.f;
    }
}

// gather only, arbitrary placement of the weights and of the gradient rows (the segment level keeps its attention
// weights [time][clip][natt] and the two directions side by side in the d_gi rows):
//   qh[(inst, h)][c] = sum_k att(inst)[att_off + k*H + h] * dgi[(inst*O + k) * dgi_ld + c]
__global__ __launch_bounds__(256) void ssp_gather_kernel(const float* dgi, int64_t dgi_ld, const float* att,
                                                         int64_t att_ld_clip, int64_t att_ld_frame, int att_off,
                                                         float* qh, int inst_per_clip, int H, int O, int cols) {
    const int inst = blockIdx.x;
    const int clip = inst / inst_per_clip, frame = inst - clip * inst_per_clip;
    const float* w = att + clip * att_ld_clip + frame * att_ld_frame + att_off;
    const int c4 = cols >> 2;
    for (int i = threadIdx.x; i < c4; i += blockDim.x) {
        float4 q4[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) q4[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < O; ++k) {
            const float4 g = reinterpret_cast<const float4*>(dgi + ((int64_t)inst * O + k) * dgi_ld)[i];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (h >= H) break;
                const float wk = w[k * H + h];
                q4[h].x = fmaf(wk, g.x, q4[h].x); q4[h].y = fmaf(wk, g.y, q4[h].y);
                q4[h].z = fmaf(wk, g.z, q4[h].z); q4[h].w = fmaf(wk, g.w, q4[h].w);
            }
        }
        for (int h = 0; h < H && h < 4; ++h) reinterpret_cast<float4*>(qh + ((int64_t)inst * H + h) * cols)[i] = q4[h];
    }
}

}  // namespace

extern "C" int twog_ssp_gather(const float* dgi, int64_t dgi_ld, const float* att, int64_t att_ld_clip,
                               int64_t att_ld_frame, int att_off, float* qh, int n_inst, int inst_per_clip, int H, int O,
                               int cols, void* stream) {
    if (n_inst <= 0 || O <= 0 || H <= 0) return 0;
    if ((cols & 3) || (dgi_ld & 3) || H > 4 || O > 16 || inst_per_clip <= 0 || !att || !qh ||
        (reinterpret_cast<uintptr_t>(dgi) & 15))
        return -2;
    hipLaunchKernelGGL(ssp_gather_kernel, dim3(n_inst), dim3(256), 0, (hipStream_t)stream, dgi, dgi_ld, att, att_ld_clip,
                       att_ld_frame, att_off, qh, inst_per_clip, H, O, cols);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_ssp_fwd(float* gi, const float* ph, const float* ps, const float* att, const float* mask, int n_inst,
                            int inst_per_clip, int H, int O, int cols, int natt, int att_off, void* stream) {
    if (n_inst <= 0 || O <= 0) return 0;
    if ((cols & 3) || H > 4 || O > 16 || H < 0 || (ph && !att) || inst_per_clip <= 0) return -2;
    if (!ph && !ps) return 0;
    hipLaunchKernelGGL(ssp_fwd_kernel, dim3(n_inst), dim3(256), 0, (hipStream_t)stream, gi, ph, ps, att, mask, n_inst,
                       inst_per_clip, H, O, cols, natt, att_off);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_ssp_bwd(const float* dgi, const float* ph, const float* att, const float* mask, float* qh, float* qs,
                            float* dw, int n_inst, int inst_per_clip, int H, int O, int cols, int natt, int att_off,
                            void* stream) {
    if (n_inst <= 0 || O <= 0) return 0;
    if ((cols & 3) || H > 4 || O > 16 || H < 0 || (ph && !att) || inst_per_clip <= 0) return -2;
    hipLaunchKernelGGL(ssp_bwd_kernel, dim3(n_inst), dim3(256), 0, (hipStream_t)stream, dgi, ph, att, mask, qh, qs, dw,
                       n_inst, inst_per_clip, H, O, cols, natt, att_off);
    TWOG_CHECK_LAUNCH();
    return 0;
}
