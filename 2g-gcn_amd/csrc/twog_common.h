// Shared device helpers for the gfx950 kernels of the 2G-GCN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/twog_gcn.h"

#define TWOG_WAVE 64

#define TWOG_CHECK_LAUNCH()                         \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return -(int)e_;      \
    } while (0)

// Raises a kernel's dynamic-LDS limit once per device (function attributes are per device; the flag word is one bit per
// device ordinal, so a process that drives several GPUs -- or several host threads -- stays correct).
template <class K>
inline void twog_allow_dynamic_lds(K kernel, int bytes, std::atomic<uint32_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const uint32_t bit = 1u << (dev & 31);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}

// library-internal (gemm_f32.hip): grouped C += A B launch whose epilogue runs the GRU gate backward of the next chain step
// chain_ws / chain_ws_bytes: the chain workspace of twog_gemm_f32_chain (NULL: the reduction is never split over workgroups)
int twog_internal_gemm_gate_bwd(const twog_gemm_t* problems, int n, const twog_gru_step_bwd_t* gates,
                                float* const* du_part, int dry_run, void* chain_ws, size_t chain_ws_bytes, void* stream);

// library-internal (gemm_f32.hip): one launch for the W_hh (and message) products of a forward chain step AND its gates
int twog_internal_gemm_gru_fwd(const twog_gemm_t* gh, const twog_gemm_t* gim, const twog_gru_step_t* steps, int n,
                               int dry_run, void* stream);

int twog_internal_gru_fwd_mode(void);   // TWOG_GRU_FWD_FUSION (part of the chains' hipGraph keys: it changes what is captured)

// address of row r in a twog_rows_t (see include/twog_gcn.h)
__device__ __forceinline__ int64_t twog_row_off(const twog_rows_t& m, int r) {
    if (m.inner <= 1) return (int64_t)r * m.ld_outer;
    const int o = r / m.inner;
    return (int64_t)o * m.ld_outer + (int64_t)(r - o * m.inner) * m.ld_inner;
}
__device__ __forceinline__ float* twog_row_ptr(const twog_rows_t& m, int r) { return m.ptr + twog_row_off(m, r); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x a multiple of 64, <= 1024; `red` is >= 16 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
