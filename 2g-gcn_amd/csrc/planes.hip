// Pre-split weights: the three exact bf16 planes (h, m, l) of an fp32 weight matrix, written once per optimizer step in
// the layouts the X3 GEMM kernels (gemm_f32.hip) copy straight into LDS with buffer_load ... lds. See include/twog_gcn.h
// (twog_weight_planes_build) for the contract and the layouts. The split is the one of gemm_f32.hip::split3 -- truncation,
// exact: h = top 8 significant bits, m = top 8 of x - h, l = x - h - m -- so a kernel fed from planes computes bit for bit
// what the same kernel computes when it splits B itself.
#include "twog_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_hi16(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302); }

__device__ __forceinline__ void split3(const f32x4 v, uint2& ph, uint2& pm, uint2& pl) {
    uint32_t x[4], r1[4], r2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f0 = v[i];
        x[i] = __float_as_uint(f0);
        const float f1 = f0 - __uint_as_float(x[i] & 0xffff0000u);
        r1[i] = __float_as_uint(f1);
        r2[i] = __float_as_uint(f1 - __uint_as_float(r1[i] & 0xffff0000u));
    }
    ph = make_uint2(pack_hi16(x[0], x[1]), pack_hi16(x[2], x[3]));
    pm = make_uint2(pack_hi16(r1[0], r1[1]), pack_hi16(r1[2], r1[3]));
    pl = make_uint2(pack_hi16(r2[0], r2[1]), pack_hi16(r2[2], r2[3]));
}

// RM: [3][cols/16][rows_pad][16] bf16. One thread per (k-tile, row, quad of k): consecutive threads write consecutive
// 8-byte pieces of a plane (row-fastest inside a k-tile), i.e. whole lines; the reads are 16-byte pieces of 64-byte row
// segments.
__global__ __launch_bounds__(256) void planes_rm_kernel(const float* __restrict__ w, int rows, int cols, int64_t ld,
                                                        int rows_pad, char* __restrict__ out) {
    const int64_t n_kt = cols / 16;
    const int64_t total = n_kt * rows_pad * 4;
    const int64_t plane = n_kt * rows_pad * 32;   // bytes
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3);
        const int64_t rn = i >> 2;
        const int n = (int)(rn % rows_pad);
        const int64_t kt = rn / rows_pad;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < rows) v = *reinterpret_cast<const f32x4*>(w + (int64_t)n * ld + kt * 16 + q * 4);
        uint2 ph, pm, pl;
        split3(v, ph, pm, pl);
        char* o = out + (kt * rows_pad + n) * 32 + q * 8;
        *reinterpret_cast<uint2*>(o) = ph;
        *reinterpret_cast<uint2*>(o + plane) = pm;
        *reinterpret_cast<uint2*>(o + 2 * plane) = pl;
    }
}

// KM: [3][rows][cols_pad] bf16: the matrix as it stands, one thread per (row, quad of columns)
__global__ __launch_bounds__(256) void planes_km_kernel(const float* __restrict__ w, int rows, int cols, int64_t ld,
                                                        int cols_pad, char* __restrict__ out) {
    const int64_t qpr = cols_pad / 4;
    const int64_t total = (int64_t)rows * qpr;
    const int64_t plane = (int64_t)rows * cols_pad * 2;
    const bool vec = (cols & 3) == 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / qpr;
        const int c = (int)(i - k * qpr) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* src = w + k * ld + c;
        if (vec) {
            if (c < cols) v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j < cols) v[j] = src[j];
        }
        uint2 ph, pm, pl;
        split3(v, ph, pm, pl);
        char* o = out + (k * cols_pad + c) * 2;
        *reinterpret_cast<uint2*>(o) = ph;
        *reinterpret_cast<uint2*>(o + plane) = pm;
        *reinterpret_cast<uint2*>(o + 2 * plane) = pl;
    }
}

// KF: the matrix read as a k-major B operand ([K = rows][N = cols]) in MFMA FRAGMENT order: [3][rows / 16][cols_pad / 32][64 lanes][8]
// bf16 -- lane l = n + 32 h of the (k-step ks, column block nb) piece holds W[16 ks + 8 h + j][32 nb + n], j = 0..7: the B
// fragment of v_mfma_f32_32x32x16_bf16, so a wave loads its fragment with ONE 16-byte load per lane, 1 KB contiguous.
__global__ __launch_bounds__(256) void planes_kf_kernel(const float* __restrict__ w, int rows, int cols, int64_t ld,
                                                        int cols_pad, char* __restrict__ out) {
    const int64_t nb = cols_pad / 32, nks = rows / 16;
    const int64_t total = nks * nb * 64;
    const int64_t plane = nks * nb * 1024;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), n = lane & 31, h = lane >> 5;
        const int64_t piece = i >> 6, b = piece % nb, ks = piece / nb;
        const int c = (int)(b * 32 + n);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = c < cols ? w[(ks * 16 + 8 * h + j) * ld + c] : 0.f;
        uint2 ph0, pm0, pl0, ph1, pm1, pl1;
        split3(f32x4{v[0], v[1], v[2], v[3]}, ph0, pm0, pl0);
        split3(f32x4{v[4], v[5], v[6], v[7]}, ph1, pm1, pl1);
        char* o = out + i * 16;
        *reinterpret_cast<uint4*>(o) = make_uint4(ph0.x, ph0.y, ph1.x, ph1.y);
        *reinterpret_cast<uint4*>(o + plane) = make_uint4(pm0.x, pm0.y, pm1.x, pm1.y);
        *reinterpret_cast<uint4*>(o + 2 * plane) = make_uint4(pl0.x, pl0.y, pl1.x, pl1.y);
    }
}

inline int pad128(int v) { return (v + 127) / 128 * 128; }

}  // namespace

extern "C" size_t twog_weight_planes_bytes(int rows, int cols, int kind) {
    if (rows <= 0 || cols <= 0) return 0;
    if (kind == TWOG_PLANES_RM) return (cols % 16) ? 0 : (size_t)3 * (cols / 16) * pad128(rows) * 32;
    if (kind == TWOG_PLANES_KM || kind == TWOG_PLANES_KF) return (rows % 16) ? 0 : (size_t)3 * rows * pad128(cols) * 2;
    return 0;
}

extern "C" int twog_weight_planes_build(const float* w, int rows, int cols, int64_t ld, int kind, void* planes, void* stream) {
    if (!w || !planes || rows <= 0 || cols <= 0 || ld < cols) return -2;
    if ((reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return -2;
    if (twog_weight_planes_bytes(rows, cols, kind) == 0) return -2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (kind == TWOG_PLANES_RM) {
        if (ld & 3) return -2;
        const int rp = pad128(rows);
        const int64_t total = (int64_t)(cols / 16) * rp * 4;
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(planes_rm_kernel, dim3(blocks), dim3(256), 0, st, w, rows, cols, ld, rp, reinterpret_cast<char*>(planes));
    } else if (kind == TWOG_PLANES_KF) {
        const int cp = pad128(cols);
        const int64_t total = (int64_t)(rows / 16) * (cp / 32) * 64;
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(planes_kf_kernel, dim3(blocks), dim3(256), 0, st, w, rows, cols, ld, cp, reinterpret_cast<char*>(planes));
    } else {
        if ((cols & 3) == 0 && (ld & 3)) return -2;
        const int cp = pad128(cols);
        const int64_t total = (int64_t)rows * (cp / 4);
        const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(planes_km_kernel, dim3(blocks), dim3(256), 0, st, w, rows, cols, ld, cp, reinterpret_cast<char*>(planes));
    }
    TWOG_CHECK_LAUNCH();
    return 0;
}
