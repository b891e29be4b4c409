// Adjacency attention of the geometric-level GCN on the matrix cores, with the two similarity projections folded.
//
// Reference: compute_similarity + the adjacency product of Geo_gcn.forward (pyrutils/torch/models_gcn.py:86-100, :30-34):
//   theta = Wq x + bq, phi = Wk x + bk (1x1 convs 64 -> 128), S = softmax_j(theta_i . phi_j), Z = S X.
// theta_i . phi_j = x_i^T (Wq^T Wk) x_j + (Wk^T bq) . x_j + [terms that do not depend on j], and the softmax over j is
// invariant to the latter, so with M = Wq^T Wk (64x64) and d = Wk^T bq (64):
//   P = X M + 1 d^T,   S = softmax_j(P X^T),   Z = S X
// is the same function (the key bias bk has an identically zero gradient, as in the reference). That removes the
// (frames*N) x 256 theta/phi tensor (267 MB written and read back per C3 batch), its two projection GEMMs and 60 % of the
// attention FLOPs; M and d are 16 KB and live in LDS. Everything per frame is N <= 64 rows, so the products run on
// v_mfma_f32_16x16x4_f32 (exact fp32): lane l supplies A[row = l%16][k = l/16] and B[k = l/16][col = l%16], both read
// from LDS arrays whose contiguous dimension is k with a row stride = 4 (mod 16) words (conflict-free ds_read_b32:
// bank = 4*(l%16) + l/16 + const); the accumulator lane layout is C[row = 4*(l/16) + r][col = l%16].
// Operands needed in both orientations are kept twice in LDS (as [row][k] and transposed), written once at staging.
#include "twog_common.h"

typedef float f32x4m __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXN = 64;
constexpr int LDK = 68;  // stride of LDS arrays whose contiguous dimension is the 64-wide feature axis

__device__ __forceinline__ f32x4m mfma16(float a, float b, f32x4m c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc += A[rt-th row tile][0..K) * B[ct-th col tile][0..K)^T with both operands stored [row][k], k contiguous
__device__ __forceinline__ f32x4m tile_mm(const float* a, int lda, const float* b, int ldb, int ksteps, int i16, int g,
                                          f32x4m acc) {
    const float* pa = a + i16 * lda + g;
    const float* pb = b + i16 * ldb + g;
#pragma unroll 4
    for (int kk = 0; kk < ksteps; ++kk) acc = mfma16(pa[kk * 4], pb[kk * 4], acc);
    return acc;
}
// same with B stored [k][col] (col contiguous, row stride ldb): bank-conflicted reads, only used when LDS is short
__device__ __forceinline__ f32x4m tile_mm_bt(const float* a, int lda, const float* b, int ldb, int ksteps, int i16, int g,
                                             f32x4m acc) {
    const float* pa = a + i16 * lda + g;
    const float* pb = b + g * ldb + i16;
#pragma unroll 4
    for (int kk = 0; kk < ksteps; ++kk) acc = mfma16(pa[kk * 4], pb[kk * 4 * ldb], acc);
    return acc;
}

// stage a [N][64] row block: dst[r][c] (stride LDK) and, if dstT, dstT[c][r] (stride ldn); rows N..NP-1 are zeroed
__device__ __forceinline__ void stage_rows64(const float* src, int N, int NP, float* dst, float* dstT, int ldn) {
    for (int i = threadIdx.x; i < NP * 16; i += blockDim.x) {
        const int r = i >> 4, c = (i & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < N) v = *reinterpret_cast<const float4*>(src + (int64_t)r * 64 + c);
        *reinterpret_cast<float4*>(dst + r * LDK + c) = v;
        if (dstT) {
            dstT[(c + 0) * ldn + r] = v.x;
            dstT[(c + 1) * ldn + r] = v.y;
            dstT[(c + 2) * ldn + r] = v.z;
            dstT[(c + 3) * ldn + r] = v.w;
        }
    }
}

// The same in two halves, so that the NEXT frame's rows travel while this frame is computed: fetch_rows64 issues the loads of
// a [N][64] block into registers (1024 / NT float4 per thread cover NP <= 64 rows), put_rows64 writes them to LDS.
template <int NT> struct Rows64 { float4 v[1024 / NT]; };
template <int NT>
__device__ __forceinline__ void fetch_rows64(const float* src, int N, int NP, Rows64<NT>& R) {
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
        const int i = threadIdx.x + u * NT, r = i >> 4, c = (i & 15) * 4;
        R.v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < NP * 16 && r < N) R.v[u] = *reinterpret_cast<const float4*>(src + (int64_t)r * 64 + c);
    }
}
template <int NT>
__device__ __forceinline__ void put_rows64(const Rows64<NT>& R, int NP, float* dst, float* dstT, int ldn) {
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
        const int i = threadIdx.x + u * NT, r = i >> 4, c = (i & 15) * 4;
        if (i < NP * 16) {
            const float4 v = R.v[u];
            *reinterpret_cast<float4*>(dst + r * LDK + c) = v;
            if (dstT) {
                dstT[(c + 0) * ldn + r] = v.x;
                dstT[(c + 1) * ldn + r] = v.y;
                dstT[(c + 2) * ldn + r] = v.z;
                dstT[(c + 3) * ldn + r] = v.w;
            }
        }
    }
}

__global__ __launch_bounds__(512, 1) void gcn_attn2_fwd_kernel(const float* xin, const float* md, int n_frames, int N,
                                                               float* adj, float* z) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int NP = (N + 15) & ~15, RT = NP >> 4, LDN = NP + 4;
    float* sMt = sm;               // [64][LDK]  Mt[n][k]
    float* sd = sMt + 64 * LDK;    // [64]
    float* sX = sd + 64;           // [NP][LDK]
    float* sP = sX + NP * LDK;     // [NP][LDK]
    float* sXt = sP + NP * LDK;    // [64][LDN]
    float* sS = sXt + 64 * LDN;    // [NP][LDN]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6, i16 = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
        const int r = i >> 4, c = (i & 15) * 4;
        *reinterpret_cast<float4*>(sMt + r * LDK + c) = *reinterpret_cast<const float4*>(md + r * 64 + c);
    }
    if (threadIdx.x < 64) sd[threadIdx.x] = md[64 * 64 + threadIdx.x];
    Rows64<512> rx;
    if ((int)blockIdx.x < n_frames) fetch_rows64(xin + (int64_t)blockIdx.x * N * 64, N, NP, rx);
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const int64_t r0 = (int64_t)f * N;
        __syncthreads();
        put_rows64(rx, NP, sX, sXt, LDN);
        if (f + (int)gridDim.x < n_frames) fetch_rows64(xin + (r0 + (int64_t)gridDim.x * N) * 64, N, NP, rx);   // next frame, in flight under this one
        __syncthreads();
        // P = X M + d
        for (int t = wv; t < RT * 4; t += nw) {
            const int rt = t >> 2, ct = t & 3;
            f32x4m acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_mm(sX + rt * 16 * LDK, LDK, sMt + ct * 16 * LDK, LDK, 16, i16, g, acc);
            const float dv = sd[ct * 16 + i16];
#pragma unroll
            for (int r = 0; r < 4; ++r) sP[(rt * 16 + 4 * g + r) * LDK + ct * 16 + i16] = acc[r] + dv;
        }
        __syncthreads();
        // scores = P X^T
        for (int t = wv; t < RT * RT; t += nw) {
            const int rt = t / RT, ct = t - rt * RT;
            f32x4m acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_mm(sP + rt * 16 * LDK, LDK, sX + ct * 16 * LDK, LDK, 16, i16, g, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) sS[(rt * 16 + 4 * g + r) * LDN + ct * 16 + i16] = acc[r];
        }
        __syncthreads();
        // row softmax over the N real columns; padding rows / columns become exact zeros
        for (int i = wv; i < NP; i += nw) {
            const bool on = i < N && lane < N;
            const float p = on ? sS[i * LDN + lane] : -INFINITY;
            const float m = wave_max(p);
            const float e = on ? expf(p - m) : 0.f;
            const float s = wave_sum(e);
            const float v = on ? e / s : 0.f;
            if (lane < NP) sS[i * LDN + lane] = v;
            if (on) adj[(r0 + i) * N + lane] = v;
        }
        __syncthreads();
        // Z = S X
        for (int t = wv; t < RT * 4; t += nw) {
            const int rt = t >> 2, ct = t & 3;
            f32x4m acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_mm(sS + rt * 16 * LDN, LDN, sXt + ct * 16 * LDN, LDN, RT * 4, i16, g, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + 4 * g + r;
                if (row < N) z[(r0 + row) * 64 + ct * 16 + i16] = acc[r];
            }
        }
    }
}

// Backward per frame (dZ given):   dA = dZ X^T;  dS = S o (dA - rowsum(S o dA));  dP = dS X;
//   dX = S^T dZ + dS^T P + dP M^T;   dMt += dP^T X;   dd += colsum(dP)        (P = X M + d is recomputed)
template <int NT>
__global__ __launch_bounds__(NT, 1) void gcn_attn2_bwd_kernel(const float* xin, const float* md, const float* adj,
                                                               const float* dz, int n_frames, int N, float* dx,
                                                               float* partials, int with_m) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int NP = (N + 15) & ~15, RT = NP >> 4, LDN = NP + 4;
    float* sMt = sm;                 // [64][LDK]  Mt[n][k]
    float* sM = sMt + 64 * LDK;      // [64][LDK]  M[k][n]  (absent when with_m == 0: N > 48 leaves no room for it)
    float* sd = sM + (with_m ? 64 * LDK : 0);  // [64]
    float* sX = sd + 64;             // [NP][LDK]
    float* sdZ = sX + NP * LDK;      // [NP][LDK]   later dP
    float* sXt = sdZ + NP * LDK;     // [64][LDN]
    float* sdZt = sXt + 64 * LDN;    // [64][LDN]   later dP^T
    float* sPt = sdZt + 64 * LDN;    // [64][LDN]
    float* sA = sPt + 64 * LDN;      // [NP][LDN]
    float* sAt = sA + NP * LDN;      // [NP][LDN]   later dS^T
    float* sdS = sAt + NP * LDN;     // [NP][LDN]   dA, then dS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6, i16 = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
        const int r = i >> 4, c = (i & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(md + r * 64 + c);
        *reinterpret_cast<float4*>(sMt + r * LDK + c) = v;
        if (with_m) {
            sM[(c + 0) * LDK + r] = v.x;
            sM[(c + 1) * LDK + r] = v.y;
            sM[(c + 2) * LDK + r] = v.z;
            sM[(c + 3) * LDK + r] = v.w;
        }
    }
    if (threadIdx.x < 64) sd[threadIdx.x] = md[64 * 64 + threadIdx.x];
    // dMt accumulators: 16 tiles (nt, kt) of 16x16, tile t owned by wave t % nw (two tiles per wave at 8 waves)
    constexpr int TU = 1024 / NT;   // tiles of a 16-tile list per wave: 2 with 8 waves, 1 with 16
    f32x4m accM[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) accM[u] = f32x4m{0.f, 0.f, 0.f, 0.f};
    float dd_acc = 0.f;  // threads 0..63: dd[n]
    // the next frame's X, dZ and S rows are fetched into registers while this frame is computed (the frame loop is a chain
    // of six barrier-separated phases on ONE workgroup per CU: nothing else would hide the loads)
    Rows64<NT> rx, rz;
    float ra[4096 / NT];   // S: NP * NP <= 4096 values
    auto fetch_frame = [&](int64_t r0) {
        fetch_rows64(xin + r0 * 64, N, NP, rx);
        fetch_rows64(dz + r0 * 64, N, NP, rz);
#pragma unroll
        for (int u = 0; u < 4096 / NT; ++u) {
            const int i = threadIdx.x + u * NT, r = i / NP, c = i - r * NP;
            ra[u] = (i < NP * NP && r < N && c < N) ? adj[(r0 + r) * N + c] : 0.f;
        }
    };
    if ((int)blockIdx.x < n_frames) fetch_frame((int64_t)blockIdx.x * N);
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const int64_t r0 = (int64_t)f * N;
        __syncthreads();
        put_rows64(rx, NP, sX, sXt, LDN);
        put_rows64(rz, NP, sdZ, sdZt, LDN);
#pragma unroll
        for (int u = 0; u < 4096 / NT; ++u) {
            const int i = threadIdx.x + u * NT, r = i / NP, c = i - r * NP;
            if (i < NP * NP) {
                sA[r * LDN + c] = ra[u];
                sAt[c * LDN + r] = ra[u];
            }
        }
        if (f + (int)gridDim.x < n_frames) fetch_frame(r0 + (int64_t)gridDim.x * N);
        __syncthreads();
        // P^T (recomputed) and dA; dX1 = S^T dZ starts the dX accumulators
        for (int t = wv; t < RT * 4; t += nw) {
            const int rt = t >> 2, ct = t & 3;
            f32x4m acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_mm(sX + rt * 16 * LDK, LDK, sMt + ct * 16 * LDK, LDK, 16, i16, g, acc);
            const float dv = sd[ct * 16 + i16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + 4 * g + r;
                sPt[(ct * 16 + i16) * LDN + row] = row < N ? acc[r] + dv : 0.f;
            }
        }
        for (int t = wv; t < RT * RT; t += nw) {
            const int rt = t / RT, ct = t - rt * RT;
            f32x4m acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_mm(sdZ + rt * 16 * LDK, LDK, sX + ct * 16 * LDK, LDK, 16, i16, g, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) sdS[(rt * 16 + 4 * g + r) * LDN + ct * 16 + i16] = acc[r];
        }
        // this wave's dX tiles (at most 2 with 8 waves and RT <= 4): tile u -> t = wv + u * nw
        f32x4m accX[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) accX[u] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int t = wv + u * nw;
            if (t < RT * 4) {
                const int rt = t >> 2, ct = t & 3;
                accX[u] = tile_mm(sAt + rt * 16 * LDN, LDN, sdZt + ct * 16 * LDN, LDN, RT * 4, i16, g, accX[u]);
            }
        }
        __syncthreads();
        // softmax backward per row; dS in place, dS^T over the (now free) S^T buffer
        for (int i = wv; i < NP; i += nw) {
            const bool on = i < N && lane < N;
            const float sv = on ? sA[i * LDN + lane] : 0.f;
            const float da = on ? sdS[i * LDN + lane] : 0.f;
            const float tsum = wave_sum(sv * da);
            const float ds = on ? sv * (da - tsum) : 0.f;
            if (lane < NP) {
                sdS[i * LDN + lane] = ds;
                sAt[lane * LDN + i] = ds;
            }
        }
        __syncthreads();
        // dP = dS X  (-> sdZ / sdZt buffers, dZ is no longer needed)
        for (int t = wv; t < RT * 4; t += nw) {
            const int rt = t >> 2, ct = t & 3;
            f32x4m acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_mm(sdS + rt * 16 * LDN, LDN, sXt + ct * 16 * LDN, LDN, RT * 4, i16, g, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + 4 * g + r;
                sdZ[row * LDK + ct * 16 + i16] = acc[r];
                sdZt[(ct * 16 + i16) * LDN + row] = acc[r];
            }
        }
        __syncthreads();
        // dX += dS^T P + dP M^T ; store
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int t = wv + u * nw;
            if (t < RT * 4) {
                const int rt = t >> 2, ct = t & 3;
                accX[u] = tile_mm(sAt + rt * 16 * LDN, LDN, sPt + ct * 16 * LDN, LDN, RT * 4, i16, g, accX[u]);
                if (with_m) accX[u] = tile_mm(sdZ + rt * 16 * LDK, LDK, sM + ct * 16 * LDK, LDK, 16, i16, g, accX[u]);
                else accX[u] = tile_mm_bt(sdZ + rt * 16 * LDK, LDK, sMt + ct * 16, LDK, 16, i16, g, accX[u]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + 4 * g + r;
                    if (row < N) dx[(r0 + row) * 64 + ct * 16 + i16] = accX[u][r];
                }
            }
        }
        // dMt += dP^T X  (tile (nt, kt): rows n of dP^T, columns k of X) ; dd += column sums of dP
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int t = wv + u * nw;
            if (t < 16) {
                const int nt = t >> 2, kt = t & 3;
                accM[u] = tile_mm(sdZt + nt * 16 * LDN, LDN, sXt + kt * 16 * LDN, LDN, RT * 4, i16, g, accM[u]);
            }
        }
        if (threadIdx.x < 64) {
            float s = 0.f;
            for (int i = 0; i < N; ++i) s += sdZ[i * LDK + threadIdx.x];
            dd_acc += s;
        }
    }
    float* out = partials + (int64_t)blockIdx.x * (65 * 64);
#pragma unroll
    for (int u = 0; u < TU; ++u) {
        const int t = wv + u * nw;
        if (t < 16) {
            const int nt = t >> 2, kt = t & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(nt * 16 + 4 * g + r) * 64 + kt * 16 + i16] = accM[u][r];
        }
    }
    if (threadIdx.x < 64) out[64 * 64 + threadIdx.x] = dd_acc;
}

inline size_t lds_fwd_bytes(int N) {
    const int NP = (N + 15) & ~15, LDN = NP + 4;
    return sizeof(float) * (size_t)(64 * LDK + 64 + 2 * NP * LDK + 64 * LDN + NP * LDN);
}
inline size_t lds_bwd_bytes(int N, bool with_m) {
    const int NP = (N + 15) & ~15, LDN = NP + 4;
    return sizeof(float) * (size_t)((with_m ? 2 : 1) * 64 * LDK + 64 + 2 * NP * LDK + 3 * 64 * LDN + 3 * NP * LDN);
}

}  // namespace

extern "C" int twog_gcn_attn2_fwd(const float* x, const float* md, int n_frames, int n_nodes, float* adj, float* z,
                                  void* stream) {
    if (n_nodes > MAXN || n_nodes < 1) return -1;
    if (n_frames <= 0) return 0;
    const size_t lds = lds_fwd_bytes(n_nodes);
    static std::atomic<uint32_t> lds_attr_done{0};
    twog_allow_dynamic_lds(gcn_attn2_fwd_kernel, 160 * 1024, lds_attr_done);
    const int grid = n_frames < 512 ? n_frames : 512;
    hipLaunchKernelGGL(gcn_attn2_fwd_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, x, md, n_frames, n_nodes,
                       adj, z);
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_gcn_attn2_bwd_blocks(int n_frames) { return n_frames < 256 ? (n_frames > 0 ? n_frames : 1) : 256; }

extern "C" int twog_gcn_attn2_bwd(const float* x, const float* md, const float* adj, const float* dz, int n_frames,
                                  int n_nodes, float* dx_att, float* partials, int n_blocks, void* stream) {
    if (n_nodes > MAXN || n_nodes < 1) return -1;
    if (n_frames <= 0) return 0;
    if (n_blocks != twog_gcn_attn2_bwd_blocks(n_frames)) return -2;
    const bool with_m = lds_bwd_bytes(n_nodes, true) <= 160 * 1024;
    const size_t lds = lds_bwd_bytes(n_nodes, with_m);
    if (lds > 160 * 1024) return -3;
    // 16 waves: the phases are lists of 9 ... 16 independent 16x16 tiles, each a chain of 12-16 dependent MFMAs -- with 8 waves
    // every list takes two rounds with half the waves idle in the second (TWOG_GCN_ATTN2_WAVES=8: the 8-wave form)
    static const int waves = getenv("TWOG_GCN_ATTN2_WAVES") ? atoi(getenv("TWOG_GCN_ATTN2_WAVES")) : 16;
    static std::atomic<uint32_t> lds_attr_done{0}, lds_attr_done16{0};
    if (waves == 16) {
        twog_allow_dynamic_lds(gcn_attn2_bwd_kernel<1024>, 160 * 1024, lds_attr_done16);
        hipLaunchKernelGGL(gcn_attn2_bwd_kernel<1024>, dim3(n_blocks), dim3(1024), lds, (hipStream_t)stream, x, md, adj, dz,
                           n_frames, n_nodes, dx_att, partials, with_m ? 1 : 0);
    } else {
        twog_allow_dynamic_lds(gcn_attn2_bwd_kernel<512>, 160 * 1024, lds_attr_done);
        hipLaunchKernelGGL(gcn_attn2_bwd_kernel<512>, dim3(n_blocks), dim3(512), lds, (hipStream_t)stream, x, md, adj, dz,
                           n_frames, n_nodes, dx_att, partials, with_m ? 1 : 0);
    }
    TWOG_CHECK_LAUNCH();
    return 0;
}
