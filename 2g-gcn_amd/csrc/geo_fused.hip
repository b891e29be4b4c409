// Geometric-level GCN forward as ONE kernel per group of frames (round 3): BatchNorm fold -> 1x1 conv 4->64 + ReLU ->
// 1x1 conv 64->64 + ReLU -> folded similarity -> row softmax -> adjacency-weighted aggregation.
//
// Reference: Geo_gcn.forward (pyrutils/torch/models_gcn.py:30-37) = norm_data (:45-50), embed (:72-74, :57-63),
// compute_similarity (:95-100) and the adjacency product (:33-34); the final projection  Y = Z W  (:35-36) stays the
// strided GEMM that writes the (bs, 128, N, T) layout T-fastest (ops.geo_gcn_forward).
//
// What it replaces: embed1_fwd (wrote e1: frames*N x 64), the X GEMM (read e1, wrote X) and gcn_attn2_fwd (read X) --
// three launches that each streamed the (frames*N) x 64 activations through HBM, the last one with five workgroup
// barriers per frame. Here a workgroup takes FG consecutive frames (R = FG*N rows, FG = 7 at N = 34):
//   A  X = relu(e1 W2^T + b2) for the group's rows, tiled over the whole GROUP (7 frames x 34 nodes = 238 rows = 15 tiles;
//      frame by frame it would be 7 x 3). e1 is never stored: each lane computes the 16 values of its A fragment
//      (row = lane%16, k = lane/16 + 4 kk) from the four normalised inputs of its row. X -> LDS (+ global: the backward
//      pass reads it). Products on v_mfma_f32_16x16x4_f32 (exact fp32).
//   B  one wave takes 16 rows of ONE frame from start to end: P = X M + d (to its private LDS scratch, to change from the
//      accumulator to the operand layout), scores = P X_f^T against the frame's rows in place (kept in the accumulators),
//      row softmax with 16-lane shuffles (a row of the tile lives on the 16 lanes of one accumulator row group),
//      adjacency rows -> global and -> scratch, Z = S X_f with X_f read in place as the k-major operand (2- to 4-way
//      bank-conflicted reads: 2 + 8 LDS cycles against the MFMA's 32), Z -> global. No workgroup barrier inside B:
//      a wave only writes its own scratch. Two barriers per GROUP of frames instead of five per frame.
// W1, M = Wq^T Wk (folded by bn_finalize), d and the biases live in LDS for the whole kernel; the W2 operand fragments
// are read straight from global memory (16 KB, L1-resident); grid = one workgroup per CU, looping over groups. The backward pass recomputes e1 from the geometry input (embed1_fwd, bit-identical arithmetic).
#include "twog_common.h"

typedef float f32x4g __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXN = 64;
constexpr int LDK = 68;   // row stride of [row][64 features] LDS arrays: = 4 (mod 16) words -> conflict-free k-contiguous reads

__device__ __forceinline__ f32x4g mfma16(float a, float b, f32x4g c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct Layout {
    int FG, R, RT, RP, NP, NT, LDN;
    size_t floats;
};
constexpr int NWAVES = 8;    // two waves per SIMD, up to 256 VGPRs each (16 waves measured slower: 128 VGPRs spill)
inline Layout layout_of(int N, int FG) {
    Layout L;
    L.FG = FG;
    L.R = FG * N;
    L.RT = (L.R + 15) / 16;
    L.NP = (N + 15) & ~15;
    L.NT = L.NP / 16;
    L.RP = L.RT * 16 + 16 + L.NP;   // per-frame tiles of the last frame read up to NP rows past its first row
    L.LDN = L.NP + 4;
    L.floats = (size_t)64 * LDK + 64 + 64 + 256 + 64 + 2 * 4 * MAXN + (size_t)L.RP * LDK + (size_t)NWAVES * 16 * LDK;
    return L;
}

// Operand fragment of a 64-deep product from a [row][64] LDS array (k contiguous): the MFMA's k order is free as long as
// A and B agree, so lane group g = lane / 16 takes k = 16c + 4g + j for step 4c + j -- four consecutive floats per chunk
// c, one ds_read_b128 each (row stride = 4 mod 16 words: conflict-free). All 16 values are requested before the first
// MFMA: the compiler otherwise waits for every pair of ds_read_b32 right in front of the two MFMAs that use them.
__device__ __forceinline__ void frag64(const float* row, int g, float (&f)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(row + 16 * c + 4 * g);
        f[4 * c] = v.x; f[4 * c + 1] = v.y; f[4 * c + 2] = v.z; f[4 * c + 3] = v.w;
    }
}
__device__ __forceinline__ f32x4g mm64(const float (&a)[16], const float (&b)[16], f32x4g acc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = mfma16(a[i], b[i], acc);
    return acc;
}

// reductions over the 16 lanes that share lane / 16 (one accumulator row group of the 16x16 MFMA layout): DPP row
// rotations (row_ror:8/4/2/1 inside each 16-lane row) -- every lane ends up with the result, no LDS traffic
// (__shfl_xor with width 16 compiles to ds_bpermute: ~150 cycles each, 32 of them per tile)
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wave's outstanding global stores (X,
// adjacency and Z rows of the previous frames), which nothing behind the barrier reads
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float dpp_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp_ror<0x128>(v));
    v = fmaxf(v, dpp_ror<0x124>(v));
    v = fmaxf(v, dpp_ror<0x122>(v));
    v = fmaxf(v, dpp_ror<0x121>(v));
    return v;
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_ror<0x128>(v);
    v += dpp_ror<0x124>(v);
    v += dpp_ror<0x122>(v);
    v += dpp_ror<0x121>(v);
    return v;
}

template <int NT>
__global__ __launch_bounds__(64 * NWAVES, 1) void gcn_fused_fwd_kernel(const float* x, int64_t fstride, int n_frames, int N,
                                                                       const float* ab, const float* w1, const float* b1,
                                                                       const float* w2, const float* b2, const float* md,
                                                                       float* xout, float* adj, float* z, int FG) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NP = NT * 16, LDN = NP + 4, KSTEPS = NP / 4;   // NP = N rounded up to the 16-wide MFMA tile
    const int R = FG * N, RT = (R + 15) / 16, RP = RT * 16 + 16 + NP;
    float* sMt = sm;                     // [64][LDK]  Mt[n][k]
    float* sd = sMt + 64 * LDK;          // [64]
    float* sb2 = sd + 64;                // [64]
    float* sw1 = sb2 + 64;               // [64][4]
    float* sb1 = sw1 + 256;              // [64]
    float* sab = sb1 + 64;               // [2][4 * MAXN]
    float* sX = sab + 2 * 4 * MAXN;      // [RP][LDK]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i16 = lane & 15, g = lane >> 4;
    float* priv = sX + RP * LDK + wv * 16 * LDK;   // this wave's [16][LDK] scratch: P rows, then the softmax rows [16][LDN]
    const int nch = 4 * N;
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
        const int r = i >> 4, c = (i & 15) * 4;
        *reinterpret_cast<float4*>(sMt + r * LDK + c) = *reinterpret_cast<const float4*>(md + r * 64 + c);
    }
    if (threadIdx.x < 64) {
        sd[threadIdx.x] = md[64 * 64 + threadIdx.x];
        sb2[threadIdx.x] = b2[threadIdx.x];
        sb1[threadIdx.x] = b1[threadIdx.x];
        *reinterpret_cast<float4*>(sw1 + threadIdx.x * 4) = *reinterpret_cast<const float4*>(w1 + threadIdx.x * 4);
    }
    for (int i = threadIdx.x; i < 2 * nch; i += blockDim.x) sab[(i / nch) * 4 * MAXN + (i % nch)] = ab[i];
    // rows past the group's last tile are only ever read as padding of a frame's last row / column tile: keep them zero
    for (int i = threadIdx.x; i < (RP - RT * 16) * LDK; i += blockDim.x) sX[RT * 16 * LDK + i] = 0.f;
    // wave w takes the column tiles 2 (w & 1), 2 (w & 1) + 1 of every row tile it computes in phase A; the W2 operand
    // fragments come straight from global memory (16 KB, L1-resident): no LDS copy, no registers held across phase B
    const int c0 = (wv & 1) * 2;
    const int n_groups = (n_frames + FG - 1) / FG;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int f0 = grp * FG, nf = min(FG, n_frames - f0), rows = nf * N;
        lds_barrier();     // the previous group's frames are done with sX; first pass: the weights are staged
        // ---- A: X of the group's rows; item = (16-row tile, pair of 16-column tiles)
        for (int rt = wv >> 1; rt < RT; rt += NWAVES / 2) {
            const int arow = rt * 16 + i16;
            float4 xh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow < rows) {   // x^ = a x + b of this lane's row, straight from the geometry input
                const int f = arow / N, n = arow - f * N;
                const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)(f0 + f) * fstride + n * 4);
                xh.x = fmaf(sab[n], v.x, sab[4 * MAXN + n]);
                xh.y = fmaf(sab[N + n], v.y, sab[4 * MAXN + N + n]);
                xh.z = fmaf(sab[2 * N + n], v.z, sab[4 * MAXN + 2 * N + n]);
                xh.w = fmaf(sab[3 * N + n], v.w, sab[4 * MAXN + 3 * N + n]);
            }
            float ae[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);   // the k of fragment slot kk (see frag64)
                const float4 w = *reinterpret_cast<const float4*>(sw1 + k * 4);
                float acc = sb1[k];
                acc = fmaf(w.x, xh.x, acc);
                acc = fmaf(w.y, xh.y, acc);
                acc = fmaf(w.z, xh.z, acc);
                acc = fmaf(w.w, xh.w, acc);
                ae[kk] = fmaxf(acc, 0.f);
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int ct = c0 + cc;
                float bf[16];
                frag64(w2 + (ct * 16 + i16) * 64, g, bf);
                const f32x4g acc = mm64(ae, bf, f32x4g{0.f, 0.f, 0.f, 0.f});
                const int col = ct * 16 + i16;
                const float bias = sb2[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + 4 * g + r;
                    const float v = row < rows ? fmaxf(acc[r] + bias, 0.f) : 0.f;
                    sX[row * LDK + col] = v;
                }
            }
        }
        lds_barrier();
        if (xout) {   // X of the group -> global, whole 256-byte rows (the backward pass reads it)
            for (int i = threadIdx.x; i < rows * 16; i += blockDim.x) {
                const int r = i >> 4, c = (i & 15) * 4;
                *reinterpret_cast<float4*>(xout + ((int64_t)f0 * N + r) * 64 + c) = *reinterpret_cast<const float4*>(sX + r * LDK + c);
            }
        }
        // ---- B: item = 16 rows of one frame, start to end in ONE wave: P rows -> scores -> softmax -> Z rows. The only
        // LDS it writes is the wave's own scratch, so the frames need no workgroup barrier between these steps.
        for (int it = wv; it < nf * NT; it += NWAVES) {
            const int f = it / NT, rt = it - f * NT;
            const int fbase = f * N, base = fbase + rt * 16;
            // P = X M + d for the item's rows -> scratch [16][LDK]
            {
                float af[16];
                frag64(sX + (base + i16) * LDK, g, af);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    float bf[16];
                    frag64(sMt + (ct * 16 + i16) * LDK, g, bf);
                    const f32x4g acc = mm64(af, bf, f32x4g{0.f, 0.f, 0.f, 0.f});
                    const int col = ct * 16 + i16;
                    const float dv = sd[col];
#pragma unroll
                    for (int r = 0; r < 4; ++r) priv[(4 * g + r) * LDK + col] = acc[r] + dv;
                }
            }
            // scores against every node of the frame: NT column tiles (NT <= 4), kept in the accumulators
            f32x4g sc[4];
            {
                float af[16];
                frag64(priv + i16 * LDK, g, af);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    sc[ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
                    if (ct < NT) {
                        float bf[16];
                        frag64(sX + (fbase + ct * 16 + i16) * LDK, g, bf);
                        sc[ct] = mm64(af, bf, sc[ct]);
                    }
                }
            }
            // row softmax over the N real columns: row 4g + r of the tile lives on the 16 lanes of group g, one column per
            // lane and column tile. Padding columns become exact zeros. Result -> scratch [16][LDN] (P is dead) and global.
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (ct < NT && ct * 16 + i16 < N) m = fmaxf(m, sc[ct][r]);
                m = max16(m);
                float e[4], s = 0.f;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    e[ct] = (ct < NT && ct * 16 + i16 < N) ? __expf(sc[ct][r] - m) : 0.f;
                    s += e[ct];
                }
                const float inv = 1.0f / sum16(s);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (ct < NT) priv[(4 * g + r) * LDN + ct * 16 + i16] = e[ct] * inv;
            }
            {   // adjacency rows of the item -> global, row by row on the lanes (no index division: these kernels are bound by
                // their VALU instruction count -- a wave64 VALU instruction issues over 4 cycles -- not by MFMA or LDS)
                const int nr = min(16, N - rt * 16);
                float* dst = adj + ((int64_t)(f0 + f) * N + rt * 16) * N + lane;
                float av[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = priv[r * LDN + lane];   // (lanes >= LDN - ... read inside the scratch: LDN <= LDK)
                if (lane < N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (r < nr) dst[r * N] = av[r];
                }
            }
            // Z = S X for the item's rows: k runs over the frame's nodes (rows of X_f), k = g + 4 kk; fragments first
            {
                // (k runs to NP: the padding columns of S are exact zeros and the rows of sX behind the frame are finite --
                // the next frame's rows or the zeroed tail -- so the extra steps add nothing; a compile-time trip count
                // keeps the 16 + 4 x KSTEPS operand loads and the MFMAs free of branches)
                float sa[KSTEPS];
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) sa[kk] = priv[i16 * LDN + g + 4 * kk];
                f32x4g zc[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const float* pb = sX + (fbase + g) * LDK + ct * 16 + i16;
                    float xb[KSTEPS];
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) xb[kk] = pb[kk * 4 * LDK];
                    zc[ct] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) zc[ct] = mfma16(sa[kk], xb[kk], zc[ct]);
                }
                // the softmax rows are dead: Z rows -> scratch [16][LDK] -> global as whole 256-byte rows
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) priv[(4 * g + r) * LDK + ct * 16 + i16] = zc[ct][r];
                const int nr = min(16, N - rt * 16);
                float* dst = z + ((int64_t)(f0 + f) * N + rt * 16) * 64;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = lane + 64 * j, r = i >> 4, c = (i & 15) * 4;
                    if (r < nr) *reinterpret_cast<float4*>(dst + r * 64 + c) = *reinterpret_cast<const float4*>(priv + r * LDK + c);
                }
            }
        }
    }
}

}  // namespace

// x_geo: geometry of human 0 (x_human + 2048), frame f at x_geo + f * frame_stride; ab [2][4N] from twog_bn_finalize;
// w1 [64][4], b1 [64] (joint_embed.cnn.1), w2 [64][64], b2 [64] (joint_embed.cnn.3), md [65][64] = [Mt | d] (the folded
// similarity, see geo_attn_mfma.hip). Outputs: x_out [(f,n)][64] (may be NULL: inference), adj [f][N][N], z [(f,n)][64].
extern "C" int twog_gcn_fused_fwd(const float* x_geo, int64_t frame_stride, int n_frames, int n_nodes, const float* ab,
                                  const float* w1, const float* b1, const float* w2, const float* b2, const float* md,
                                  float* x_out, float* adj, float* z, void* stream) {
    if (n_nodes > MAXN || n_nodes < 1) return -1;
    if (n_frames <= 0) return 0;
    // frames per group: as many as fit the CU's LDS (rows of X for the group + one scratch tile per wave), at most 8
    static const int force_fg = getenv("TWOG_GCN_FG") ? atoi(getenv("TWOG_GCN_FG")) : 0;
    Layout best = layout_of(n_nodes, 1);
    for (int fg = 2; fg <= 8; ++fg) {
        const Layout L = layout_of(n_nodes, fg);
        if (L.floats * sizeof(float) <= 160 * 1024) best = L;
    }
    if (force_fg > 0 && layout_of(n_nodes, force_fg).floats * sizeof(float) <= 160 * 1024) best = layout_of(n_nodes, force_fg);
    const size_t lds = best.floats * sizeof(float);
    if (lds > 160 * 1024) return -3;
    const int n_groups = (n_frames + best.FG - 1) / best.FG;
    const int grid = n_groups < 256 ? n_groups : 256;
    static std::atomic<uint32_t> done1{0}, done2{0}, done3{0}, done4{0};
#define TWOG_GCN_LAUNCH(NT_, FLAG_)                                                                                      \
    do {                                                                                                                 \
        twog_allow_dynamic_lds(gcn_fused_fwd_kernel<NT_>, 160 * 1024, FLAG_);                                            \
        hipLaunchKernelGGL(gcn_fused_fwd_kernel<NT_>, dim3(grid), dim3(64 * NWAVES), lds, (hipStream_t)stream, x_geo,    \
                           frame_stride, n_frames, n_nodes, ab, w1, b1, w2, b2, md, x_out, adj, z, best.FG);            \
    } while (0)
    switch (best.NT) {
        case 1: TWOG_GCN_LAUNCH(1, done1); break;
        case 2: TWOG_GCN_LAUNCH(2, done2); break;
        case 3: TWOG_GCN_LAUNCH(3, done3); break;
        default: TWOG_GCN_LAUNCH(4, done4); break;
    }
#undef TWOG_GCN_LAUNCH
    TWOG_CHECK_LAUNCH();
    return 0;
}
