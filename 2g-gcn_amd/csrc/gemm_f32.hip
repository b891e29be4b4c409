// Grouped fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 256 FLOP/clk/CU).
//
//   C[m][n] = act( sum_k A(m,k) * B(n,k) + bias[n] ) (+ C)
//
// Replaces every dense projection of the hot path: build_mlp Linear layers (reference pyrutils/torch/models.py:31-36),
// GRU / GRUCell input+hidden projections (vhoi/models.py:267-320), message MLPs (:323-520), label heads (:552-580)
// and -- through the k-major operand forms -- their backward passes.
//
// Design (MI355X-first): workgroups of 4 waves (2x2 grid; the 64x64 class) or 8 waves (4x2 grid; the 128x128 class), each
// wave owning its share of the tile as 32x32 MFMA accumulators. Operand tiles are staged global -> registers -> LDS with a 2-deep software pipeline
// (loads for k-tile t+1 are issued before the MFMAs of tile t, written to the other LDS buffer after them; one
// barrier per k-tile). K-contiguous operands are kept [row][BK+4] in LDS and read as one ds_read_b128 per 4 k-steps
// (the MFMA k order is arbitrary as long as A and B agree, so lane half kh takes k = 4kh..4kh+3 of each 8-chunk;
// the +4 pad makes the 16-lane b128 groups conflict-free); k-major operands are kept [k][rows+4] and read with
// conflict-free ds_read_b32. Workgroup ids are remapped so that tiles sharing an A row-panel run on one XCD (its
// L2 then serves the panel once). Tall reductions (dW = dY^T X, K = rows) use deterministic split-K: partial slabs in
// the caller's workspace, summed in fixed order by a second kernel.
#include <cstdlib>
#include "twog_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: loads/stores stay in registers (SROA)

namespace {

constexpr int BK = 32;
constexpr int MAXP = 8;

struct Prob {
    twog_rows_t A, B, C;
    const float* bias;
    int M, N, K;
    int act, accumulate;
    int tiles_m, tiles_n, tile_start;
    int a_vec, b_vec;  // 16-byte vector loads legal for this operand
    int n_major;       // logical tile order: consecutive tiles share the B column panel (weights larger than activations)
    int batch;         // independent problems sharing shapes; operand b of batch i = ptr + i * *_bs
    int64_t a_bs, b_bs, c_bs;
    float* cs;         // column sums of the k-major A operand (twog_gemm_t::a_colsum), or nullptr
    int cs_acc;
};

struct Group {
    Prob p[MAXP];
    int n;
    int total_tiles;
    int n_cls;       // classes of equal K: tiles [cls_start, cls_start + cls_ntiles), remainder ring offset cls_rot
    int cls_start[MAXP], cls_ntiles[MAXP], cls_rot[MAXP];
    int splitk;      // >= 1
    int k_per_split; // multiple of BK
    int group;       // tile order inside a problem: groups of `group` panels of the major dimension (tile_coords)
    int xcd_split;   // split-K launches with splitk % 8 == 0: XCD x works the k-splits x, x + 8, ... of ALL tiles (see gemm_tile)
    float* slabs;    // split-K partials: [problem-tile-major] see below
    float* cs_part;  // split-K launches with a_colsum requests: [split][tile][128] partial column sums (tiles with tn == 0)
    unsigned* xcnt;  // XS kernels (split-K over workgroups, combined in the launch): arrival ticket per tile, zero between launches
    int xs_early;    // XS: every slice requests the epilogue operands before its reduction loop (small launches: the
                     // combining workgroup then has them when it draws the last ticket; large ones fetch them once, late)
};

// Fused "next gate step" epilogue of the recurrent backward chains (gru.hip / segrnn.hip): the launch that adds the last
// contribution to the carried state gradient finishes, per output tile, the element-wise GRU gate backward of the NEXT
// chain step from the tile still in its accumulators (no separate gate launch, no re-read of the carry).
struct GateArgs {
    twog_gru_step_bwd_t s[MAXP];
    float* du_part[MAXP];  // [2 * tiles_n][rows] partial row sums of d * (gru - h_prev) for this step, or nullptr
    int gate_of[MAXP];     // per (sorted) problem: index into s, -1 = plain epilogue
};

// Tile index -> (row panel, column panel) of a problem. Plain order: major panel by major panel (row panels; column
// panels for n_major problems), minor index fastest -- the tiles that share the major panel start together and meet its
// k-slices in L2. When the minor dimension is WIDE (>= 12 panels) the list is walked in groups of `group` major panels
// instead, major index fastest inside the group: the 64 tiles an XCD has in flight (32 CUs x 2 workgroups) then cover
// about 8 x 8 panels instead of 2 x 34, so its L2 is not asked for every minor panel again for each major panel.
// Measured per launch at the L2<->fabric boundary (rocprofv3 FETCH_SIZE, bs64 step): dX of the geometry MLP
// (7 680 x 4 352 x 2 048, 60 x 34 tiles) 2 273 -> 920 MB of reads for 232 MB of operands; the 12-panel-wide forward
// launches -12...-15 %; with 4 or 8 minor panels grouping made it WORSE (+25...+40 %: the tiles sharing a row panel no
// longer start together), hence the width test. Time is unchanged either way (the class is MFMA-bound).
__device__ __forceinline__ void tile_coords(const Prob& P, int tile, int group, int& tm, int& tn) {
    const int major = P.n_major ? P.tiles_n : P.tiles_m, minor = P.n_major ? P.tiles_m : P.tiles_n;
    int a, b;
    if (group <= 1 || minor < 12) {
        a = tile / minor;
        b = tile - a * minor;
    } else {
        const int per_group = group * minor;
        const int grp = tile / per_group, first = grp * group;
        const int gsize = min(major - first, group);
        const int r = tile - grp * per_group;
        b = r / gsize;
        a = first + (r - b * gsize);
    }
    if (P.n_major) { tn = a; tm = b; }
    else { tm = a; tn = b; }
}

// tile of ROWS x COLS (COLS contiguous in memory) -> registers; out-of-range elements read as 0
template <int ROWS, int COLS, int NT>
struct TileRegs {
    static constexpr int R = ROWS, C = COLS;
    static constexpr int F4_PER_ROW = COLS / 4;
    static constexpr int ROWS_PER_PASS = NT / F4_PER_ROW;
    static constexpr int PASSES = ROWS / ROWS_PER_PASS;
    f32x4 v[PASSES];
};

// a wave-uniform pointer pinned to scalar registers, so that loads can use the scalar-base + 32-bit vector-offset form
__device__ __forceinline__ const char* uniform_ptr(const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// guarded load (edge tiles / unaligned operands): out-of-range elements read as 0
template <int ROWS, int COLS, int NT>
__device__ __forceinline__ void load_tile(TileRegs<ROWS, COLS, NT>& t, const twog_rows_t& m, int r0, int c0, int rmax,
                                          int cmax, int vec_ok) {
    using T = TileRegs<ROWS, COLS, NT>;
    const int tid = threadIdx.x;
    const int c = c0 + (tid % T::F4_PER_ROW) * 4;
#pragma unroll
    for (int i = 0; i < T::PASSES; ++i) {
        const int r = r0 + tid / T::F4_PER_ROW + i * T::ROWS_PER_PASS;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rmax && c < cmax) {
            const float* p = m.ptr + twog_row_off(m, r) + c;
            if (vec_ok && c + 3 < cmax) {
                v = *reinterpret_cast<const f32x4*>(p);
            } else {
                v.x = p[0];
                if (c + 1 < cmax) v.y = p[1];
                if (c + 2 < cmax) v.z = p[2];
                if (c + 3 < cmax) v.w = p[3];
            }
        }
        t.v[i] = v;
    }
}

template <int ROWS, int COLS, int LD, int NT>
__device__ __forceinline__ void store_tile(const TileRegs<ROWS, COLS, NT>& t, float* s) {
    using T = TileRegs<ROWS, COLS, NT>;
    const int tid = threadIdx.x;
    const int c = (tid % T::F4_PER_ROW) * 4;
#pragma unroll
    for (int i = 0; i < T::PASSES; ++i) {
        const int r = tid / T::F4_PER_ROW + i * T::ROWS_PER_PASS;
        *reinterpret_cast<f32x4*>(s + r * LD + c) = t.v[i];
    }
}

// KS = 2 (k-split inside the workgroup): the NT threads form two groups of NT/2; both stage the operand tiles together,
// group g multiplies the k-chunks [g * BK/16, (g+1) * BK/16) of every k-tile; the two partial tiles are added through
// LDS by the caller (gemm_tile). For launches with fewer tiles than CUs (the recurrent chains) this puts two waves on
// every SIMD and halves the dependent MFMA chain of a tile: +1 % on the bs64 step, +7 % at 8 clips per GPU. (The
// k-loop of such a launch is NOT bound by its MFMAs or by load latency -- a prefetch distance of 4 changed nothing --
// but by the ~27 GB/s a lone 64x64 tile pulls through its CU's L2 port at 16 FLOP/B; measured, see profiles/HISTORY.md section 8.)
// G3 (fused GRU forward step, gemm_gru_fwd_kernel): B holds the three gate blocks of a GRU weight, [3N][K] rows r | z | n,
// and the tile's 192 columns are 64 hidden units x 3 gates laid out so that accumulator b of every wave is gate b of
// the SAME 32 units: tile column c -> gate (c % 96) / 32, unit n0 + 32 * (c / 96) + c % 32. N counts hidden units.
template <int BM, int BN, int NT, bool AKM, bool BKM, bool FAST, int TM, int TN, int D, bool KG, int KS = 1, bool G3 = false>
__device__ __forceinline__ void gemm_mainloop(const twog_rows_t A, const twog_rows_t B, int M, int N, int a_vec,
                                              int b_vec, int m0, int n0, int k_begin, int k_end, float* smem,
                                              f32x16 (&acc)[TM][TN]) {
    constexpr int NTG = NT / KS;                        // threads of one k-group
    constexpr int WM = BM / (NTG / 128), WN = BN / 2;  // waves of a group in a (NTG/128) x 2 grid
    constexpr int LDA = AKM ? (BM + 4) : (BK + 4);
    constexpr int LDB = BKM ? (BN + 4) : (BK + 4);
    constexpr int A_ELEMS = AKM ? BK * LDA : BM * LDA;
    constexpr int B_ELEMS = BKM ? BK * LDB : BN * LDB;
    constexpr int STAGE = A_ELEMS + B_ELEMS;  // buffer b: A at smem + b*STAGE, B right behind it
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % (NTG / 64), kgrp = (threadIdx.x >> 6) / (NTG / 64);
    const int li = lane & 31, kh = lane >> 5;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int tid = threadIdx.x;

    using ARegs = TileRegs<(AKM ? BK : BM), (AKM ? BM : BK), NT>;
    using BRegs = TileRegs<(BKM ? BK : BN), (BKM ? BN : BK), NT>;
    // FAST: 16-byte aligned operands and only whole k-tiles -> every lane issues unconditional global_load_dwordx4
    // (row / column indices beyond the matrix are clamped: they only feed outputs that are never stored), so the
    // loads stay in flight under the MFMAs. All row addressing is resolved before the loop: per pass one pointer
    // (row-major operand: the tile row; k-major operand with plain rows: row 0 of the pass, advanced by k0 * ld).
    // per pass ONE 32-bit BYTE offset from a wave-uniform base (the operand pointer advanced by the k-tile): the loads
    // take the scalar-base + vector-offset form, no 64-bit address arithmetic and no extra registers inside the loop
    // (the host only enables FAST for operands smaller than 2^30 elements)
    uint32_t oa[ARegs::PASSES], ob[BRegs::PASSES];
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(A.ptr, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(B.ptr, 0, 0xffffffff, 0x00020000);
    if constexpr (FAST) {
#pragma unroll
        for (int i = 0; i < ARegs::PASSES; ++i) {
            const int rr = tid / ARegs::F4_PER_ROW + i * ARegs::ROWS_PER_PASS, cc = (tid % ARegs::F4_PER_ROW) * 4;
            if constexpr (AKM) oa[i] = 4u * (uint32_t)((int64_t)rr * A.ld_outer + min(m0 + cc, M - 4));
            else oa[i] = 4u * (uint32_t)(twog_row_off(A, min(m0 + rr, M - 1)) + cc);
        }
#pragma unroll
        for (int i = 0; i < BRegs::PASSES; ++i) {
            const int rr = tid / BRegs::F4_PER_ROW + i * BRegs::ROWS_PER_PASS, cc = (tid % BRegs::F4_PER_ROW) * 4;
            if constexpr (G3) {
                static_assert(!G3 || (FAST && !BKM && !KG && BN == 192 && TN == 3), "gate-aware column map: 64 x 192 tiles");
                const int gate = (rr % 96) / 32, unit = n0 + 32 * (rr / 96) + (rr & 31);
                ob[i] = 4u * (uint32_t)(twog_row_off(B, gate * N + min(unit, N - 1)) + cc);
            } else if constexpr (BKM) ob[i] = 4u * (uint32_t)((int64_t)rr * B.ld_outer + min(n0 + cc, N - 4));
            else ob[i] = 4u * (uint32_t)(twog_row_off(B, min(n0 + rr, N - 1)) + cc);
        }
    }
    // KG kernels (k-major operand whose rows are (outer, inner) grouped, e.g. "all but the first time step of every
    // clip"): the row pointer of every pass is carried from k-tile to k-tile (k only moves forward): no division in the
    // loop. Kept out of the plain kernels, whose registers are full.
    const float* qa[KG ? ARegs::PASSES : 1];
    const float* qb[KG ? BRegs::PASSES : 1];
    int qa_i[KG ? ARegs::PASSES : 1], qb_i[KG ? BRegs::PASSES : 1], qa_k = k_begin, qb_k = k_begin;
    const int a_inner = A.inner <= 1 ? 0x7fffffff : A.inner, b_inner = B.inner <= 1 ? 0x7fffffff : B.inner;
    const int64_t a_step = A.inner <= 1 ? A.ld_outer : A.ld_inner, b_step = B.inner <= 1 ? B.ld_outer : B.ld_inner;
    if constexpr (FAST && KG && AKM) {
#pragma unroll
        for (int i = 0; i < ARegs::PASSES; ++i) {
            const int row = k_begin + tid / ARegs::F4_PER_ROW + i * ARegs::ROWS_PER_PASS;
            const int o = A.inner <= 1 ? 0 : row / A.inner;
            qa_i[i] = A.inner <= 1 ? 0 : row - o * A.inner;
            qa[i] = A.ptr + twog_row_off(A, row) + min(m0 + (tid % ARegs::F4_PER_ROW) * 4, M - 4);
        }
    }
    if constexpr (FAST && KG && BKM) {
#pragma unroll
        for (int i = 0; i < BRegs::PASSES; ++i) {
            const int row = k_begin + tid / BRegs::F4_PER_ROW + i * BRegs::ROWS_PER_PASS;
            const int o = B.inner <= 1 ? 0 : row / B.inner;
            qb_i[i] = B.inner <= 1 ? 0 : row - o * B.inner;
            qb[i] = B.ptr + twog_row_off(B, row) + min(n0 + (tid % BRegs::F4_PER_ROW) * 4, N - 4);
        }
    }
    auto gload = [&](ARegs& ra, BRegs& rb, int k0) {
        if constexpr (FAST && KG) {
            if constexpr (AKM) {
                const int delta = k0 - qa_k;  // >= 0: tiles are visited in order (the clamped tail repeats the last one)
                qa_k = k0;
                const int64_t wrap = A.ld_outer - (int64_t)A.inner * A.ld_inner;
#pragma unroll
                for (int i = 0; i < ARegs::PASSES; ++i) {
                    qa_i[i] += delta;
                    qa[i] += (int64_t)delta * a_step;
                    while (qa_i[i] >= a_inner) { qa_i[i] -= a_inner; qa[i] += wrap; }
                }
            }
            if constexpr (BKM) {
                const int delta = k0 - qb_k;
                qb_k = k0;
                const int64_t wrap = B.ld_outer - (int64_t)B.inner * B.ld_inner;
#pragma unroll
                for (int i = 0; i < BRegs::PASSES; ++i) {
                    qb_i[i] += delta;
                    qb[i] += (int64_t)delta * b_step;
                    while (qb_i[i] >= b_inner) { qb_i[i] -= b_inner; qb[i] += wrap; }
                }
            }
#pragma unroll
            for (int i = 0; i < ARegs::PASSES; ++i)
                ra.v[i] = *reinterpret_cast<const f32x4*>(AKM ? reinterpret_cast<const char*>(qa[i]) : reinterpret_cast<const char*>(A.ptr + k0) + oa[i]);
#pragma unroll
            for (int i = 0; i < BRegs::PASSES; ++i)
                rb.v[i] = *reinterpret_cast<const f32x4*>(BKM ? reinterpret_cast<const char*>(qb[i]) : reinterpret_cast<const char*>(B.ptr + k0) + ob[i]);
        } else if constexpr (FAST) {
            // buffer loads: 128-bit scalar descriptor + scalar k-tile offset + per-lane 32-bit offset (no address VALU)
            const int sa = (int)(AKM ? (uint32_t)k0 * (uint32_t)A.ld_outer * 4u : (uint32_t)k0 * 4u);
            const int sb = (int)(BKM ? (uint32_t)k0 * (uint32_t)B.ld_outer * 4u : (uint32_t)k0 * 4u);
#pragma unroll
            for (int i = 0; i < ARegs::PASSES; ++i)
                ra.v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)oa[i], sa, 0));
#pragma unroll
            for (int i = 0; i < BRegs::PASSES; ++i)
                rb.v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)ob[i], sb, 0));
        } else {
            if constexpr (AKM) load_tile(ra, A, k0, m0, k_end, M, a_vec);
            else               load_tile(ra, A, m0, k0, M, k_end, a_vec);
            if constexpr (BKM) load_tile(rb, B, k0, n0, k_end, N, b_vec);
            else               load_tile(rb, B, n0, k0, N, k_end, b_vec);
        }
    };
    auto sstore = [&](const ARegs& ra, const BRegs& rb, int buf) {
        float* dst = smem + buf * STAGE;
        store_tile<ARegs::R, ARegs::C, LDA, NT>(ra, dst);
        store_tile<BRegs::R, BRegs::C, LDB, NT>(rb, dst + A_ELEMS);
    };
    // fragments of 8 k-steps: lane (li, kh) holds k = 4kh..4kh+3 of the chunk for its row / column
    auto frag_load = [&](const float* a_s, const float* b_s, int kk, float (&af)[TM][4], float (&bf)[TN][4]) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if constexpr (AKM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) af[a][r] = a_s[(kk * 8 + kh * 4 + r) * LDA + wm + a * 32 + li];
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(a_s + (wm + a * 32 + li) * LDA + kk * 8 + kh * 4);
                af[a][0] = v.x; af[a][1] = v.y; af[a][2] = v.z; af[a][3] = v.w;
            }
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            if constexpr (BKM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bf[b][r] = b_s[(kk * 8 + kh * 4 + r) * LDB + wn + b * 32 + li];
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(b_s + (wn + b * 32 + li) * LDB + kk * 8 + kh * 4);
                bf[b][0] = v.x; bf[b][1] = v.y; bf[b][2] = v.z; bf[b][3] = v.w;
            }
        }
    };
    // single-accumulator waves (64x64 class): a second accumulator takes the odd k-steps, so consecutive MFMAs never
    // wait on each other's result; the two are summed once after the reduction loop
    constexpr bool SPLIT_ACC = (TM * TN == 1);
    f32x16 acc_odd;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;
    // the fragments of chunk kk+1 are read from LDS while the MFMAs of chunk kk run (two register sets)
    auto compute = [&](int buf) {
        const float* a_s = smem + buf * STAGE;
        const float* b_s = a_s + A_ELEMS;
        float af[2][TM][4], bf[2][TN][4];
        constexpr int KCH = (BK / 8) / KS;   // 8-deep k-chunks per k-tile handled by this group
        const int kbase = kgrp * KCH;
        frag_load(a_s, b_s, kbase, af[0], bf[0]);
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk) {
            if (kk + 1 < KCH) frag_load(a_s, b_s, kbase + kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        if (SPLIT_ACC && (r & 1))
                            acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][a][r], bf[kk & 1][b][r], acc_odd, 0, 0, 0);
                        else
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][a][r], bf[kk & 1][b][r], acc[a][b], 0, 0, 0);
                    }
        }
    };
    auto fold = [&]() {
        if constexpr (SPLIT_ACC) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[r];
        }
    };

    const int nkt = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;
    if (nkt == 0) return;
    ARegs ra0;
    BRegs rb0;
    // Loads and LDS stores past the last k-tile are clamped to it instead of branched around (FAST path): a branch
    // makes the compiler merge its outstanding-load counters conservatively (s_waitcnt vmcnt(0) before the LDS
    // stores), which would serialise the younger in-flight tile. The redundant tail tile is never read.
    const int k_last = k_begin + (nkt - 1) * BK;
    if constexpr (D == 1 || !FAST) {
        // loads of k-tile t+1 in flight under the MFMAs of tile t
        gload(ra0, rb0, k_begin);
        sstore(ra0, rb0, 0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (FAST || kt + 1 < nkt) gload(ra0, rb0, min(k_begin + (kt + 1) * BK, k_last));
            compute(buf);
            if (FAST || kt + 1 < nkt) sstore(ra0, rb0, buf ^ 1);
            __syncthreads();
        }
        fold();
    } else {
        // prefetch distance 2 (two register stages): the loads of tile t+2 are issued before the MFMAs of tile t and are
        // consumed a whole iteration later, so short-k-tile kernels (64x64 tiles: 16 MFMAs per wave per k-tile) still
        // cover the L2/HBM latency. Loop unrolled by two so the register stages keep static names.
        ARegs ra1;
        BRegs rb1;
        gload(ra0, rb0, k_begin);
        gload(ra1, rb1, min(k_begin + BK, k_last));
        sstore(ra0, rb0, 0);
        __syncthreads();
        // whole pairs of k-tiles in the loop (no exit in the middle of the body: the accumulators keep one register
        // assignment), a possible odd last tile after it -- it already sits in LDS buffer 0
        for (int kt = 0; kt + 1 < nkt; kt += 2) {
            gload(ra0, rb0, min(k_begin + (kt + 2) * BK, k_last));
            compute(0);
            sstore(ra1, rb1, 1);
            __syncthreads();
            gload(ra1, rb1, min(k_begin + (kt + 3) * BK, k_last));
            compute(1);
            sstore(ra0, rb0, 0);
            __syncthreads();
        }
        if (nkt & 1) compute(0);
        fold();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// X3: the 128x128 class on the bf16 matrix cores with fp32-exact operands (default; TWOG_GEMM_X3=0 selects the native
// fp32-MFMA kernels above for every launch).
//
// Every fp32 operand element is split EXACTLY into three bf16 values, x = h + m + l: h = x truncated to its top 8
// significant bits, m = (x - h) truncated, l = x - h - m (at most 8 significant bits are left: no rounding anywhere,
// the 24-bit significand is three 8-bit chunks). A product a b = sum of nine chunk products; the six with weight
// >= 2^-16 -- hh, hm, mh, mm, hl, lh -- run on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in the fp32
// accumulator); the three dropped ones (ml, lm, ll) are <= 2^-21 |a b| in the worst case and ~2^-24 |a b| on average,
// i.e. of the order of the rounding the fp32 accumulation applies to every partial sum anyway: measured against fp64
// products the kernel's error equals the native fp32-MFMA kernel's on every shape of tools/gemm_x3_bench.py (0.7-2.5e-6
// of the largest output for K = 512 ... 61 440); adding the two products m l and l m (exact to 2^-30) changes no digit of
// that error and costs 15 % (TWOG_X3_PRODUCTS=8 at build time; the shipped default is 6). The truncating split makes the
// dropped terms carry the sign of a b -- a bias towards zero of ~2^-24 |a b| per product, far below what the accumulate
// itself does (see TMPACC in compute()). Six bf16 MFMAs of 16 k-steps cost 192 cycles per
// 32x32 block where the fp32 MFMA (32x32x2) needs 512: 2.67x the matrix rate for the same result to fp32 rounding.
//
// Structure: 128x128 tile, 8 waves of 32x64, k-tiles of 16 (one bf16 MFMA k-step), the same branch-free 16-byte global
// loads into registers (two stages); the split happens once per element on the way into LDS (5.5 VALU operations per
// element), LDS holds three bf16 planes per operand in two stages (2 x 24 KB; two workgroups per CU by registers), one barrier per
// k-tile. Row-major operands ([row][k], k contiguous) are stored as [row][16 k] rows of 32 bytes whose two 16-byte chunks
// are swapped on every second group of 16 rows (the ds_read_b128 fragment reads and the ds_write_b64 stores are conflict-free); k-major operands ([k][row]) are stored as they come, [k][128 rows] rows of 256 bytes with 16-byte
// chunks XOR-swizzled by the k row, and transposed on the way out by ds_read_b64_tr_b16 (each 16-lane group receives a
// 4 k x 16 rows block column-major: the MFMA's k-contiguous fragment with no data movement of our own).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_hi16(uint32_t lo, uint32_t hi) {   // {hi[31:16], lo[31:16]}
    return __builtin_amdgcn_perm(hi, lo, 0x07060302);
}
// (Non-finite operands: NaN stays NaN; an operand of +-Inf gives NaN where an fp32 multiply would give +-Inf, because the
// split subtracts Inf - Inf. The path's GEMM operands are activations, weights and gradients -- finite in any run that is
// not already broken; the masked softmax's -inf lives inside the attention kernels, never in a GEMM operand.)
// four consecutive fp32 values -> their three bf16 planes (4 x 2 bytes each), by truncation: h = the top 8 significant bits,
// m = the top 8 of x - h, l = x - h - m. Exact (both subtractions are, and at most 8 significant bits are left for l), never
// overflows (a rounding split -- v_cvt_pk_bf16_f32 -- measured 5-8 % slower with the same end-to-end error).
__device__ __forceinline__ void split3(const f32x4 v, i32x2& ph, i32x2& pm, i32x2& pl) {
    uint32_t x[4], r1[4], r2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float f0 = v[i];   // (a scalar copy: __builtin_bit_cast applied to the vector ELEMENT v[i] reads element 0)
        x[i] = __float_as_uint(f0);
        const float f1 = f0 - __uint_as_float(x[i] & 0xffff0000u);
        r1[i] = __float_as_uint(f1);
        r2[i] = __float_as_uint(f1 - __uint_as_float(r1[i] & 0xffff0000u));
    }
    ph = i32x2{(int)pack_hi16(x[0], x[1]), (int)pack_hi16(x[2], x[3])};
    pm = i32x2{(int)pack_hi16(r1[0], r1[1]), (int)pack_hi16(r1[2], r1[3])};
    pl = i32x2{(int)pack_hi16(r2[0], r2[1]), (int)pack_hi16(r2[2], r2[3])};
}

#ifndef TWOG_X3_PRODUCTS
#define TWOG_X3_PRODUCTS 6
#endif
#ifndef TWOG_X3_TMPACC
#define TWOG_X3_TMPACC 0   // 1: 128x128 class with each k-step's products through a fresh accumulator, added by fp32 VALU adds (see
#endif                     // compute()): removes the accumulate bias, but at 128 VGPRs (two workgroups per CU) the temporary spills
// (TTMP, a template flag of the TT kernels selected at run time by TWOG_X3_DW_SPLIT_ACC=1: the same fresh-accumulator form with
// ONE workgroup per CU -- 256 VGPRs, no spill, four register stages kept. Same-sign K = 61 440: bias -2.07e-6 -> -4.5e-8, rms
// 2.08e-6 -> 9.3e-8 (the fp32-MFMA kernel: 1.07e-7); the dW launches +18 ... 25 %, the bs64 step 65.9 -> 68.9 ms: off by default,
// profiles/r05_dw_split_accumulator.txt)

constexpr int X3_PRODUCTS = TWOG_X3_PRODUCTS;  // chunk products per element product: 8 (exact to 2^-30) or 6 (drops m l, l m)
constexpr int X3_BK = 16;                    // one v_mfma_f32_32x32x16_bf16 k-step per k-tile
constexpr int X3_RROW = 32;                  // bytes per row of a [row][16 k] image; its two 16-byte chunks are swapped on rows
                                             // 16..31 (mod 32): conflict-free ds_read_b128 (lane groups of the guide) and ds_write_b64
constexpr int X3_RPLANE = 128 * X3_RROW;     // 4 096
constexpr int X3_TPLANE = X3_BK * 256;       // [16 k][128 rows] image: 4 096
constexpr int X3_STAGE = 6 * X3_RPLANE;      // one LDS stage (three planes of both operands): 24 576 bytes; two stages = 49 152
__device__ __forceinline__ int x3_swz(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

// cs (k-major A only): when cs_on, every thread also adds the A values it stages (4 consecutive tile columns of one k row per
// k-tile) into cs -- the column sums of A over this workgroup's k-range, finished by gemm_tile (twog_gemm_t::a_colsum).
// KU (round 6): k-tiles per barrier interval, KU sub-images per LDS stage -- for launches of at most one tile per CU (the
// segment level's per-step projection: 240 tiles), where ONE workgroup per CU moves through a barrier every 12 MFMAs per wave
// and nothing else hides the fragment reads and the barrier's skew. KU = 2 needs 2 x 2 x 24 KB of LDS and ~64 more registers:
// one workgroup per CU, which is what such a launch has anyway. Same MFMA sequence into the same accumulators: bit-identical.
// KS = 2 (round 6): SIXTEEN waves = two k-groups of eight, each with its own pair of LDS stages and its own half of the
// reduction (the caller passes the group's k-range; both halves hold the same number of k-tiles, so the workgroup-wide
// barriers match) -- four waves per SIMD for launches of at most ONE tile per CU, which is what the two co-resident
// workgroups of a big launch have and an 8-wave tile alone on its CU has not. gemm_tile adds the two partial tiles.
// PIPE (round 6): the fragment reads of k-tile t + 1 are issued BEFORE the MFMAs of k-tile t (two fragment sets in registers,
// three LDS stages: tile t + 1 is read and tile t + 2 stored while tile t is multiplied). Without it every wave of the
// workgroup reads its nine fragments right behind the barrier and multiplies afterwards: the CU's LDS phase (768 cycles per
// k-tile) and its MFMA phase (768 per SIMD) alternate. ~36 more registers: ONE workgroup per CU (the launches of at most one
// tile per CU). Same MFMA sequence into the same accumulators: bit-identical.
template <bool AKM, bool BKM, bool KG, bool TTMP = false, int KU = 1, int KS = 1, bool PIPE = false>
__device__ __forceinline__ void gemm_mainloop_x3(const twog_rows_t A, const twog_rows_t B, int M, int N, int m0, int n0,
                                                 int k_begin, int k_end, float* smem, f32x16 (&acc)[1][2], f32x4& cs, bool cs_on) {
    constexpr int BM = 128, BN = 128, NT = 512, XK = X3_BK;
    constexpr bool TMP = TWOG_X3_TMPACC || TTMP;
    static_assert(KU == 1 || (!AKM && !BKM && !KG && !TMP), "KU > 1: the row-major (forward) form only");
    static_assert(KS == 1 || (KS == 2 && KU == 1 && !AKM && !BKM && !KG && !TMP), "KS = 2: the row-major (forward) form only");
    constexpr int PA = AKM ? X3_TPLANE : X3_RPLANE, PB = BKM ? X3_TPLANE : X3_RPLANE;
    // stage b: A planes at b * X3_STAGE, B planes behind them (k-group 1: behind k-group 0's two stages)
    char* lds = reinterpret_cast<char*>(smem) + (KS == 2 ? (int)(threadIdx.x >> 9) * 2 * KU * X3_STAGE : 0);
    const int tid = KS == 2 ? (int)(threadIdx.x & 511) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
    using ARegs = TileRegs<(AKM ? XK : BM), (AKM ? BM : XK), NT>;
    using BRegs = TileRegs<(BKM ? XK : BN), (BKM ? BN : XK), NT>;
    static_assert(ARegs::PASSES == 1 && BRegs::PASSES == 1, "one 16-byte load per operand and thread");
    // global addressing as in gemm_mainloop's FAST path (branch-free buffer loads, clamped edges)
    uint32_t oa, ob;
    int sa_off, sb_off;   // LDS byte offset (stage 0, plane 0) of this thread's stores
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(A.ptr, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(B.ptr, 0, 0xffffffff, 0x00020000);
    {
        const int rr = tid / ARegs::F4_PER_ROW, cq = tid % ARegs::F4_PER_ROW;
        if constexpr (AKM) {   // rr = k row of the tile, cq = quad of tile rows
            oa = KG ? 4u * (uint32_t)min(m0 + cq * 4, M - 4) : 4u * (uint32_t)((int64_t)rr * A.ld_outer + min(m0 + cq * 4, M - 4));
            sa_off = 256 * rr + 16 * ((cq >> 1) ^ x3_swz(rr)) + 8 * (cq & 1);
        } else {               // rr = tile row, cq = quad of k
            oa = 4u * (uint32_t)(twog_row_off(A, min(m0 + rr, M - 1)) + cq * 4);
            sa_off = rr * X3_RROW + 16 * ((cq >> 1) ^ ((rr >> 4) & 1)) + 8 * (cq & 1);
        }
    }
    {
        const int rr = tid / BRegs::F4_PER_ROW, cq = tid % BRegs::F4_PER_ROW;
        if constexpr (BKM) {
            ob = KG ? 4u * (uint32_t)min(n0 + cq * 4, N - 4) : 4u * (uint32_t)((int64_t)rr * B.ld_outer + min(n0 + cq * 4, N - 4));
            sb_off = 3 * PA + 256 * rr + 16 * ((cq >> 1) ^ x3_swz(rr)) + 8 * (cq & 1);
        } else {
            ob = 4u * (uint32_t)(twog_row_off(B, min(n0 + rr, N - 1)) + cq * 4);
            sb_off = 3 * PA + rr * X3_RROW + 16 * ((cq >> 1) ^ ((rr >> 4) & 1)) + 8 * (cq & 1);
        }
    }
    struct Stage { f32x4 a, b; f32x4 a2[KU > 1 ? KU - 1 : 1], b2[KU > 1 ? KU - 1 : 1]; };   // (a2 / b2: k-tiles 1 .. KU-1 of the interval)
    // KG: a k-major operand whose rows (= k) are (outer, inner) grouped, e.g. "all but the first time step of every clip":
    // the (outer, inner) position of this thread's row is carried from k-tile to k-tile (k only moves forward; the clamped
    // tail repeats the last tile), no division in the loop. Offsets stay below 2^32 bytes (vec_ok).
    int ka_o = 0, ka_i = 0, kb_o = 0, kb_i = 0, ka_k = k_begin, kb_k = k_begin;
    const int a_in = A.inner <= 1 ? 0x7fffffff : A.inner, b_in = B.inner <= 1 ? 0x7fffffff : B.inner;
    const uint32_t a_ldi = (uint32_t)(A.inner <= 1 ? A.ld_outer : A.ld_inner), b_ldi = (uint32_t)(B.inner <= 1 ? B.ld_outer : B.ld_inner);
    if constexpr (KG) {
        const int ra_ = k_begin + tid / ARegs::F4_PER_ROW, rb_ = k_begin + tid / BRegs::F4_PER_ROW;
        ka_o = A.inner <= 1 ? 0 : ra_ / A.inner; ka_i = A.inner <= 1 ? ra_ : ra_ - ka_o * A.inner;
        kb_o = B.inner <= 1 ? 0 : rb_ / B.inner; kb_i = B.inner <= 1 ? rb_ : rb_ - kb_o * B.inner;
    }
    auto gload = [&](Stage& r, int k0) {
        if constexpr (KG && AKM) {
            ka_i += k0 - ka_k;
            ka_k = k0;
            while (ka_i >= a_in) { ka_i -= a_in; ++ka_o; }
            const uint32_t off = 4u * ((uint32_t)ka_o * (uint32_t)A.ld_outer + (uint32_t)ka_i * a_ldi) + oa;
            r.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)off, 0, 0));
        } else {
            const int sa = (int)(AKM ? (uint32_t)k0 * (uint32_t)A.ld_outer * 4u : (uint32_t)k0 * 4u);
            r.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)oa, sa, 0));
        }
        if constexpr (KG && BKM) {
            kb_i += k0 - kb_k;
            kb_k = k0;
            while (kb_i >= b_in) { kb_i -= b_in; ++kb_o; }
            const uint32_t off = 4u * ((uint32_t)kb_o * (uint32_t)B.ld_outer + (uint32_t)kb_i * b_ldi) + ob;
            r.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)off, 0, 0));
        } else {
            const int sb = (int)(BKM ? (uint32_t)k0 * (uint32_t)B.ld_outer * 4u : (uint32_t)k0 * 4u);
            r.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)ob, sb, 0));
        }
        if constexpr (KU > 1) {
#pragma unroll
            for (int u = 1; u < KU; ++u) {
                const int su = (int)((uint32_t)(k0 + u * XK) * 4u);
                r.a2[u - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)oa, su, 0));
                r.b2[u - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)ob, su, 0));
            }
        }
    };
    const int nkt = (k_end - k_begin) / (XK * KU);
    auto split_store = [&](const Stage& r, int buf, int t) {   // t: index of the k-tile held by r (clamped repeats: t >= nkt)
        char* base = lds + buf * KU * X3_STAGE;
        i32x2 ph, pm, pl;
        if constexpr (AKM && BKM && !KG) {
            if (cs_on && t < nkt) cs += r.a;
        }
        split3(r.a, ph, pm, pl);
        *reinterpret_cast<i32x2*>(base + sa_off) = ph;
        *reinterpret_cast<i32x2*>(base + sa_off + PA) = pm;
        *reinterpret_cast<i32x2*>(base + sa_off + 2 * PA) = pl;
        split3(r.b, ph, pm, pl);
        *reinterpret_cast<i32x2*>(base + sb_off) = ph;
        *reinterpret_cast<i32x2*>(base + sb_off + PB) = pm;
        *reinterpret_cast<i32x2*>(base + sb_off + 2 * PB) = pl;
        if constexpr (KU > 1) {
#pragma unroll
            for (int u = 1; u < KU; ++u) {
                char* bu = base + u * X3_STAGE;
                split3(r.a2[u - 1], ph, pm, pl);
                *reinterpret_cast<i32x2*>(bu + sa_off) = ph;
                *reinterpret_cast<i32x2*>(bu + sa_off + PA) = pm;
                *reinterpret_cast<i32x2*>(bu + sa_off + 2 * PA) = pl;
                split3(r.b2[u - 1], ph, pm, pl);
                *reinterpret_cast<i32x2*>(bu + sb_off) = ph;
                *reinterpret_cast<i32x2*>(bu + sb_off + PB) = pm;
                *reinterpret_cast<i32x2*>(bu + sb_off + 2 * PB) = pl;
            }
        }
    };
    // fragment addressing: lane l = (r = l & 31 row / column of the 32x32 block, h = l >> 5 half of the 16-deep k-step)
    const int r32 = lane & 31, h = lane >> 5;
    // row-major image: 16 bytes (k = 8 h .. 8 h + 7) of row (block row + r)
    const int fa_r = (wm + r32) * X3_RROW + 16 * (h ^ ((r32 >> 4) & 1));
    const int fb_r = 3 * PA + (wn + r32) * X3_RROW + 16 * (h ^ ((r32 >> 4) & 1));
    // k-major image, transposed read: lane 4q + p of a 16-lane group g supplies block row k = 8 (g >> 1) + q (+ 4 for the second
    // read), rows 16 (g & 1) + 4p .. 4p + 3 of the block: chunk (block row / 8) + 2 (g & 1) + (p >> 1), byte 8 (p & 1)
    const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int tk0 = 8 * (g16 >> 1) + q4, tk1 = tk0 + 4;
    const int ta0 = 256 * tk0 + 8 * (p4 & 1), ta1 = 256 * tk1 + 8 * (p4 & 1);   // + 16 * (chunk ^ swz)
    const int tchA = (wm >> 3) + 2 * (g16 & 1) + (p4 >> 1);
    const int tchB = (wn >> 3) + 2 * (g16 & 1) + (p4 >> 1);                       // block b: + 4 b
    const int sw0 = x3_swz(tk0), sw1 = x3_swz(tk1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto frag_r = [&](const char* base, int off) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4*>(base + off));
    };
    auto frag_t = [&](const char* base, int tch) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + ta0 + 16 * (tch ^ sw0)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + ta1 + 16 * (tch ^ sw1)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto compute = [&](int buf) {
#pragma unroll
      for (int u_ = 0; u_ < KU; ++u_) {
        const char* base = lds + (buf * KU + u_) * X3_STAGE;
        // X3_PRODUCTS chunk products per block, smallest terms first. 6 (shipped): everything >= 2^-16 |a b|; 8 (TWOG_X3_PRODUCTS=8
        // at build time): also m l and l m -- measured identical to three digits (profiles/r04_x3_products_6_vs_8*.txt).
        // WHERE the products are added matters more than how many there are: the bf16 MFMA's accumulate is not a
        // round-to-nearest fp32 add -- every v_mfma_f32_32x32x16_bf16 into a LARGE accumulator loses ~2^-29 of it towards
        // zero (tools/x3_bias_probe.py: on same-sign operands six accumulations per 16 k into the running sum gave a
        // relative bias of -4.3e-7 at K = 1 536 and -2.1e-6 at K = 61 440, where the fp32 MFMA has 4e-10). TMPACC (build-time
        // option TWOG_X3_TMPACC=1; NOT the default: with four register stages the kernel sits at its 128-VGPR budget and the
        // 16-register temporary spills 360-570 bytes per lane into scratch -- profiles/HISTORY.md section 8):
        // the products of one k-step are chained through a FRESH accumulator (C = 0: its roundings are relative to one
        // k-step's partial sum) and that partial sum is added to the running sum by 16 fp32 VALU adds per block (round to
        // nearest even, unbiased); the B fragments of a block are read right before its chain, so the temporary takes the
        // registers the second block's fragments held. Default (0): all products into the running sum (round 3's form).
        constexpr int PI[8] = {2, 1, 2, 0, 1, 1, 0, 0}, PJ[8] = {1, 2, 0, 2, 1, 0, 1, 0};
        constexpr int FIRST = 8 - X3_PRODUCTS;
        static_assert(X3_PRODUCTS == 8 || X3_PRODUCTS == 6, "l m and m l are the two optional products");
        bf16x8 af[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = AKM ? frag_t(base + p * PA, tchA) : frag_r(base + p * PA, fa_r);
        if constexpr (TMP) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            bf16x8 bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[p] = BKM ? frag_t(base + 3 * PA + p * PB, tchB + 4 * b) : frag_r(base + p * PB, fb_r + b * 32 * X3_RROW);
            f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PI[FIRST]], bf[PJ[FIRST]], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int q = FIRST + 1; q < 8; ++q) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PI[q]], bf[PJ[q]], t, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][b][i] += t[i];
            __builtin_amdgcn_sched_barrier(0);   // one temporary: the compiler must not run the two blocks' chains side by side
        }
        } else {
        bf16x8 bf[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[b][p] = BKM ? frag_t(base + 3 * PA + p * PB, tchB + 4 * b) : frag_r(base + p * PB, fb_r + b * 32 * X3_RROW);
#pragma unroll
        for (int t = FIRST; t < 8; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PI[t]], bf[b][PJ[t]], acc[0][b], 0, 0, 0);
        }
      }
    };
    // Four register stages and two LDS stages, one barrier per k-tile: the loads of k-tile t + 4 are issued before the MFMAs
    // of tile t (a k-tile of 16 lasts under two microseconds at this matrix rate: two stages do not cover a miss to HBM);
    // tile t + 1 is split and stored into the other LDS stage in the same basic block as the MFMAs of tile t (the VALU work
    // of the split issues between the MFMAs). Loads past the last k-tile are clamped to it instead of branched around (see
    // gemm_mainloop); the redundant tiles are stored but never read.
    if (nkt <= 0) return;
    const int k_last = k_begin + (nkt - 1) * XK * KU;
    auto kof = [&](int t) { return min(k_begin + t * XK * KU, k_last); };
    if constexpr (PIPE) {
        static_assert(!PIPE || (KU == 1 && KS == 1 && !TMP), "PIPE: one k-tile per interval, one k-group, products into the running sum");
        constexpr int PI[8] = {2, 1, 2, 0, 1, 1, 0, 0}, PJ[8] = {1, 2, 0, 2, 1, 0, 1, 0};
        constexpr int FIRST = 8 - X3_PRODUCTS;
        auto frags = [&](int buf, bf16x8 (&af)[3], bf16x8 (&bf)[2][3]) {
            const char* base = lds + buf * X3_STAGE;
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p] = AKM ? frag_t(base + p * PA, tchA) : frag_r(base + p * PA, fa_r);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[b][p] = BKM ? frag_t(base + 3 * PA + p * PB, tchB + 4 * b) : frag_r(base + p * PB, fb_r + b * 32 * X3_RROW);
        };
        auto mfmas = [&](const bf16x8 (&af)[3], const bf16x8 (&bf)[2][3]) {
#pragma unroll
            for (int t = FIRST; t < 8; ++t)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PI[t]], bf[b][PJ[t]], acc[0][b], 0, 0, 0);
        };
        Stage r0, r1, r2, r3;
        gload(r0, kof(0));
        gload(r1, kof(1));
        gload(r2, kof(2));
        gload(r3, kof(3));
        split_store(r0, 0, 0);
        __syncthreads();
        bf16x8 fa0[3], fb0[2][3], fa1[3], fb1[2][3];
        frags(0, fa0, fb0);
        split_store(r1, 1, 1);
        gload(r0, kof(4));
        gload(r1, kof(5));
        __syncthreads();
        // iteration t: fragment set "cur" holds tile t, LDS stage s1 tile t + 1; tile t + 2 goes from its registers into stage s2
        int s1 = 1, s2 = 2;
        auto rot = [&]() { s1 = s2; s2 = s2 == 2 ? 0 : s2 + 1; };
        int kt = 0;
        for (; kt + 3 < nkt; kt += 4) {
            frags(s1, fa1, fb1); mfmas(fa0, fb0); split_store(r2, s2, kt + 2); gload(r2, kof(kt + 6)); __syncthreads(); rot();
            frags(s1, fa0, fb0); mfmas(fa1, fb1); split_store(r3, s2, kt + 3); gload(r3, kof(kt + 7)); __syncthreads(); rot();
            frags(s1, fa1, fb1); mfmas(fa0, fb0); split_store(r0, s2, kt + 4); gload(r0, kof(kt + 8)); __syncthreads(); rot();
            frags(s1, fa0, fb0); mfmas(fa1, fb1); split_store(r1, s2, kt + 5); gload(r1, kof(kt + 9)); __syncthreads(); rot();
        }
        // up to three k-tiles left: set 0 holds tile kt, stage s1 tile kt + 1, r2 tile kt + 2
        if (kt < nkt) {
            if (kt + 1 < nkt) frags(s1, fa1, fb1);
            mfmas(fa0, fb0);
            if (kt + 2 < nkt) { split_store(r2, s2, kt + 2); __syncthreads(); }
        }
        if (kt + 1 < nkt) {
            if (kt + 2 < nkt) frags(s2, fa0, fb0);
            mfmas(fa1, fb1);
        }
        if (kt + 2 < nkt) mfmas(fa0, fb0);
        return;
    }
#if TWOG_X3_TMPACC
    // (TMPACC: two register stages -- the temporary accumulator takes the registers of the other two)
    Stage r0, r1;
    gload(r0, kof(0));
    gload(r1, kof(1));
    split_store(r0, 0, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        gload(r0, kof(kt + 2)); compute(0); split_store(r1, 1, kt + 1); __syncthreads();
        gload(r1, kof(kt + 3)); compute(1); split_store(r0, 0, kt + 2); __syncthreads();
    }
    if (kt < nkt) compute(0);
#else
    Stage r0, r1, r2, r3;
    gload(r0, kof(0));
    gload(r1, kof(1));
    gload(r2, kof(2));
    gload(r3, kof(3));
    split_store(r0, 0, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 3 < nkt; kt += 4) {
        gload(r0, kof(kt + 4)); compute(0); split_store(r1, 1, kt + 1); __syncthreads();
        gload(r1, kof(kt + 5)); compute(1); split_store(r2, 0, kt + 2); __syncthreads();
        gload(r2, kof(kt + 6)); compute(0); split_store(r3, 1, kt + 3); __syncthreads();
        gload(r3, kof(kt + 7)); compute(1); split_store(r0, 0, kt + 4); __syncthreads();
    }
    // up to three k-tiles left: tile kt sits in LDS stage 0, tiles kt + 1, kt + 2 in r1, r2
    if (kt < nkt) compute(0);
    if (kt + 1 < nkt) { split_store(r1, 1, kt + 1); __syncthreads(); compute(1); }
    if (kt + 2 < nkt) { split_store(r2, 0, kt + 2); __syncthreads(); compute(0); }
#endif
}

// X3 for the 64-row tile class (the recurrent chains at a real batch: 160 ... 960 tiles of K = 512 ... 1 536, MFMA-bound
// in fp32: profiles/HISTORY.md section 8): the same exact three-way bf16 split and six chunk products as gemm_mainloop_x3, for
//   64 x 64 tiles with 4 waves (KS = 1, k-tiles of 16) or 8 waves = two k-groups (KS = 2, k-tiles of 32: group g multiplies
//   k-step g), and the fused GRU forward step's 64 x 192 gate-aware tiles (G3, three accumulators per wave).
// A is always row-major here (previous states / gradients of a chain step); B is row-major ([N][K] weights: forward) or
// k-major ([K][N]: the backward chains multiply by W, not W^T). Images: row-major operands [row][XK k] with 32- or 64-byte
// rows whose 16-byte chunks are XOR-swizzled by the row (conflict-free ds_read_b128 for the hardware's lane groups and
// conflict-free ds_write_b64); the k-major operand [k][64 rows] in 128-byte rows, chunks swizzled by the k row, read
// through ds_read_b64_tr_b16.
#ifndef TWOG_X3S_RS
#define TWOG_X3S_RS 4
#endif
// cache policy of the chain kernels' operand loads (the aux field of buffer_load: 0 default, 2 = nt: streamed, evicted first).
// A of a chain launch is read once per launch (previous states / gradients), B is the weight every step comes back to.
#ifndef TWOG_X3S_A_AUX
#define TWOG_X3S_A_AUX 0
#endif
#ifndef TWOG_X3S_B_AUX
#define TWOG_X3S_B_AUX 0
#endif
// Diagnostic build only (-DTWOG_STAMPS, tools/stamps_probe.sh): cycle stamps at the phase boundaries of the X3 chain loop,
// summed per wave of workgroup 0 and written to a buffer of their own at the end (cdna_hip_programming.md, In-kernel stamps).
#ifdef TWOG_STAMPS
__device__ unsigned long long twog_stamp_buf[16 * 8];
#define TWOG_STAMP_DECL unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = 0; { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_last = t_; }
#define TWOG_STAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_acc[i] += t_ - st_last; st_last = t_; __builtin_amdgcn_sched_barrier(0); }
#define TWOG_STAMP_FLUSH if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 6; ++i_) twog_stamp_buf[(threadIdx.x >> 6) * 8 + i_] = st_acc[i_]; }
#else
#define TWOG_STAMP_DECL
#define TWOG_STAMP(i)
#define TWOG_STAMP_FLUSH
#endif
template <int BM, int BN, int NT, bool BKM, int KS, int TN, bool G3, int RS = 4, bool LO2 = (TN == 1), int KU = 1>
__device__ __forceinline__ void gemm_mainloop_x3s(const twog_rows_t A, const twog_rows_t B, int M, int N, int m0, int n0,
                                                  int k_begin, int k_end, float* smem, f32x16 (&acc)[1][TN]) {
    constexpr int XK = 16 * KS, NTG = NT / KS, RB = 2 * XK;       // k-tile depth, threads per k-group, bytes per image row
    constexpr int WM = BM / (NTG / 128), WN = BN / 2;
    static_assert(WM == 32 && WN == 32 * TN && (!BKM || (BN == 64 && !G3)), "one 32-row block per wave, TN column blocks");
    constexpr int PA = BM * RB, PB = BN * RB, STAGE = 3 * PA + 3 * PB;
    constexpr int FA = BM * XK / 4, FB = BN * XK / 4;             // 16-byte loads per k-tile
    constexpr int NPA = (FA + NT - 1) / NT, NPB = (FB + NT - 1) / NT;
    static_assert(FA % NT == 0 || FA < NT, "whole passes");
    char* lds = reinterpret_cast<char*>(smem);
    TWOG_STAMP_DECL
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % (NTG / 64), kgrp = (tid >> 6) / (NTG / 64);
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    auto swz_r = [](int row, int chunk) { return XK == 32 ? (chunk ^ ((row >> 2) & 3)) : (chunk ^ ((row >> 4) & 1)); };
    auto swz_t = [](int k, int chunk) { return chunk ^ (((k >> 1) & 1) << 2); };
    uint32_t oa[NPA], ob[NPB];
    int sa_off[NPA], sb_off[NPB];
    bool a_on[NPA], b_on[NPB];
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(A.ptr, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(B.ptr, 0, 0xffffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int f = tid + i * NT;
        a_on[i] = f < FA;
        const int rr = (f % FA) / (XK / 4), cq = f % (XK / 4);
        oa[i] = 4u * (uint32_t)(twog_row_off(A, min(m0 + rr, M - 1)) + cq * 4);
        sa_off[i] = rr * RB + 16 * swz_r(rr, cq >> 1) + 8 * (cq & 1);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int f = tid + i * NT;
        b_on[i] = f < FB;
        if constexpr (BKM) {
            const int kk = (f % FB) / (BN / 4), cq = f % (BN / 4);   // k row of the tile, quad of tile columns
            ob[i] = 4u * (uint32_t)((int64_t)kk * B.ld_outer + min(n0 + cq * 4, N - 4));
            sb_off[i] = 3 * PA + kk * 2 * BN + 16 * swz_t(kk, cq >> 1) + 8 * (cq & 1);
        } else {
            const int rr = (f % FB) / (XK / 4), cq = f % (XK / 4);
            int brow;
            if constexpr (G3) {   // tile row rr -> gate (rr % 96) / 32 of unit n0 + 32 (rr / 96) + rr % 32 (see gemm_mainloop)
                const int gate = (rr % 96) / 32, unit = n0 + 32 * (rr / 96) + (rr & 31);
                brow = gate * N + min(unit, N - 1);
            } else {
                brow = min(n0 + rr, N - 1);
            }
            ob[i] = 4u * (uint32_t)(twog_row_off(B, brow) + cq * 4);
            sb_off[i] = 3 * PA + rr * RB + 16 * swz_r(rr, cq >> 1) + 8 * (cq & 1);
        }
    }
    // KU k-tiles per barrier interval (KU sub-images per LDS stage): a chain launch runs ONE workgroup per CU whose waves
    // move through the barriers in step, and a k-tile of 16 KS carries only six MFMAs per wave -- the fixed part of an
    // interval (fragment reads after the barrier, the store -> barrier -> load round trip, the barrier's skew) is then most
    // of it. Same MFMA sequence into the same accumulators: bit-identical for every KU.
    struct Stage { f32x4 a[KU][NPA], b[KU][NPB]; };
    auto gload = [&](Stage& r, int k0) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int sa = (int)((uint32_t)(k0 + u * XK) * 4u);
            const int sb = (int)(BKM ? (uint32_t)(k0 + u * XK) * (uint32_t)B.ld_outer * 4u : (uint32_t)(k0 + u * XK) * 4u);
#pragma unroll
            for (int i = 0; i < NPA; ++i) r.a[u][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)oa[i], sa, TWOG_X3S_A_AUX));
#pragma unroll
            for (int i = 0; i < NPB; ++i) r.b[u][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)ob[i], sb, TWOG_X3S_B_AUX));
        }
    };
    auto split_store = [&](const Stage& r, int buf) {
        i32x2 ph, pm, pl;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            char* base = lds + (buf * KU + u) * STAGE;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                if (NPA * NT == FA || a_on[i]) {
                    split3(r.a[u][i], ph, pm, pl);
                    *reinterpret_cast<i32x2*>(base + sa_off[i]) = ph;
                    *reinterpret_cast<i32x2*>(base + sa_off[i] + PA) = pm;
                    *reinterpret_cast<i32x2*>(base + sa_off[i] + 2 * PA) = pl;
                }
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                if (NPB * NT == FB || b_on[i]) {
#ifdef TWOG_PROBE_NO_BSPLIT   // timing probe only (wrong results): what the split of the B operand costs a chain launch
                    ph = i32x2{__builtin_bit_cast(int, r.b[u][i][0]), __builtin_bit_cast(int, r.b[u][i][1])};
                    pm = i32x2{__builtin_bit_cast(int, r.b[u][i][2]), __builtin_bit_cast(int, r.b[u][i][3])};
                    pl = ph;
#else
                    split3(r.b[u][i], ph, pm, pl);
#endif
                    *reinterpret_cast<i32x2*>(base + sb_off[i]) = ph;
                    *reinterpret_cast<i32x2*>(base + sb_off[i] + PB) = pm;
                    *reinterpret_cast<i32x2*>(base + sb_off[i] + 2 * PB) = pl;
                }
            }
        }
    };
    const int r32 = lane & 31, h = lane >> 5;
    const int chunk = (XK == 32 ? 2 * kgrp : 0) + h;                 // this lane's 16 bytes (8 k) of its k-group's k-step
    const int fa_r = (wm + r32) * RB + 16 * swz_r(wm + r32, chunk);
    int fb_r[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) fb_r[b] = 3 * PA + (wn + 32 * b + r32) * RB + 16 * swz_r(wn + 32 * b + r32, chunk);
    const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int tk0 = 16 * kgrp + 8 * (g16 >> 1) + q4, tk1 = tk0 + 4;
    const int tch = (wn >> 3) + 2 * (g16 & 1) + (p4 >> 1);
    const int ft0 = 3 * PA + tk0 * 2 * BN + 16 * swz_t(tk0, tch) + 8 * (p4 & 1);
    const int ft1 = 3 * PA + tk1 * 2 * BN + 16 * swz_t(tk1, tch) + 8 * (p4 & 1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    f32x16 lo[LO2 ? TN : 1];
#pragma unroll
    for (int b = 0; b < (LO2 ? TN : 1); ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) lo[b][i] = 0.0f;
    auto compute = [&](int buf) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const char* base = lds + (buf * KU + u) * STAGE;
        bf16x8 af[3], bf[TN][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4*>(base + p * PA + fa_r));
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if constexpr (BKM) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + p * PB + ft0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + p * PB + ft1));
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bf[b][p] = __builtin_bit_cast(bf16x8, v);
                } else {
                    bf[b][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4*>(base + p * PB + fb_r[b]));
                }
            }
#ifdef TWOG_STAMPS
        asm volatile("" :: "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(bf[0][0]), "v"(bf[0][1]), "v"(bf[0][2]));
        TWOG_STAMP(4)
#endif
        // h h goes to the main accumulator, the five small products (2^-8 ... 2^-16 of it) to a second one that is added
        // once after the loop: the main accumulator is rounded once per 16 k instead of six times, and the roundings
        // of the small one are 2^-8 of an ulp of the result.
        // (LO2 = false -- the three-accumulator GRU tile, whose register file is full: all six into the main accumulator,
        // small products first, as the 128x128 class does)
        constexpr int PI[5] = {2, 0, 1, 1, 0}, PJ[5] = {0, 2, 1, 0, 1};   // l h, h l, m m, m h, h m
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                if constexpr (LO2) lo[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PI[t]], bf[b][PJ[t]], lo[b], 0, 0, 0);
                else acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PI[t]], bf[b][PJ[t]], acc[0][b], 0, 0, 0);
            }
#pragma unroll
        for (int b = 0; b < TN; ++b)
            acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[b][0], acc[0][b], 0, 0, 0);
      }
    };
    const int nkt = (k_end - k_begin) / (XK * KU);
    if (nkt <= 0) return;
    const int k_last = k_begin + (nkt - 1) * XK * KU;
    auto kof = [&](int t) { return min(k_begin + t * XK * KU, k_last); };
    // RS register stages in a ring (loads of k-tile t + RS are issued before the MFMAs of tile t), two LDS stages, one
    // barrier per k-tile. The ring is unrolled so that every stage keeps a static name. RS = 2 for the kernels whose
    // epilogue operands already fill the register file; 4 by default; 8 where a lone tile per CU is bound by how many
    // bytes it keeps in flight (chain launches: ~30 GB/s per CU with 4 stages).
    Stage r[RS];
#pragma unroll
    for (int i = 0; i < RS; ++i) gload(r[i], kof(i));
    split_store(r[0], 0);
    __syncthreads();
    int kt = 0;
    for (; kt + RS - 1 < nkt; kt += RS) {
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            gload(r[i], kof(kt + RS + i));
            TWOG_STAMP(0)
            compute(i & 1);
            TWOG_STAMP(1)
            split_store(r[(i + 1) % RS], (i + 1) & 1);
            TWOG_STAMP(2)
            __syncthreads();
            TWOG_STAMP(3)
        }
    }
    TWOG_STAMP_FLUSH
    // up to RS - 1 k-tiles left: tile kt sits in LDS stage 0, tiles kt + j in r[j]
#pragma unroll
    for (int j = 0; j < RS - 1; ++j) {
        if (kt + j < nkt) {
            compute(j & 1);
            if (kt + j + 1 < nkt) { split_store(r[j + 1], (j + 1) & 1); __syncthreads(); }
        }
    }
    if constexpr (LO2) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][b][i] += lo[b][i];
    }
}

// XS (recurrent-chain launches with fewer tiles than the chip has CUs): the reduction is ALSO split over workgroups
// (blockIdx.y = k-slice) and combined inside the launch, without a grid barrier and without waiting: every workgroup
// writes its partial tile write-through (16-byte sc1 stores, so no release fence), drains them, and one lane draws an
// agent-scope ticket; the workgroup whose ticket is the last one re-reads all slices' partials with sc1 loads (they
// bypass its L1: no acquire), adds them in SLICE order -- bit-identical whoever arrives last -- and runs the epilogue
// (bias / accumulate / the fused gate backward) exactly as the unsplit kernel does. It also resets the ticket, so the
// counters are zero again at the launch boundary (graph replays need no memset node). The hand-off form is the
// guide's split-K recipe (cdna_hip_programming.md section 5 item 2 / MI355X_MICROARCH.md "Valid forms": sc1 payload,
// every storing wave drained, workgroup barrier, one relaxed agent atomic; the last arriver's loads all sc1).
template <int BM, int BN, int NT, bool AKM, bool BKM, int D, bool KG, bool GATE, int KS = 1, bool XS = false, bool X3 = false, int KU = 1, bool TTMP = false, bool PIPE = false>
__device__ __forceinline__ void gemm_tile(const Group& g, const GateArgs* ga) {
    constexpr int NTG = NT / KS;
    constexpr int WM = BM / (NTG / 128), WN = BN / 2;  // per-wave tile; the waves of a k-group in a (NTG/128) x 2 grid
    constexpr int TM = WM / 32, TN = WN / 32;  // 32x32 MFMA tiles per wave
    constexpr int LDA = AKM ? (BM + 4) : (BK + 4);
    constexpr int LDB = BKM ? (BN + 4) : (BK + 4);
    constexpr int A_ELEMS = AKM ? BK * LDA : BM * LDA;
    constexpr int B_ELEMS = BKM ? BK * LDB : BN * LDB;
    // (X3 on the 64-row class: two stages of three bf16 planes per operand, 2 x 6 x 64 rows x 2 XK bytes, XK = 16 KS)
    constexpr int SMEM_FLOATS = (X3 && BM == 128 && BN == 64) ? 2 * KU * 3 * (128 + 64) * 32 / 4   // 128x64 chain tile: 2 stages x KU images
                                : (X3 && BM == 128) ? (PIPE ? 3 : KS * 2 * KU) * X3_STAGE / 4
                                : (X3 && BM == 64 && 12 * 64 * 8 * KS * KU > 2 * (A_ELEMS + B_ELEMS)) ? 12 * 64 * 8 * KS * KU : 2 * (A_ELEMS + B_ELEMS);
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];

    // XCD-aware, bijective block -> (problem, tile) map. Blocks are dealt round-robin over the 8 XCDs (speed only:
    // correctness never depends on it), so XCD x = blockIdx & 7 owns the blocks with local index l = blockIdx >> 3.
    // Problems with the same reduction length form a class (the host sorts them adjacent, longest first); the tile list
    // of every class is cut into 8 contiguous chunks, one per XCD (its L2 then serves the row / column panels the
    // chunk shares), and each XCD walks the classes in order: all XCDs get the same share of long and short tiles, so
    // heterogeneous groups stay balanced. Chunk sizes differ by at most one tile; the classes' remainders are dealt
    // around the XCD ring one after the other (cls_rot), which makes the per-XCD totals match the hardware's deal.
    int bid = 0;  // logical tile id
    int split = blockIdx.y;
    if (!XS && g.xcd_split) {
        // Tall reductions (dW = dY^T X: few tiles, many k-splits): deal whole K-SPLITS to XCDs instead of tile chunks. The
        // tiles of one split read the same k-range of both operands; on one XCD, started together, they walk that range
        // in step and its L2 serves every row / column panel chunk once -- dealt by tile chunks every XCD needs (almost)
        // every chunk panel of every split (measured: 2.1 x the operand bytes at the L2<->fabric boundary). Workgroups
        // are dispatched round-robin over the XCDs in linear order, so XCD x = L & 7 owns the local indices j = L >> 3;
        // job j of XCD x is tile j % tiles of split x + 8 (j / tiles). Bijective because splitk % 8 == 0 (host).
        const int L = blockIdx.y * gridDim.x + blockIdx.x, x = L & 7, j = L >> 3, tiles = gridDim.x;
        const int sj = j / tiles;
        split = x + 8 * sj;
        bid = j - sj * tiles;
    } else {
        const int x = blockIdx.x & 7;
        int l = blockIdx.x >> 3;
#pragma unroll 1
        for (int c = 0; c < g.n_cls; ++c) {
            const int nt = g.cls_ntiles[c], q = nt >> 3, r = nt & 7, o = (x - g.cls_rot[c]) & 7;
            const int cnt = q + (o < r ? 1 : 0);
            if (l < cnt) {
                bid = g.cls_start[c] + o * q + min(o, r) + l;
                break;
            }
            l -= cnt;
        }
    }
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (bid >= g.p[i].tile_start) pi = i;
    const Prob& G = g.p[pi];
    int tile = bid - G.tile_start;
    const int per_batch = G.tiles_m * G.tiles_n, bi = tile / per_batch;
    tile -= bi * per_batch;
    twog_rows_t A = G.A, B = G.B, C = G.C;
    A.ptr += bi * G.a_bs;
    B.ptr += bi * G.b_bs;
    C.ptr += bi * G.c_bs;
    const int M = G.M, N = G.N, K = G.K, a_vec = G.a_vec, b_vec = G.b_vec, act = G.act, accumulate = G.accumulate;
    const float* bias = G.bias;
    int tm_idx, tn_idx;
    tile_coords(G, tile, g.group, tm_idx, tn_idx);
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;
    const int k_begin = split * g.k_per_split;
    const int k_end = min(K, k_begin + g.k_per_split);
    // XS: slices of THIS problem (a grouped launch mixes reduction lengths; the grid has the longest one's count)
    const int xs_slices = XS ? min(g.splitk, (K + g.k_per_split - 1) / g.k_per_split) : 1;
    if (XS && split >= xs_slices) return;   // whole workgroup, before any barrier

    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % (NTG / 64), kgrp = (threadIdx.x >> 6) / (NTG / 64);
    const int li = lane & 31, kh = lane >> 5;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    static_assert(KS == 1 || (KS == 2 && (TM * TN == 1 || (X3 && BM == 128 && BN == 128))), "the in-workgroup k-split: the 64x64 class and the X3 128x128 tile");

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // epilogue operands are fetched BEFORE the reduction loop so their latency hides under it (it is a visible share of
    // the short recurrent launches: 16 k-tiles): the bias of this lane's columns and, for the single-accumulator
    // 64x64 class, the previous C values of accumulate launches. XS kernels fetch them only in the workgroup that
    // combines the slices (next to its slab loads), not in every slice.
    constexpr bool PREFETCH_C = (TM * TN == 1);
    // LA: the deterministic split-K of the non-chain launches (tall dW reductions) combined INSIDE the launch by the last
    // arriver of each tile, as the XS kernels do (round 5; built, bit-identical to the two-launch form,
    // measured slower and left OFF: see prepare_group). Tickets: g.xcnt (host: the first 16 KB of the split-K workspace).
    const bool la = !XS && g.splitk > 1 && g.xcnt != nullptr;
    const bool epi_here = XS || g.splitk == 1 || la;
    float bv[TN];
    float cprev[PREFETCH_C ? 16 : 1];
    // fused gate epilogue: its operands do not depend on this launch, so they are requested here, next to the first
    // operand tiles, and arrive under the reduction loop (the tile has the registers: these launches run one or two
    // workgroups per CU). Addressing as checked by the host (twog_internal_gemm_gate_bwd): dh, save, h_prev, dgi, dgh and
    // u share one (outer, inner) row grouping, resolved per row with an exact float reciprocal (rows < 2^22); every
    // offset fits 31 bits; absent operands are redirected to a valid address and masked afterwards: no branch and no
    // division between the loads.
    constexpr int GN = GATE ? 16 : 1;
    float g_dh[GN], g_rg[GN], g_z[GN], g_n[GN], g_hn[GN], g_h0[GN], g_uu[GN];
    int gidx = -1;
    if constexpr (GATE) gidx = ga->gate_of[pi];
    auto fetch_epilogue = [&]() {
#pragma unroll
        for (int b = 0; b < TN; ++b)
            bv[b] = (PREFETCH_C && bias && epi_here && kgrp == 0) ? bias[min(n0 + wn + b * 32 + li, N - 1)] : 0.f;
        if constexpr (PREFETCH_C) {
            if (accumulate && epi_here && kgrp == 0) {
                const int col = min(n0 + wn + li, N - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh, M - 1);
                    cprev[r] = C.ptr[twog_row_off(C, row) + col];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) cprev[r] = 0.f;
            }
        }
        if constexpr (GATE) {
            if (gidx >= 0 && kgrp == 0) {
                const twog_gru_step_bwd_t& S = ga->s[gidx];
                const int H = S.hidden;
                const int colc = min(n0 + wn + li, N - 1);
                const bool has_u = S.u != nullptr, has_hp = S.h_prev.ptr != nullptr;
                const int inner = S.dh.inner > 1 ? S.dh.inner : 1;
                const float inv_inner = 1.0f / (float)inner;
                const float* hp_ptr = has_hp ? S.h_prev.ptr : S.dh.ptr;
                const int hp_lo = has_hp ? (int)S.h_prev.ld_outer : (int)S.dh.ld_outer, hp_li = has_hp ? (int)S.h_prev.ld_inner : (int)S.dh.ld_inner;
                const float* u_ptr = has_u ? S.u : S.dh.ptr;
                const int u_lo = has_u ? (int)S.u_ld_outer : 0, u_li = has_u ? (int)S.u_ld_inner : 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowc = min(m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh, M - 1);
                    const int o = (int)(((float)rowc + 0.5f) * inv_inner), i = rowc - o * inner;
                    const float* sv = S.save.ptr + (o * (int)S.save.ld_outer + i * (int)S.save.ld_inner);
                    g_dh[r] = S.dh.ptr[o * (int)S.dh.ld_outer + i * (int)S.dh.ld_inner + colc];
                    g_rg[r] = sv[colc];
                    g_z[r] = sv[H + colc];
                    g_n[r] = sv[2 * H + colc];
                    g_hn[r] = sv[3 * H + colc];
                    g_h0[r] = hp_ptr[o * hp_lo + i * hp_li + colc];
                    g_uu[r] = u_ptr[o * u_lo + i * u_li];
                }
            }
        }
    };
    const bool early = XS ? (xs_slices == 1 || g.xs_early) : !la;
    if (early) fetch_epilogue();

    // uniform per workgroup: aligned operands and a reduction range made of whole k-tiles -> branch-free staging
    const bool fast = a_vec && b_vec && ((k_end - k_begin) % BK == 0);
    if constexpr (X3) {
        // (the host launches X3 kernels for aligned operands and whole k-tiles only)
        if constexpr (BM == 128 && BN == 64) {
            // 128 x 64 chain tile (round 5): eight waves of 32 x 32, row-major A -- a tile takes in (128 + 64) K operand values
            // for twice the products of a 64 x 64 tile's (64 + 64) K
            static_assert(!X3 || BM != 128 || BN != 64 || (NT == 512 && KS == 1 && !AKM && !KG && !XS && !GATE), "X3: 128x64 chain tiles");
            gemm_mainloop_x3s<BM, BN, NT, BKM, 1, 1, false, 2, true, KU>(A, B, M, N, m0, n0, k_begin, k_end, smem, acc);
        } else if constexpr (BM == 128) {
            static_assert(!X3 || BM != 128 || (BN == 128 && NT == 512 * KS && !GATE && !XS), "X3: the 8-wave (16-wave: KS = 2) 128x128 class");
            f32x4 cs = {0.f, 0.f, 0.f, 0.f};
            const bool cs_on = AKM && BKM && !KG && G.cs != nullptr && tn_idx == 0;   // uniform over the workgroup
            if constexpr (KS == 2) {   // (host: whole PAIRS of k-tiles, so both k-groups run the same number of barriers)
                const int half = (k_end - k_begin) / 2;
                gemm_mainloop_x3<AKM, BKM, KG, TTMP, KU, 2>(A, B, M, N, m0, n0, k_begin + kgrp * half, k_begin + (kgrp + 1) * half, smem, acc, cs, cs_on);
            } else
            gemm_mainloop_x3<AKM, BKM, KG, TTMP, KU, 1, PIPE>(A, B, M, N, m0, n0, k_begin, k_end, smem, acc, cs, cs_on);
            if constexpr (AKM && BKM && !KG) {
                if (cs_on) {
                    // thread (k row tid / 32, column quad tid % 32) holds its k rows' sums: the 16 k rows are added in row order
                    // (fixed: bit-reproducible); split-K launches park the slice's sums for splitk_reduce_kernel
                    __syncthreads();   // every wave is done with the operand tiles
                    *reinterpret_cast<f32x4*>(smem + (threadIdx.x >> 5) * 128 + (threadIdx.x & 31) * 4) = cs;
                    __syncthreads();
                    if (threadIdx.x < 128) {
                        float v = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) v += smem[r * 128 + threadIdx.x];
                        const int col = m0 + (int)threadIdx.x;
                        if (g.splitk > 1) g.cs_part[((int64_t)split * g.total_tiles + bid) * 128 + threadIdx.x] = v;
                        else if (col < M) G.cs[col] = G.cs_acc ? G.cs[col] + v : v;
                    }
                    __syncthreads();   // (the LA branch below reuses smem[0])
                }
            }
        } else {
            static_assert(!X3 || BM == 128 || (BM == 64 && BN == 64 && !AKM && !KG && !XS), "X3: 64x64 tiles, row-major A");
            gemm_mainloop_x3s<BM, BN, NT, BKM, KS, 1, false, (GATE || KU > 1 ? 2 : TWOG_X3S_RS), true, KU>(A, B, M, N, m0, n0, k_begin, k_end, smem, acc);
        }
    } else if (fast)
        gemm_mainloop<BM, BN, NT, AKM, BKM, true, TM, TN, D, KG, KS>(A, B, M, N, a_vec, b_vec, m0, n0, k_begin, k_end, smem, acc);
    else
        gemm_mainloop<BM, BN, NT, AKM, BKM, false, TM, TN, 1, false, KS>(A, B, M, N, a_vec, b_vec, m0, n0, k_begin, k_end, smem, acc);
    if constexpr (KS == 2) {
        // add the two k-groups' partial tiles (fixed order: group 0 + group 1) through LDS; group 0 runs the epilogue
        __syncthreads();   // every wave is done with the operand tiles
        float* red = smem + ((wave * TM * TN * 16) << 6) + lane;
        if (kgrp == 1) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((a * TN + b) * 16 + r) << 6] = acc[a][b][r];
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((a * TN + b) * 16 + r) << 6];
    }

    if constexpr (XS) {
        static_assert(!XS || (TM * TN == 1 && KS == 2), "the in-launch combine is written for the k-split 64x64 / 32x64 classes");
        const int S = xs_slices;
        if (S > 1) {   // uniform over the workgroup
            // partial tile of slice `split`: [wave][4 register quads][lane] x 16 bytes -- every store / load instruction
            // of a wave covers 1 KB of whole 128-byte lines; the layout is private to this kernel
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.slabs, 0, 0xffffffff, 0x00020000);
            const uint32_t tile_bytes = BM * BN * 4u;
            const uint32_t slice_stride = (uint32_t)g.total_tiles * tile_bytes;
            const int tile_off = (int)((uint32_t)bid * tile_bytes);
            const int lane_off = (int)(((uint32_t)wave * 4u * 64u + (uint32_t)lane) * 16u);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rs,
                                                       lane_off + q * 1024, tile_off + (int)((uint32_t)split * slice_stride), 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
            __syncthreads();
            unsigned* flag = reinterpret_cast<unsigned*>(smem + 4096);   // beyond the k-group exchange area
            if (threadIdx.x == 0)
                *flag = __hip_atomic_fetch_add(g.xcnt + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag != (unsigned)(S - 1)) return;
            if (threadIdx.x == 0) __hip_atomic_store(g.xcnt + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!early) fetch_epilogue();   // in flight together with the slab loads below
            f32x4 sum[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            int s0 = 0;
            for (; s0 + 1 < S; s0 += 2) {   // two slices' loads in flight; added in slice order
                f32x4 va[4], vb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    va[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + q * 1024, tile_off + (int)((uint32_t)s0 * slice_stride), 16));
                    vb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + q * 1024, tile_off + (int)((uint32_t)(s0 + 1) * slice_stride), 16));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { sum[q] += va[q]; sum[q] += vb[q]; }
            }
            if (s0 < S) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    sum[q] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + q * 1024, tile_off + (int)((uint32_t)s0 * slice_stride), 16));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][0][4 * q] = sum[q].x; acc[0][0][4 * q + 1] = sum[q].y; acc[0][0][4 * q + 2] = sum[q].z; acc[0][0][4 * q + 3] = sum[q].w;
            }
        }
    }

    if constexpr (!XS && KS == 1) {
        if (la) {   // uniform over the workgroup
            // partial tile of slice `split`: [wave][TM x TN x 4 register quads][lane] x 16 bytes -- every store / load
            // instruction of a wave covers 1 KB of whole 128-byte lines; the layout is private to this kernel
            constexpr int QN = TM * TN * 4;
            const int S = g.splitk;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.slabs, 0, 0xffffffff, 0x00020000);
            const uint32_t tile_bytes = BM * BN * 4u;
            const uint32_t slice_stride = (uint32_t)g.total_tiles * tile_bytes;   // (host: S * slice_stride < 2^32)
            const int tile_off = (int)((uint32_t)bid * tile_bytes);
            const int wv = (int)(threadIdx.x >> 6);
            const int lane_off = (int)(((uint32_t)wv * QN * 64u + (uint32_t)lane) * 16u);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rs,
                                                               lane_off + ((a * TN + b) * 4 + q) * 1024, tile_off + (int)((uint32_t)split * slice_stride), 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
            __syncthreads();                                   // (also: every wave is done with the operand tiles in LDS)
            unsigned* flag = reinterpret_cast<unsigned*>(smem);
            if (threadIdx.x == 0)
                *flag = __hip_atomic_fetch_add(g.xcnt + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag != (unsigned)(S - 1)) return;
            if (threadIdx.x == 0) __hip_atomic_store(g.xcnt + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fetch_epilogue();   // in flight together with the slab loads below
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
            for (int s0 = 0; s0 < S; ++s0) {   // added in slice order: bit-identical whoever arrives last
                f32x4 v[QN];
#pragma unroll
                for (int i = 0; i < QN; ++i)
                    v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off + i * 1024, tile_off + (int)((uint32_t)s0 * slice_stride), 16));
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 w = v[(a * TN + b) * 4 + q];
                            acc[a][b][4 * q] += w.x; acc[a][b][4 * q + 1] += w.y; acc[a][b][4 * q + 2] += w.z; acc[a][b][4 * q + 3] += w.w;
                        }
            }
        }
    }

    // epilogue. C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (!XS && g.splitk > 1 && !la) {
        // raw partials: slab[split][tile][BM][BN]
        float* slab = g.slabs + ((int64_t)split * g.total_tiles + bid) * (BM * BN);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int col = wn + b * 32 + li;
                    slab[row * BN + col] = acc[a][b][r];
                }
        return;
    }
    if constexpr (GATE) {
        static_assert(TM * TN == 1 && PREFETCH_C, "the fused gate epilogue is written for the 64x64 class");
        if (gidx >= 0) {
            // acc + cprev is the complete carried gradient of this (row, unit): run the gate backward of the next chain
            // step on it (same arithmetic as gru_step_bwd_kernel, gru.hip) and leave the direct path in the carry
            const twog_gru_step_bwd_t& S = ga->s[gidx];
            const int H = S.hidden;
            const int col = n0 + wn + li;
            const bool has_u = S.u != nullptr, has_hp = S.h_prev.ptr != nullptr;
            const int inner = S.dh.inner > 1 ? S.dh.inner : 1;
            const float inv_inner = 1.0f / (float)inner;
            float dul[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const bool ok = row < M && col < N;
                const float d = g_dh[r] + acc[0][0][r] + cprev[r];
                const float rg = g_rg[r], z = g_z[r], n = g_n[r], hn = g_hn[r];
                const float h0 = has_hp ? g_h0[r] : 0.f, uu = has_u ? g_uu[r] : 1.0f;
                const float gnew = (1.0f - z) * n + z * h0;
                dul[r] = ok ? d * (gnew - h0) : 0.f;
                const float dg = has_u ? uu * d : d;
                float dprev = has_u ? (1.0f - uu) * d : 0.f;
                const float dn = dg * (1.0f - z);
                const float dz = dg * (h0 - n);
                dprev += dg * z;
                const float dn_pre = dn * (1.0f - n * n);
                const float dr_pre = dn_pre * hn * rg * (1.0f - rg);
                const float dz_pre = dz * z * (1.0f - z);
                if (ok) {
                    const int o = (int)(((float)row + 0.5f) * inv_inner), i = row - o * inner;
                    float* dgi = S.dgi.ptr + (o * (int)S.dgi.ld_outer + i * (int)S.dgi.ld_inner);
                    float* dgh = S.dgh.ptr + (o * (int)S.dgh.ld_outer + i * (int)S.dgh.ld_inner);
                    dgi[col] = dr_pre;
                    dgi[H + col] = dz_pre;
                    dgi[2 * H + col] = dn_pre;
                    dgh[col] = dr_pre;
                    dgh[H + col] = dz_pre;
                    dgh[2 * H + col] = dn_pre * rg;
                    S.dh_prev.ptr[row * (int)S.dh_prev.ld_outer + col] = dprev;
                }
            }
            float* part = ga->du_part[gidx];
            if (part) {
                // row sums over this wave's 32 units (lanes of one half-wave share the rows); the 2 * tiles_n partials
                // per row are added in fixed order by du_reduce_kernel: deterministic, unlike atomics
                const int p = 2 * tn_idx + (wave & 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = dul[r];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (li == 0 && row < M) part[(int64_t)p * S.rows + row] = v;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row >= M) continue;
            float* crow = C.ptr + twog_row_off(C, row);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + wn + b * 32 + li;
                if (col >= N) continue;
                float v = acc[a][b][r];
                if constexpr (PREFETCH_C) {
                    v = (v + bv[b]) + cprev[r];   // (the order of splitk_reduce_kernel and of the 128x128 branch below)
                } else {  // 128x128 class: registers are full, and its long reduction makes the epilogue latency immaterial
                    if (bias) v += bias[col];
                    if (accumulate) v += crow[col];
                }
                if (act == 1) v = fmaxf(v, 0.f);
                crow[col] = v;
            }
        }
}

template <int BM, int BN, int NT, bool AKM, bool BKM, int D, bool KG>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const Group g) {
    gemm_tile<BM, BN, NT, AKM, BKM, D, KG, false>(g, nullptr);
}

template <bool AKM, bool BKM, bool KG>
__global__ __launch_bounds__(512, 4) void gemm_x3_kernel(const Group g) {   // 4 waves per SIMD = two workgroups per CU: <= 128 VGPRs
    gemm_tile<128, 128, 512, AKM, BKM, 2, KG, false, 1, false, true>(g, nullptr);
}
// forward (row-major) form with two k-tiles per barrier interval: one workgroup per CU (launches of at most one tile per CU)
__global__ __launch_bounds__(512, 2) void gemm_x3_nn_ku2_kernel(const Group g) {
    gemm_tile<128, 128, 512, false, false, 2, false, false, 1, false, true, 2>(g, nullptr);
}
// dW = dY^T X with every k-step's products through a fresh accumulator (TTMP): one workgroup per CU
// fragment reads one k-tile ahead of the MFMAs (PIPE): one workgroup per CU
template <bool AKM, bool BKM, bool KG>
__global__ __launch_bounds__(512, 1) void gemm_x3_pipe_kernel(const Group g) {
    gemm_tile<128, 128, 512, AKM, BKM, 2, KG, false, 1, false, true, 1, false, true>(g, nullptr);
}
// 16 waves = two k-groups on one tile (KS = 2): launches of at most one tile per CU (the segment level's per-step projection)
__global__ __launch_bounds__(1024, 1) void gemm_x3_nn_k2_kernel(const Group g) {
    gemm_tile<128, 128, 1024, false, false, 2, false, false, 2, false, true>(g, nullptr);
}
template <bool KG>
__global__ __launch_bounds__(512, 2) void gemm_x3_tt_split_acc_kernel(const Group g) {
    gemm_tile<128, 128, 512, true, true, 2, KG, false, 1, false, true, 1, true>(g, nullptr);
}

// Issue priority of the waves of a recurrent chain's launches (s_setprio, 0-3): beside the side stream's dW GEMMs a chain
// workgroup shares its CU with two GEMM workgroups, and the SIMD's arbiter serves equal-priority waves in turn.
#ifndef TWOG_CHAIN_PRIO
#define TWOG_CHAIN_PRIO 0
#endif
#if TWOG_CHAIN_PRIO > 0
#define TWOG_CHAIN_SETPRIO() __builtin_amdgcn_s_setprio(TWOG_CHAIN_PRIO)
#else
#define TWOG_CHAIN_SETPRIO() ((void)0)
#endif
// X3 on the 64x64 class (gemm_mainloop_x3s): 4 waves, or 8 waves = two k-groups; plain and gate-fused epilogues
template <bool BKM, int KS>
__global__ __launch_bounds__(256 * KS, 2) void gemm_x3s_kernel(const Group g) {
    TWOG_CHAIN_SETPRIO();
    gemm_tile<64, 64, 256 * KS, false, BKM, 2, false, false, KS, false, true>(g, nullptr);
}
template <int KS>
__global__ __launch_bounds__(256 * KS, 2) void gemm_gate_bwd_x3s_kernel(const Group g, const GateArgs ga) {
    TWOG_CHAIN_SETPRIO();
    gemm_tile<64, 64, 256 * KS, false, true, 2, false, true, KS, false, true>(g, &ga);
}

// 128 x 64 tiles for chain launches with enough rows (pick_rows128): one 8-wave workgroup per CU, two k-tiles per barrier
template <bool BKM>
__global__ __launch_bounds__(512, 1) void gemm_x3su128_kernel(const Group g) {
    TWOG_CHAIN_SETPRIO();
    gemm_tile<128, 64, 512, false, BKM, 2, false, false, 1, false, true, 2>(g, nullptr);
}

// KU k-tiles per barrier interval (launches that run one workgroup per CU: at most 256 tiles)
template <bool BKM, int KS, int KU>
__global__ __launch_bounds__(256 * KS, KS == 2 ? 1 : 2) void gemm_x3su_kernel(const Group g) {
    TWOG_CHAIN_SETPRIO();
    gemm_tile<64, 64, 256 * KS, false, BKM, 2, false, false, KS, false, true, KU>(g, nullptr);
}
template <int KS, int KU>
__global__ __launch_bounds__(256 * KS, KS == 2 ? 1 : 2) void gemm_gate_bwd_x3su_kernel(const Group g, const GateArgs ga) {
    TWOG_CHAIN_SETPRIO();
    gemm_tile<64, 64, 256 * KS, false, true, 2, false, true, KS, false, true, KU>(g, &ga);
}

// 64x64 class, A row-major, B k-major (dX = dY W): the only form the recurrent backward chains use
template <int D>
__global__ __launch_bounds__(256, 2) void gemm_gate_bwd_kernel(const Group g, const GateArgs ga) {
    gemm_tile<64, 64, 256, false, true, D, false, true>(g, &ga);
}

// 64x64 tiles, 8 waves, k-split inside the workgroup (see gemm_mainloop): launches with fewer tiles than workgroup slots
template <bool BKM, int D>
__global__ __launch_bounds__(512, 2) void gemm_ks_kernel(const Group g) {
    gemm_tile<64, 64, 512, false, BKM, D, false, false, 2>(g, nullptr);
}
template <int D>
__global__ __launch_bounds__(512, 2) void gemm_gate_bwd_ks_kernel(const Group g, const GateArgs ga) {
    gemm_tile<64, 64, 512, false, true, D, false, true, 2>(g, &ga);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused GRU forward step of the recurrent chains (gru.hip, segrnn.hip): one launch per chain step computes
//   [r | z | n_h] = h_prev W_hh^T            (and  [r | z | n_i] += m W_ih[:, msg]^T  at the segment level)
// on 64-row x (64 units x 3 gates) tiles and finishes the step in the epilogue -- sigmoid / tanh, the blend with h_prev,
// the segment gate u -- from the accumulators: no gh round trip through memory, no gate launch. Same arithmetic as
// gru_step_fwd_kernel (gru.hip); r, z, n and W_hn h + b_hn are saved for the backward chain as before.
// A workgroup stages 64 + 192 operand rows per k-tile (24 FLOP/B instead of the 64x64 tile's 16 -- these launches are
// bound by the L2 port of their CU, profiles/HISTORY.md section 8) and the unit tile index is the block's XCD, so each L2 keeps
// one 192-row slice of every weight.
struct GruFwdProb {
    twog_rows_t A, B, A2, B2;      // previous states x W_hh; aggregated messages x W_ih[:, msg] (K2 == 0: absent)
    twog_rows_t gi, h_out, save;  // W_ih x + b_ih of this step [rows][3h]; new state; saved gates [rows][4h]
    const float* b_hh;            // [3h] or nullptr
    const float* u;               // segment gate per row or nullptr
    int u_ld_outer, u_ld_inner;
    int K2, rows, has_prev, rt_start, inner;
};
struct GruFwdGroup {
    GruFwdProb p[MAXP];
    int n, tiles_n, hidden;
};

template <int D, int KS, bool X3 = false>
__global__ __launch_bounds__(256 * KS, 1) void gemm_gru_fwd_kernel(const GruFwdGroup g) {
    TWOG_CHAIN_SETPRIO();
    constexpr int BM = 64, BN = 192, NT = 256 * KS, TM = 1, TN = 3;
    // X3 (bf16 x 3 on the bf16 matrix cores, see gemm_mainloop_x3s): two stages of three bf16 planes of both operand tiles
    __shared__ __attribute__((aligned(16))) float smem[X3 ? 2 * 3 * (BM + BN) * (16 * KS) * 2 / 4 : 2 * (BM + BN) * (BK + 4)];
    const int ut = blockIdx.x % g.tiles_n, rt = blockIdx.x / g.tiles_n;
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (rt >= g.p[i].rt_start) pi = i;
    const GruFwdProb& P = g.p[pi];
    const int H = g.hidden, M = P.rows, m0 = (rt - P.rt_start) * BM, n0 = ut * 64;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, kgrp = threadIdx.x >> 8;   // KS == 2: two k-groups
    const int li = lane & 31, kh = lane >> 5;
    const int wm = (wave >> 1) * 32;
    const int unit = n0 + (wave & 1) * 32 + li, unitc = min(unit, H - 1);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;

    // epilogue operands: requested next to the first operand tiles, they arrive under the reduction loop
    const bool has_u = P.u != nullptr, has_hp = P.has_prev != 0;
    const int inner = P.inner;
    const float inv_inner = 1.0f / (float)inner;
    const float b_r = P.b_hh ? P.b_hh[unitc] : 0.f, b_z = P.b_hh ? P.b_hh[H + unitc] : 0.f, b_n = P.b_hh ? P.b_hh[2 * H + unitc] : 0.f;
    float e_r[16], e_z[16], e_n[16], e_h0[16], e_u[16];
    if (kgrp == 0) {
        const float* hp_ptr = has_hp ? P.A.ptr : P.gi.ptr;
        const int hp_lo = has_hp ? (int)P.A.ld_outer : (int)P.gi.ld_outer, hp_li = has_hp ? (int)P.A.ld_inner : (int)P.gi.ld_inner;
        const float* u_ptr = has_u ? P.u : P.gi.ptr;
        const int u_lo = has_u ? P.u_ld_outer : 0, u_li = has_u ? P.u_ld_inner : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowc = min(m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh, M - 1);
            const int o = (int)(((float)rowc + 0.5f) * inv_inner), i = rowc - o * inner;
            const float* gi = P.gi.ptr + (o * (int)P.gi.ld_outer + i * (int)P.gi.ld_inner);
            e_r[r] = gi[unitc];
            e_z[r] = gi[H + unitc];
            e_n[r] = gi[2 * H + unitc];
            e_h0[r] = hp_ptr[o * hp_lo + i * hp_li + unitc];
            e_u[r] = u_ptr[o * u_lo + i * u_li];
        }
    }

    if constexpr (X3) gemm_mainloop_x3s<BM, BN, NT, false, KS, TN, true, 2, false>(P.A, P.B, M, H, m0, n0, 0, H, smem, acc);
    else gemm_mainloop<BM, BN, NT, false, false, true, TM, TN, D, false, KS, true>(P.A, P.B, M, H, 1, 1, m0, n0, 0, H, smem, acc);
    f32x16 hn = acc[0][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][2][r] = 0.f;
    if (P.K2 > 0) {   // uniform per workgroup: the message columns of W_ih; their n block stays on the input side of the gate
        __syncthreads();   // every wave is done with the operand tiles of the first product
        if constexpr (X3) gemm_mainloop_x3s<BM, BN, NT, false, KS, TN, true, 2, false>(P.A2, P.B2, M, H, m0, n0, 0, P.K2, smem, acc);
        else gemm_mainloop<BM, BN, NT, false, false, true, TM, TN, D, false, KS, true>(P.A2, P.B2, M, H, 1, 1, m0, n0, 0, P.K2, smem, acc);
    }
    if constexpr (KS == 2) {
        // add the two k-groups' partial tiles (fixed order: group 0 + group 1) through LDS; group 0 runs the epilogue
        __syncthreads();
        float* red = smem + ((wave * 64) << 6) + lane;   // 4 accumulators x 16 registers per lane, lane-contiguous
        if (kgrp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                red[r << 6] = acc[0][0][r];
                red[(16 + r) << 6] = acc[0][1][r];
                red[(32 + r) << 6] = acc[0][2][r];
                red[(48 + r) << 6] = hn[r];
            }
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][0][r] += red[r << 6];
            acc[0][1][r] += red[(16 + r) << 6];
            acc[0][2][r] += red[(32 + r) << 6];
            hn[r] += red[(48 + r) << 6];
        }
    }
    if (unit >= H) return;
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row >= M) continue;
        const float ir = e_r[r] + acc[0][0][r], hr = b_r;
        const float iz = e_z[r] + acc[0][1][r], hz = b_z;
        const float in_ = e_n[r] + acc[0][2][r], hnv = hn[r] + b_n;
        const float rg = 1.0f / (1.0f + expf(-(ir + hr)));
        const float z = 1.0f / (1.0f + expf(-(iz + hz)));
        const float n = tanhf(in_ + rg * hnv);
        const float h0 = has_hp ? e_h0[r] : 0.f;
        const float gnew = (1.0f - z) * n + z * h0;
        const float uu = e_u[r];
        const int o = (int)(((float)row + 0.5f) * inv_inner), i = row - o * inner;
        P.h_out.ptr[o * (int)P.h_out.ld_outer + i * (int)P.h_out.ld_inner + unit] = has_u ? uu * gnew + (1.0f - uu) * h0 : gnew;
        float* sv = P.save.ptr + (o * (int)P.save.ld_outer + i * (int)P.save.ld_inner);
        sv[unit] = rg;
        sv[H + unit] = z;
        sv[2 * H + unit] = n;
        sv[3 * H + unit] = hnv;
    }
}

// the same four kernels with the reduction also split over workgroups and combined in the launch (XS, see gemm_tile)
template <bool BKM, int D>
__global__ __launch_bounds__(512, 2) void gemm_xs_kernel(const Group g) {
    gemm_tile<64, 64, 512, false, BKM, D, false, false, 2, true>(g, nullptr);
}
template <int D>
__global__ __launch_bounds__(512, 2) void gemm_gate_bwd_xs_kernel(const Group g, const GateArgs ga) {
    gemm_tile<64, 64, 512, false, true, D, false, true, 2, true>(g, &ga);
}
template <bool BKM, int D>
__global__ __launch_bounds__(256, 2) void gemm_xs32_kernel(const Group g) {
    gemm_tile<32, 64, 256, false, BKM, D, false, false, 2, true>(g, nullptr);
}
template <int D>
__global__ __launch_bounds__(256, 2) void gemm_gate_bwd_xs32_kernel(const Group g, const GateArgs ga) {
    gemm_tile<32, 64, 256, false, true, D, false, true, 2, true>(g, &ga);
}

// 32 x 64 tiles, 4 waves = two k-groups of a 1 x 2 wave grid: launches with so few rows that even 32-row tiles leave one
// tile per CU (8 clips per GPU: 16 / 32 rows per entity type and direction). A 64 x 64 tile there is half padding and
// keeps every SIMD busy for the whole K/2-instruction MFMA chain twice; here each wave runs half the chain once.
template <bool BKM, int D>
__global__ __launch_bounds__(256, 2) void gemm_ks32_kernel(const Group g) {
    gemm_tile<32, 64, 256, false, BKM, D, false, false, 2>(g, nullptr);
}
template <int D>
__global__ __launch_bounds__(256, 2) void gemm_gate_bwd_ks32_kernel(const Group g, const GateArgs ga) {
    gemm_tile<32, 64, 256, false, true, D, false, true, 2>(g, &ga);
}

// sums split-K slabs in fixed order and applies the epilogue
template <int BM, int BN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const Group g) {
    int bid = blockIdx.x;
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (bid >= g.p[i].tile_start) pi = i;
    const Prob& P = g.p[pi];
    int tile = bid - P.tile_start;
    const int per_batch = P.tiles_m * P.tiles_n, bi = tile / per_batch;
    tile -= bi * per_batch;
    twog_rows_t C = P.C;
    C.ptr += bi * P.c_bs;
    int tm_idx, tn_idx;
    tile_coords(P, tile, g.group, tm_idx, tn_idx);
    const int m0 = tm_idx * BM, n0 = tn_idx * BN;
    if (P.cs && tn_idx == 0 && blockIdx.y == gridDim.y - 1 && threadIdx.x < 128 && m0 + (int)threadIdx.x < P.M) {
        float v = 0.f;   // the slices' column sums of A, in slice order
        for (int s = 0; s < g.splitk; ++s) v += g.cs_part[((int64_t)s * g.total_tiles + bid) * 128 + threadIdx.x];
        float* o = P.cs + m0 + threadIdx.x;
        *o = P.cs_acc ? *o + v : v;
    }
    // grid.y slices the tile so that small-output / deep-split problems still spread over the chip
    const int chunk = (BM * BN) / gridDim.y;
    // 16-byte path: four columns per lane when the output rows allow it (whole column quads inside N, aligned rows)
    const bool vec = (P.N % 4 == 0) && (reinterpret_cast<uintptr_t>(C.ptr) % 16 == 0) && (C.ld_outer % 4 == 0) &&
                     (C.inner <= 1 || C.ld_inner % 4 == 0) && (!P.bias || reinterpret_cast<uintptr_t>(P.bias) % 16 == 0);
    if (vec) {
        for (int e = blockIdx.y * chunk + threadIdx.x * 4; e < (blockIdx.y + 1) * chunk; e += 1024) {
            const int row = e / BN, col = e - row * BN;
            if (m0 + row >= P.M || n0 + col >= P.N) continue;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < g.splitk; ++s)
                v += *reinterpret_cast<const f32x4*>(g.slabs + ((int64_t)s * g.total_tiles + bid) * (BM * BN) + e);
            float* c = C.ptr + twog_row_off(C, m0 + row) + n0 + col;
            if (P.bias) v += *reinterpret_cast<const f32x4*>(P.bias + n0 + col);
            if (P.accumulate) v += *reinterpret_cast<const f32x4*>(c);
            if (P.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<f32x4*>(c) = v;
        }
        return;
    }
    for (int e = blockIdx.y * chunk + threadIdx.x; e < (blockIdx.y + 1) * chunk; e += 256) {
        const int row = e / BN, col = e - row * BN;
        if (m0 + row >= P.M || n0 + col >= P.N) continue;
        float v = 0.f;
        for (int s = 0; s < g.splitk; ++s) v += g.slabs[((int64_t)s * g.total_tiles + bid) * (BM * BN) + e];
        float* c = C.ptr + twog_row_off(C, m0 + row) + n0 + col;
        if (P.bias) v += P.bias[n0 + col];
        if (P.accumulate) v += *c;
        if (P.act == 1) v = fmaxf(v, 0.f);
        *c = v;
    }
}

inline int vec_ok(const twog_rows_t& m, int64_t batch_stride, int contiguous_extent, int64_t n_rows) {
    // 32-bit element offsets inside the kernel: the operand must span fewer than 2^30 elements
    const int64_t groups = m.inner > 1 ? (n_rows + m.inner - 1) / m.inner : n_rows;
    const int64_t extent = groups * (m.ld_outer < 0 ? -m.ld_outer : m.ld_outer) + (m.inner > 1 ? (int64_t)m.inner * m.ld_inner : 0);
    if (extent >= (int64_t(1) << 30)) return 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(m.ptr) % 16) == 0;
    const bool ld_ok = (m.ld_outer % 4 == 0) && (m.inner <= 1 || m.ld_inner % 4 == 0) && (batch_stride % 4 == 0);
    return (aligned && ld_ok && contiguous_extent >= 4 && contiguous_extent % 4 == 0) ? 1 : 0;
}

thread_local int g_last_class_x3 = 0;

// the bf16x3 128x128 class serves this group: aligned operands, whole 16-deep k-tiles (TWOG_GEMM_X3=0: native fp32 MFMA)
static bool x3_128_ok(const Group& g) {
    static const int x3_on = getenv("TWOG_GEMM_X3") ? atoi(getenv("TWOG_GEMM_X3")) : 1;
    bool ok = x3_on != 0 && (g.k_per_split % X3_BK) == 0;
    for (int i = 0; i < g.n; ++i) ok = ok && g.p[i].a_vec && g.p[i].b_vec && (g.p[i].K % X3_BK) == 0 && g.p[i].M >= 4 && g.p[i].N >= 4;
    return ok;
}

thread_local int g_dw_one_per_cu = getenv("TWOG_DW_ONE_PER_CU") ? atoi(getenv("TWOG_DW_ONE_PER_CU")) : 0;

template <int BM, int BN, int NT, int D>
int launch(Group& g, int akm, int bkm, hipStream_t st) {
    g_last_class_x3 = 0;
    dim3 grid(g.total_tiles, g.splitk), block(NT);
    // KG variant: some k-major operand has (outer, inner) grouped rows
    bool kg = false;
    for (int i = 0; i < g.n; ++i) kg = kg || (akm && g.p[i].A.inner > 1) || (bkm && g.p[i].B.inner > 1);
    if constexpr (BM == 128 && NT == 512) {
        // X3 (fp32-exact operands on the bf16 matrix cores, gemm_mainloop_x3): aligned operands, whole k-tiles, plain rows
        if (x3_128_ok(g)) {
            static const int split_acc = getenv("TWOG_X3_DW_SPLIT_ACC") ? atoi(getenv("TWOG_X3_DW_SPLIT_ACC")) : 0;
            // at most one tile per CU, whole pairs of k-tiles, no split-K: two k-tiles per barrier interval. Built and measured
            // in round 6 (VERDICT r05 item 7's "raise the 128x128 class"), SLOWER: the segment level's 240-tile projection launch
            // 47.4 -> 49.8 us, the 64-clip step 65.59 -> 65.84 ms (same box, alternating twice): profiles/r06_gemm128_ku2_ab.txt.
            // Off by default (TWOG_X3_KU128=1 selects it).
            static const int ku128 = getenv("TWOG_X3_KU128") ? atoi(getenv("TWOG_X3_KU128")) : 0;
            bool ku2 = ku128 != 0 && !akm && !bkm && g.splitk == 1 && g.total_tiles <= 256;
            for (int i = 0; i < g.n; ++i) ku2 = ku2 && (g.p[i].K % (2 * X3_BK)) == 0;
            // at most one tile per CU, whole pairs of k-tiles: 16 waves, the reduction halved between two k-groups. Built and measured
            // in round 6 (four waves per SIMD on a lone tile, as two co-resident workgroups of a big launch have): NO gain -- the
            // 240-tile projection launch 47.0 -> 46.3 us, the step unchanged. The sixteen waves meet at ONE barrier per k-tile, so their
            // LDS phases (768 cycles of plane stores + fragment reads per k-tile on the CU) and their MFMA phases (768 cycles per
            // SIMD) still alternate instead of overlapping, which is what two independent workgroups get for free.
            // Off by default (TWOG_X3_K2=1 selects it). profiles/r06_gemm128_two_k_groups.txt
            static const int k2_on = getenv("TWOG_X3_K2") ? atoi(getenv("TWOG_X3_K2")) : 0;
            bool k2 = k2_on != 0 && !ku2 && !akm && !bkm && g.splitk == 1 && g.total_tiles <= 256;
            for (int i = 0; i < g.n; ++i) k2 = k2 && (g.p[i].K % (2 * X3_BK)) == 0 && g.p[i].K >= 8 * X3_BK;
            // fragment reads one k-tile ahead (PIPE, one workgroup per CU): TWOG_X3_PIPE bit 0 = the forward-form launches of at
            // most one tile per CU, bit 1 = every forward-form launch, bit 2 = the dX / dW forms too. Built and measured in round 6,
            // bit-identical and SLOWER: the 240-tile projection launch +3 us, the big launches with one pipelined workgroup per CU
            // lose to two unpipelined ones (roofline.frac 0.430 -> 0.398 forward forms, 0.347 all forms). Off by default.
            // profiles/r06_gemm128_fragment_reads_ahead.txt
            static const int pipe_on = getenv("TWOG_X3_PIPE") ? atoi(getenv("TWOG_X3_PIPE")) : 0;
            const bool one_per_cu = g.splitk == 1 && g.total_tiles <= 256;
            const bool pipe_nn = !akm && !bkm && (((pipe_on & 1) && one_per_cu) || (pipe_on & 2));
            if (pipe_nn) hipLaunchKernelGGL((gemm_x3_pipe_kernel<false, false, false>), grid, block, 0, st, g);
            else if ((pipe_on & 4) && !akm && bkm && !kg) hipLaunchKernelGGL((gemm_x3_pipe_kernel<false, true, false>), grid, block, 0, st, g);
            else if ((pipe_on & 4) && akm && bkm && !kg && !split_acc) hipLaunchKernelGGL((gemm_x3_pipe_kernel<true, true, false>), grid, block, 0, st, g);
            else if (k2) hipLaunchKernelGGL(gemm_x3_nn_k2_kernel, grid, dim3(1024), 0, st, g);
            else if (ku2) hipLaunchKernelGGL(gemm_x3_nn_ku2_kernel, grid, block, 0, st, g);
            else if (!akm && !bkm) hipLaunchKernelGGL((gemm_x3_kernel<false, false, false>), grid, block, 0, st, g);
            else if (!akm && bkm && !kg) hipLaunchKernelGGL((gemm_x3_kernel<false, true, false>), grid, block, 0, st, g);
            else if (!akm && bkm) hipLaunchKernelGGL((gemm_x3_kernel<false, true, true>), grid, block, 0, st, g);
            else if (akm && bkm && split_acc && !kg) hipLaunchKernelGGL((gemm_x3_tt_split_acc_kernel<false>), grid, block, 0, st, g);
            else if (akm && bkm && split_acc) hipLaunchKernelGGL((gemm_x3_tt_split_acc_kernel<true>), grid, block, 0, st, g);
            else if (akm && bkm && g_dw_one_per_cu) {
                // (experiment, round 6: the dW launches of a side stream with ONE workgroup per CU -- 40 KB of unused dynamic LDS
                // on top of the 48 KB of stages -- so that half of every CU's registers stay free for a 4-wave chain workgroup;
                // twog_gemm_dw_one_per_cu(1) / TWOG_DW_ONE_PER_CU=1; profiles/r06_dw_one_workgroup_per_cu.txt)
                static std::atomic<uint32_t> a0{0}, a1{0};
                if (!kg) { twog_allow_dynamic_lds(gemm_x3_kernel<true, true, false>, 40 * 1024, a0); hipLaunchKernelGGL((gemm_x3_kernel<true, true, false>), grid, block, 40 * 1024, st, g); }
                else { twog_allow_dynamic_lds(gemm_x3_kernel<true, true, true>, 40 * 1024, a1); hipLaunchKernelGGL((gemm_x3_kernel<true, true, true>), grid, block, 40 * 1024, st, g); }
            }
            else if (akm && bkm && !kg) hipLaunchKernelGGL((gemm_x3_kernel<true, true, false>), grid, block, 0, st, g);
            else if (akm && bkm) hipLaunchKernelGGL((gemm_x3_kernel<true, true, true>), grid, block, 0, st, g);
            else if (!kg) hipLaunchKernelGGL((gemm_x3_kernel<true, false, false>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_x3_kernel<true, false, true>), grid, block, 0, st, g);
            TWOG_CHECK_LAUNCH();
            g_last_class_x3 = 1;
            if (g.splitk > 1 && !g.xcnt) {
                hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN>), dim3(g.total_tiles, (BM * BN) / 1024), dim3(256), 0, st, g);
                TWOG_CHECK_LAUNCH();
            }
            return 0;
        }
        // not served by X3: grouped k-major rows keep the 4-wave fp32 tiles (their pointer-carrying 8-wave kernels exceed
        // 128 VGPRs and measure 7 % slower)
        if (kg) return launch<BM, BN, 256, D>(g, akm, bkm, st);
    }
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, NT, false, false, D, false>), grid, block, 0, st, g);
    else if (!akm && bkm && !kg) hipLaunchKernelGGL((gemm_kernel<BM, BN, NT, false, true, D, false>), grid, block, 0, st, g);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, NT, false, true, D, true>), grid, block, 0, st, g);
    else if (akm && bkm && !kg) hipLaunchKernelGGL((gemm_kernel<BM, BN, NT, true, true, D, false>), grid, block, 0, st, g);
    else if (akm && bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, NT, true, true, D, true>), grid, block, 0, st, g);
    else if (!kg) hipLaunchKernelGGL((gemm_kernel<BM, BN, NT, true, false, D, false>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, NT, true, false, D, true>), grid, block, 0, st, g);
    TWOG_CHECK_LAUNCH();
    if (g.splitk > 1 && !g.xcnt) {
        hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN>), dim3(g.total_tiles, (BM * BN) / 1024), dim3(256), 0, st, g);
        TWOG_CHECK_LAUNCH();
    }
    return 0;
}

thread_local int g_last_class = 0;  // twog_gemm_last_class(): variant of this thread's most recent launch

}  // namespace

extern "C" int twog_gemm_last_class(void) { return g_last_class; }

// Builds the launch descriptor of one chunk (<= MAXP problems): tile class, class-sorted problem list (order[i] = index of
// the caller's problem that became sorted problem i), XCD map, split-K. Shared by the plain and the gate-fused launch.
constexpr size_t SPLITK_TICKET_BYTES = 16384;   // 4096 tickets at the start of the split-K workspace (see prepare_group)

static void prepare_group(const twog_gemm_t* pr, int n, int a_kmajor, int b_kmajor, void* workspace,
                          size_t workspace_bytes, Group& g, int* order, bool& big, int& bm) {
    // tile choice: 128x128 tiles (4 MFMA tiles per wave, half the LDS traffic per FLOP) whenever the problems are
    // at least one tile wide and -- possibly with split-K -- still fill the chip; 64x64 for the skinny ones.
    int64_t tiles128 = 0;
    int kmax = 0;
    bool wide = true;
    for (int i = 0; i < n; ++i) {
        const int nb = pr[i].batch > 0 ? pr[i].batch : 1;
        tiles128 += (int64_t)nb * ((pr[i].M + 127) / 128) * ((pr[i].N + 127) / 128);
        if (pr[i].K > kmax) kmax = pr[i].K;
        if (pr[i].M < 96 || pr[i].N < 96) wide = false;
    }
    static const int force_tile = getenv("TWOG_GEMM_TILE") ? atoi(getenv("TWOG_GEMM_TILE")) : 0;
    static const int force_split = getenv("TWOG_GEMM_SPLITK") ? atoi(getenv("TWOG_GEMM_SPLITK")) : 0;
    const int64_t reach128 = tiles128 * (workspace ? (kmax >= 1024 ? kmax / 512 : 1) : 1);
    // (200 since the class multiplies on the bf16 pipes: its tiles run at twice the 64x64 class's rate, so a launch that
    // leaves a fifth of the CUs without a 128-tile still wins -- bs64 step 67.10 -> 66.71 ms; 224: 66.78, 176: 67.15)
    static const int big_min = getenv("TWOG_GEMM_BIG_MIN") ? atoi(getenv("TWOG_GEMM_BIG_MIN")) : 200;
    big = wide && (tiles128 >= big_min || reach128 >= 256);
    if (force_tile == 128) big = true;
    if (force_tile == 64) big = false;
    const int BMN = big ? 128 : 64;
    // 32-row tiles (gemm_ks32_kernel) for the in-library chain launches (no workspace) of small batches: one tile per CU
    // even at 32 rows, a reduction worth splitting between two waves, no grouped k-major rows
    bm = BMN;
    {
        static const int bm32_on = getenv("TWOG_GEMM_BM32") ? atoi(getenv("TWOG_GEMM_BM32")) : 1;
        static const int ks_allowed = getenv("TWOG_GEMM_KS") ? atoi(getenv("TWOG_GEMM_KS")) : 1;
        int64_t tiles32 = 0;
        bool plain = true;
        for (int i = 0; i < n; ++i) {
            tiles32 += (int64_t)((pr[i].M + 31) / 32) * ((pr[i].N + 63) / 64);
            plain = plain && pr[i].batch <= 1 && !(b_kmajor && pr[i].B.inner > 1);
        }
        if (bm32_on && ks_allowed && !big && !a_kmajor && !workspace && plain && kmax >= 256 && tiles32 <= 256) bm = 32;
    }
    g.n = n;
    // longest reductions first, equal K adjacent (stable: the caller's order within a class is kept)
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && pr[order[j]].K > pr[order[j - 1]].K; --j) { const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp; }
    int t = 0;
    g.n_cls = 0;
    for (int i = 0; i < n; ++i) {
        Prob& P = g.p[i];
        const twog_gemm_t& q = pr[order[i]];
        P.A = q.A; P.B = q.B; P.C = q.C; P.bias = q.bias;
        P.M = q.M; P.N = q.N; P.K = q.K;
        P.act = q.act; P.accumulate = q.accumulate;
        P.cs = q.a_colsum; P.cs_acc = q.a_colsum_accumulate;
        P.tiles_m = (P.M + bm - 1) / bm;
        P.tiles_n = (P.N + BMN - 1) / BMN;
        P.tile_start = t;
        P.batch = q.batch > 0 ? q.batch : 1;
        static const int force_nmajor = getenv("TWOG_GEMM_NMAJOR") ? atoi(getenv("TWOG_GEMM_NMAJOR")) : -1;
        P.n_major = force_nmajor >= 0 ? force_nmajor : (P.N > P.M ? 1 : 0);
        P.a_bs = q.a_batch_stride; P.b_bs = q.b_batch_stride; P.c_bs = q.c_batch_stride;
        const int ntiles = P.batch * P.tiles_m * P.tiles_n;
        if (i == 0 || P.K != g.p[i - 1].K) {
            g.cls_start[g.n_cls] = t;
            g.cls_ntiles[g.n_cls] = 0;
            ++g.n_cls;
        }
        g.cls_ntiles[g.n_cls - 1] += ntiles;
        t += ntiles;
        P.a_vec = vec_ok(P.A, q.a_batch_stride, a_kmajor ? P.M : P.K, a_kmajor ? P.K : P.M);
        P.b_vec = vec_ok(P.B, q.b_batch_stride, b_kmajor ? P.N : P.K, b_kmajor ? P.K : P.N);
    }
    for (int c = 0, rot = 0; c < g.n_cls; ++c) {
        g.cls_rot[c] = rot & 7;
        rot += g.cls_ntiles[c] & 7;
    }
    g.total_tiles = t;
    // grouped tile order for the 128x128 class (TWOG_GEMM_GROUP=0: plain order; the 64x64 chain launches keep it)
    static const int group = getenv("TWOG_GEMM_GROUP") ? atoi(getenv("TWOG_GEMM_GROUP")) : 8;
    static const int group64 = getenv("TWOG_GEMM_GROUP64") ? atoi(getenv("TWOG_GEMM_GROUP64")) : 0;
    g.group = big ? group : group64;
    g.splitk = 1;
    g.k_per_split = ((kmax + BK - 1) / BK) * BK;
    g.slabs = nullptr;
    g.cs_part = nullptr;
    g.xcnt = nullptr;
    g.xs_early = 0;
    g.xcd_split = 0;
    // deterministic split-K when the grid would leave CUs idle and the reduction is long
    if (t < 4096 && kmax >= 1024 && workspace) {
        // pick the split that fills whole "rounds" of resident workgroups (2 per CU for 128-tiles, 4 for 64-tiles)
        const int slots = big ? 512 : 1024;
        const int max_by_k = kmax / 256;
        int want = 1;
        double best = 0.0;
        const int max_split = t < 384 ? 64 : 16;
        for (int sft = 1; sft <= max_split && sft <= max_by_k; ++sft) {
            const int64_t wg = (int64_t)t * sft;
            if (wg > (t < 384 ? 2 : 8) * slots && sft > 1) break;
            const double eff = (double)wg / (double)(((wg + slots - 1) / slots) * slots);
            if (eff > best + 0.03) { best = eff; want = sft; }
        }
        if (force_split > 0) want = force_split;
        // 128x128 class with few tiles (the tall dW reductions): a multiple of 8 splits, whole splits dealt to XCDs
        // (g.xcd_split, see gemm_tile). An XCD holds 64 such workgroups (32 CUs x 2): take the smallest multiple of 8 whose
        // t * s / 8 workgroups per XCD fill whole rounds of 64 to 95 % (48 tiles -> 32 splits = 3 rounds; 32 -> 16; 64 -> 8),
        // else the best one -- unless the plain choice quantises clearly better. TWOG_GEMM_XCD_SPLIT=0: tile chunks.
        static const int xcd_on = getenv("TWOG_GEMM_XCD_SPLIT") ? atoi(getenv("TWOG_GEMM_XCD_SPLIT")) : 1;
        int want8 = 0;
        if (xcd_on && big && want > 1 && force_split <= 0 && t <= 512) {
            double best8 = 0.0;
            // (splits of at least 1 024 k: below that the slabs -- 64 KB written and read per tile and split -- weigh more
            // than the operand re-reads saved; measured on the K = 15 360 shapes: 0.365 -> 0.426 ms with 24 splits of 640)
            for (int s8 = 8; s8 <= 64 && s8 <= kmax / 1024; s8 += 8) {
                if ((size_t)s8 * t * BMN * BMN * sizeof(float) + SPLITK_TICKET_BYTES > workspace_bytes) break;
                const int64_t per_xcd = (int64_t)t * s8 / 8;
                const double e8 = (double)per_xcd / (double)(((per_xcd + 63) / 64) * 64);
                if (e8 > best8 + 1e-9) { best8 = e8; want8 = s8; }
                if (e8 >= 0.95) break;
            }
            if (want8 && best8 + 0.08 < best) want8 = 0;   // (48 tiles x 8 splits of 1 920 at K = 15 360: 0.150 against 0.138 ms)
        }
        if (want8) want = want8;
        // (experiment, round 6: SHORT workgroups for the tall dW reductions -- TWOG_GEMM_SLAB_K=k asks for splits of about k
        // reduction rows, a multiple of 8, as many as the workspace holds -- so that a launch chain on another stream finds
        // free compute units sooner; profiles/r06_dw_short_slabs_beside_the_chain.txt. Off by default.)
        static const int slab_k = getenv("TWOG_GEMM_SLAB_K") ? atoi(getenv("TWOG_GEMM_SLAB_K")) : 0;
        if (slab_k > 0 && big && a_kmajor && b_kmajor && force_split <= 0 && t <= 512 && kmax >= 4 * slab_k) {
            int s8 = ((kmax / slab_k + 7) / 8) * 8;
            while (s8 > want && (size_t)s8 * t * BMN * BMN * sizeof(float) + SPLITK_TICKET_BYTES + (size_t)(s8 + 1) * t * 128 * sizeof(float) > workspace_bytes) s8 -= 8;
            if (s8 > want) want = s8;
        }
        // the first 16 KB of the workspace are the arrival tickets of the in-launch combine (LA, gemm_tile): zero when the
        // workspace is first handed over, returned to zero by every launch. OFF by default -- measured on one box, same
        // session: bs64 step 69.32 ms with it against 68.57 ms with slabs + splitk_reduce_kernel, 8-clip step 16.32 against
        // 16.28 (profiles/r05_splitk_in_launch_combine_ab.txt): the one workgroup per tile that arrives last re-reads S slabs
        // of 64 KB alone, where the reduce launch spreads the same bytes over 16 workgroups per tile at 5 TB/s; what the
        // launch boundary costs is less than that. TWOG_GEMM_LA=1 selects it (tests run both).
        static const int la_on = getenv("TWOG_GEMM_LA") ? atoi(getenv("TWOG_GEMM_LA")) : 0;
        bool any_cs = false;
        for (int i = 0; i < n; ++i) any_cs = any_cs || pr[i].a_colsum != nullptr;
        const size_t cs_bytes = any_cs ? (size_t)(want + 1) * t * 128 * sizeof(float) : 0;   // (+1: the split count is rounded below)
        const size_t need = (size_t)want * t * BMN * BMN * sizeof(float) + SPLITK_TICKET_BYTES + cs_bytes;
        if (want > 1 && need <= workspace_bytes) {
            int kps = (kmax + want - 1) / want;
            g.k_per_split = ((kps + BK - 1) / BK) * BK;
            g.splitk = (kmax + g.k_per_split - 1) / g.k_per_split;
            g.slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + SPLITK_TICKET_BYTES);
            g.xcd_split = (want8 && g.splitk % 8 == 0) ? 1 : 0;
            const bool fits32 = (uint64_t)g.splitk * t * BMN * BMN * sizeof(float) < (uint64_t(1) << 32);
            if (la_on && !any_cs && bm == BMN && t <= (int)(SPLITK_TICKET_BYTES / sizeof(unsigned)) && fits32) g.xcnt = static_cast<unsigned*>(workspace);
            if (any_cs) g.cs_part = g.slabs + (size_t)g.splitk * t * BMN * BMN;
        }
    }
}

// X3 on the 64-row tile class pays where the launch is bound by its MFMAs: a real batch (many tiles), reductions of at least
// 256. TWOG_GEMM_X3=0 (everything native fp32) and TWOG_GEMM_X3S=0 (only this class) turn it off.
static bool x3s_ok(const Group& g, int a_kmajor, int xk, int min_tiles = 96) {
    static const int on = (getenv("TWOG_GEMM_X3") ? atoi(getenv("TWOG_GEMM_X3")) : 1) && (getenv("TWOG_GEMM_X3S") ? atoi(getenv("TWOG_GEMM_X3S")) : 1);
    if (!on || a_kmajor || g.splitk != 1 || g.total_tiles < min_tiles) return false;
    int kmax = 0;
    for (int i = 0; i < g.n; ++i) {
        const Prob& P = g.p[i];
        if (!P.a_vec || !P.b_vec || P.K % xk || P.batch > 1 || P.B.inner > 1 || P.M < 1 || P.N < 4) return false;
        kmax = P.K > kmax ? P.K : kmax;
    }
    return kmax >= 256;
}

// KU k-tiles per barrier interval for the X3 chain kernels (gemm_mainloop_x3s): every reduction a multiple of `depth`
static int x3s_ku(const Group& g, int depth) {
    // (2 since round 4: 176-tile K = 1 536 launch 27.8 -> 26.0 us, 160 tiles 27.3 -> 24.7 us; bit-identical results)
    static const int ku = getenv("TWOG_X3S_KU") ? atoi(getenv("TWOG_X3S_KU")) : 2;
    if (ku <= 1) return 1;
    for (int i = 0; i < g.n; ++i)
        if (g.p[i].K % depth) return 1;
    return ku;
}

// Split of the reduction over workgroups for a chain launch (XS kernels). Model: a k-tile of one tile costs `unit` us of
// fp32 MFMA on a CU (64x64: 16 x 64-cycle MFMAs per SIMD = 0.49 us at the ~2.1 GHz the part sustains; 32x64: half), the
// workgroups of a launch are dealt over 256 CUs, a CU works its workgroups off one after the other, and a launch with
// S > 1 pays the hand-off (slab stores, ticket, the last arriver's S slab loads). TWOG_GEMM_XSPLIT: 0 = this model,
// 1 = never split, n > 1 = always n slices (tuning / tests).
constexpr size_t XS_COUNTER_BYTES = 16384;   // 4096 tickets at the start of the chain workspace
static int pick_xsplit(int tiles, int kmax, int bm, size_t ws_bytes) {
    static const int force = getenv("TWOG_GEMM_XSPLIT") ? atoi(getenv("TWOG_GEMM_XSPLIT")) : 0;
    if (force == 1 || tiles <= 0 || tiles > 4096 || ws_bytes <= XS_COUNTER_BYTES) return 1;
    const int kt = (kmax + BK - 1) / BK;
    const size_t tile_bytes = (size_t)bm * 64 * sizeof(float);
    const int by_ws = (int)((ws_bytes - XS_COUNTER_BYTES) / (tile_bytes * (size_t)tiles));
    if (bm == 32 && force <= 1) {
        // small batches (32-row tiles, at most one per CU): every launch is latency-bound and the hand-off costs about a
        // microsecond at this scale; measured (tools/xs_sweep.sh, 8 clips): K = 1 536: 22 -> 12 us for 4 ... 8 slices,
        // K = 1 024: 17 -> 11.5, K = 512: 10.4 ... 11.3 -> 8.6 ... 10.9 at 4 slices. Slices of four k-tiles, at most 8 of them,
        // at most two workgroups per CU.
        int S = kt / 4 < 8 ? kt / 4 : 8;
        while (S > 1 && (tiles * S > 512 || S > by_ws)) --S;
        return S < 1 ? 1 : S;
    }
    const double unit = bm == 32 ? 0.25 : 0.49;
    auto cost = [&](int S) {
        const int per_cu = (tiles * S + 255) / 256, kts = (kt + S - 1) / S;
        return per_cu * kts * unit + (S > 1 ? 3.0 + 0.35 * S : 0.0);
    };
    int best = 1;
    double best_c = cost(1);
    static const int cand[] = {2, 3, 4, 6, 8, 12, 16};
    for (int S : cand) {
        if (S > by_ws || kt / S < 2) break;
        const double c = force > 1 ? (S == force ? -1.0 : 1e30) : cost(S);
        if (c < best_c - 2.5) { best_c = c; best = S; }   // a split must be worth more than the model's error
    }
    return best;
}

// launches the XS variant when the chain workspace allows a split that pays; returns false when the caller should use
// the unsplit kernels
template <class LaunchFn>
static bool try_xsplit(Group& g, int bm, int kmax, void* chain_ws, size_t chain_ws_bytes, LaunchFn launch_fn) {
    if (!chain_ws) return false;
    const int S = pick_xsplit(g.total_tiles, kmax, bm, chain_ws_bytes);
    if (S <= 1) return false;
    const int kt = (kmax + BK - 1) / BK;
    g.k_per_split = ((kt + S - 1) / S) * BK;
    g.splitk = (kmax + g.k_per_split - 1) / g.k_per_split;
    if (g.splitk <= 1) { g.splitk = 1; g.k_per_split = kt * BK; return false; }
    g.xcnt = reinterpret_cast<unsigned*>(chain_ws);
    g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(chain_ws) + XS_COUNTER_BYTES);
    // epilogue operands (previous C, gate operands: up to 8 values per output element) in every slice: only while the
    // launch is latency-bound, i.e. its outputs are a few hundred KB (8 clips per GPU: 48 rows x 512)
    static const int early_tiles = getenv("TWOG_GEMM_XS_EARLY") ? atoi(getenv("TWOG_GEMM_XS_EARLY")) : 96;
    g.xs_early = g.total_tiles <= early_tiles ? 1 : 0;
    launch_fn(dim3(g.total_tiles, g.splitk));
    return true;
}

// XL: chain launches of the X3 64x64 class at real batches (96+ tiles). Such a launch is bound by the bytes a CU takes in
// (~0.5 us per 32-deep k-tile of a 64x64 tile whether one 8-wave workgroup or two 4-wave ones share the CU; section 8 of
// DESIGN.md), and its tile count rarely fills the chip in whole rounds: 176 tiles leave 80 CUs idle, 320 put two tiles on 64
// CUs and one on the rest. Splitting every reduction into S slices (4-wave workgroups, two resident per CU, so that one's
// prologue and hand-off hide under the other's loop) evens the per-CU byte count; the slices are combined inside the launch
// by the tile's last arriver in slice order (the LA branch of gemm_tile): deterministic, and the fused gate epilogue runs
// on the complete sum as before. TWOG_X3_XL: 0 = the model below, 1 = never, n > 1 = always n slices.
static int pick_xl_split(int tiles, int kmax, size_t ws_bytes) {
    static const int force = getenv("TWOG_X3_XL") ? atoi(getenv("TWOG_X3_XL")) : 0;
    if (force == 1 || tiles < 96 || tiles > 4096 || ws_bytes <= XS_COUNTER_BYTES || kmax % 32) return 1;
    const int kt = kmax / 32;
    const int by_ws = (int)((ws_bytes - XS_COUNTER_BYTES) / ((size_t)64 * 64 * sizeof(float) * (size_t)tiles));
    // cost in k-tile times of the busiest CU: its workgroups' k-tiles one after the other + what a split adds (hand-off,
    // the last arriver's combine, one prologue per further round). Fitted to tools/gemm_chain_bench.py on the bs64 shapes
    // (us, S = 1 / 2 / 3 / 4 / 6 / 8): 176 tiles K = 1 536: 26.0 / 28.1 / 27.4 / 25.3 / 27.7 / 28.4; 160 tiles: 25.7 / 27.7 /
    // 23.6 / 24.2 / 24.7 / 25.7; 320 tiles K = 1 536: 40.8 / 36.6 / 37.0 / 36.4 / 39.2 / 41.7; 320 tiles K = 512: 18.6 /
    // 17.6 / 18.7 / 20.2 -- the gains are a third of what the byte count alone predicts (every workgroup pays ~4 us of
    // prologue and hand-off that only its sibling on the CU can hide), so a split must promise 15 %: that leaves the 176-tile
    // BiGRU backward carry (model -12.5 %, measured -3 %) unsplit -- its 80 idle CUs are what the side stream's weight-gradient
    // GEMMs run on (ops.tggcn_backward); split, the step gains 0.13 ms and every one of those GEMMs takes 8-16 % longer
    // (bs64, same box: 63.84 / 63.93 ms at 10 %, 63.97 / 64.02 at 15 %, 64.34 without XL; 128x128-class time per step 28.56 / 27.51 / 27.61).
    auto cost = [&](int S) {
        const int r = (tiles * S + 255) / 256;
        return r * ((kt + S - 1) / S) + (S > 1 ? 3 + r : 0);
    };
    int best = 1;
    int best_c = cost(1);
    static const int min_gain = getenv("TWOG_X3_XL_GAIN") ? atoi(getenv("TWOG_X3_XL_GAIN")) : 15;   // per cent the model must promise
    static const int cand[] = {2, 3, 4};
    for (int S : cand) {
        if (S > by_ws || kt / S < 8) break;
        const int c = force > 1 ? (S == force ? -1 : (1 << 20)) : cost(S);
        if (c * 100 <= best_c * (100 - min_gain)) { best_c = c; best = S; }
    }
    return best;
}
static bool xl_setup(Group& g, int kmax, void* chain_ws, size_t chain_ws_bytes) {
    if (!chain_ws) return false;
    const int S = pick_xl_split(g.total_tiles, kmax, chain_ws_bytes);
    if (S <= 1) return false;
    const int kt = kmax / 32;
    g.k_per_split = ((kt + S - 1) / S) * 32;
    g.splitk = (kmax + g.k_per_split - 1) / g.k_per_split;
    if (g.splitk <= 1) { g.splitk = 1; g.k_per_split = ((kmax + BK - 1) / BK) * BK; return false; }
    g.xcnt = reinterpret_cast<unsigned*>(chain_ws);
    g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(chain_ws) + XS_COUNTER_BYTES);
    g.xcd_split = 0;
    return true;
}

// a_colsum requests are served by the 8-wave X3 128x128 kernels with k-major A and B only (gemm_mainloop_x3): true when the
// group has no request or takes that class
static bool colsums_served(const twog_gemm_t* pr, int n, int a_kmajor, int b_kmajor, const Group& g, bool big) {
    bool any = false, plain = true;
    for (int i = 0; i < n; ++i) {
        any = any || pr[i].a_colsum != nullptr;
        plain = plain && pr[i].batch <= 1 && pr[i].M % 4 == 0 && pr[i].A.inner <= 1 && pr[i].B.inner <= 1;   // (no grouped rows: the KG kernels have no register left)
    }
    if (!any) return true;
    static const int w8 = getenv("TWOG_GEMM_W8") ? atoi(getenv("TWOG_GEMM_W8")) : 1;
    return a_kmajor && b_kmajor && big && w8 && plain && x3_128_ok(g);
}

// 128 x 64 tiles for a chain launch of the X3 64x64 class (gemm_x3su128_kernel): a launch is bound by the operand bytes its
// busiest CU takes in -- rounds x (tile rows + tile columns) x K -- and 480 tiles of 64 x 64 (two per CU: 2 x 128) become 240
// of 128 x 64 (one per CU: 192), 320 become 160. Taken when that count falls by 15 % and at least half the chip stays busy.
// TWOG_X3_ROWS128=0: never.
static bool rows128_pays(const Group& g) {
    static const int on = getenv("TWOG_X3_ROWS128") ? atoi(getenv("TWOG_X3_ROWS128")) : 1;
    if (!on) return false;
    int t64 = 0, t128 = 0;
    for (int i = 0; i < g.n; ++i) {
        const Prob& P = g.p[i];
        if (P.M < 128 || P.batch > 1) return false;
        t64 += ((P.M + 63) / 64) * P.tiles_n;
        t128 += ((P.M + 127) / 128) * P.tiles_n;
    }
    if (t128 < 128 || t128 > 256) return false;
    const int c64 = ((t64 + 255) / 256) * 128, c128 = ((t128 + 255) / 256) * 192;
    return c128 * 100 <= c64 * 85;
}
// the tile bookkeeping of prepare_group again for another tile height
static void retile_rows(Group& g, int bm) {
    int t = 0;
    g.n_cls = 0;
    for (int i = 0; i < g.n; ++i) {
        Prob& P = g.p[i];
        P.tiles_m = (P.M + bm - 1) / bm;
        P.tile_start = t;
        const int ntiles = P.batch * P.tiles_m * P.tiles_n;
        if (i == 0 || P.K != g.p[i - 1].K) {
            g.cls_start[g.n_cls] = t;
            g.cls_ntiles[g.n_cls] = 0;
            ++g.n_cls;
        }
        g.cls_ntiles[g.n_cls - 1] += ntiles;
        t += ntiles;
    }
    for (int c = 0, rot = 0; c < g.n_cls; ++c) {
        g.cls_rot[c] = rot & 7;
        rot += g.cls_ntiles[c] & 7;
    }
    g.total_tiles = t;
}

static int gemm_impl(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* workspace,
                     size_t workspace_bytes, void* chain_ws, size_t chain_ws_bytes, void* stream);

extern "C" int twog_gemm_colsum_fused(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* workspace,
                                      size_t workspace_bytes) {
    for (int done = 0; done < n_problems; done += MAXP) {
        const int n = (n_problems - done) < MAXP ? (n_problems - done) : MAXP;
        Group g;
        int order[MAXP];
        bool big;
        int bm;
        prepare_group(problems + done, n, a_kmajor, b_kmajor, workspace, workspace_bytes, g, order, big, bm);
        if (!colsums_served(problems + done, n, a_kmajor, b_kmajor, g, big)) return 0;
    }
    return 1;
}

extern "C" int twog_gemm_f32(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* workspace,
                             size_t workspace_bytes, void* stream) {
    return gemm_impl(problems, n_problems, a_kmajor, b_kmajor, workspace, workspace_bytes, nullptr, 0, stream);
}

// The launches of a recurrent chain (few tiles, K = h ... 3h, each dependent on the previous one): `chain_ws` (>= 16 KB
// of tickets, ZERO when first handed over and then left to the library, + room for the partial tiles; see
// twog_chain_workspace_bytes) lets the library split the reduction over workgroups and combine it inside the launch.
extern "C" int twog_gemm_f32_chain(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* chain_ws,
                                   size_t chain_ws_bytes, void* stream) {
    return gemm_impl(problems, n_problems, a_kmajor, b_kmajor, nullptr, 0, chain_ws, chain_ws_bytes, stream);
}

extern "C" size_t twog_chain_workspace_bytes(void) {
    return XS_COUNTER_BYTES + (size_t)48 * 1024 * 1024;   // tickets + 3 072 partial 64x64 tiles
}

static int gemm_impl(const twog_gemm_t* problems, int n_problems, int a_kmajor, int b_kmajor, void* workspace,
                     size_t workspace_bytes, void* chain_ws, size_t chain_ws_bytes, void* stream) {
    if (n_problems <= 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {   // a_colsum requests are checked for EVERY chunk before the first launch: refused as a whole (-5), never half done
        bool any_cs = false;
        for (int i = 0; i < n_problems; ++i) any_cs = any_cs || problems[i].a_colsum != nullptr;
        if (any_cs && (chain_ws || !twog_gemm_colsum_fused(problems, n_problems, a_kmajor, b_kmajor, workspace, workspace_bytes))) return -5;
    }
    int done = 0;
    while (done < n_problems) {
        const int n = (n_problems - done) < MAXP ? (n_problems - done) : MAXP;
        const twog_gemm_t* pr = problems + done;
        Group g;
        int order[MAXP];
        bool big;
        int bm;
        prepare_group(pr, n, a_kmajor, b_kmajor, workspace, workspace_bytes, g, order, big, bm);
        static const int depth = getenv("TWOG_GEMM_DEPTH") ? atoi(getenv("TWOG_GEMM_DEPTH")) : 0;  // tuning knob
        const int d128 = depth ? (depth & 3) : 2, d64 = depth ? ((depth >> 2) & 3) : 2;
        int rc;
        // 128x128 tiles run with 8 waves (4 x 2 grid, 32x64 per wave): four waves per SIMD hide the LDS / barrier
        // latencies that a 4-wave tile (two per SIMD) exposes: +5 % on every big shape (TWOG_GEMM_W8=0: 4 waves)
        static const int w8 = getenv("TWOG_GEMM_W8") ? atoi(getenv("TWOG_GEMM_W8")) : 1;
        bool grouped = false;  // grouped k-major rows keep the 4-wave tiles: their pointer-carrying 8-wave kernels exceed
                               // 128 VGPRs (3 waves/SIMD) and measure 7 % slower
        for (int i = 0; i < n; ++i) grouped = grouped || (a_kmajor && pr[i].A.inner > 1) || (b_kmajor && pr[i].B.inner > 1);
        g_last_class = (big ? TWOG_GEMM_CLASS_TILE128 : 0) | (big && w8 && !grouped ? TWOG_GEMM_CLASS_WAVES8 : 0) |
                       (grouped ? TWOG_GEMM_CLASS_KG : 0) | (g.splitk > 1 ? TWOG_GEMM_CLASS_SPLITK : 0);
        // few 64x64 tiles (at most ~1.5 per CU) with a reduction worth splitting: 8-wave workgroups, k-split inside
        static const int ks_on = getenv("TWOG_GEMM_KS") ? atoi(getenv("TWOG_GEMM_KS")) : 1;
        int kmax_ = 0;
        for (int i = 0; i < n; ++i) kmax_ = pr[i].K > kmax_ ? pr[i].K : kmax_;
        const bool ks = ks_on && !big && !a_kmajor && !grouped && g.splitk == 1 && g.total_tiles <= 384 && kmax_ >= 256;
        if (!big && bm == 64 && !grouped && chain_ws && x3s_ok(g, a_kmajor, 32) && rows128_pays(g)) {
            retile_rows(g, 128);
            g_last_class |= TWOG_GEMM_CLASS_X3 | TWOG_GEMM_CLASS_WAVES8;
            dim3 grid(g.total_tiles, 1), block(512);
            if (b_kmajor) hipLaunchKernelGGL((gemm_x3su128_kernel<true>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_x3su128_kernel<false>), grid, block, 0, st, g);
            TWOG_CHECK_LAUNCH();
            done += n;
            continue;
        }
        if (!big && bm == 64 && !grouped && chain_ws && x3s_ok(g, a_kmajor, 32) && xl_setup(g, kmax_, chain_ws, chain_ws_bytes)) {
            g_last_class |= TWOG_GEMM_CLASS_X3 | TWOG_GEMM_CLASS_XSPLIT;
            dim3 grid(g.total_tiles, g.splitk), block(256);
            if (b_kmajor) hipLaunchKernelGGL((gemm_x3su_kernel<true, 1, 2>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_x3su_kernel<false, 1, 2>), grid, block, 0, st, g);
            TWOG_CHECK_LAUNCH();
            done += n;
            continue;
        }
        if (bm == 32) {   // decided in prepare_group; implies the conditions of `ks`
            g_last_class |= TWOG_GEMM_CLASS_KSPLIT | TWOG_GEMM_CLASS_ROWS32;
            if (try_xsplit(g, 32, kmax_, chain_ws, chain_ws_bytes, [&](dim3 grid) {
                    if (b_kmajor) hipLaunchKernelGGL((gemm_xs32_kernel<true, 2>), grid, dim3(256), 0, st, g);
                    else hipLaunchKernelGGL((gemm_xs32_kernel<false, 2>), grid, dim3(256), 0, st, g);
                })) {
                g_last_class |= TWOG_GEMM_CLASS_XSPLIT;
                TWOG_CHECK_LAUNCH();
                done += n;
                continue;
            }
            dim3 grid(g.total_tiles, 1), block(256);
            if (b_kmajor) hipLaunchKernelGGL((gemm_ks32_kernel<true, 2>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_ks32_kernel<false, 2>), grid, block, 0, st, g);
            TWOG_CHECK_LAUNCH();
            done += n;
            continue;
        }
        if (ks) {
            g_last_class |= TWOG_GEMM_CLASS_KSPLIT;
            if (try_xsplit(g, 64, kmax_, chain_ws, chain_ws_bytes, [&](dim3 grid) {
                    if (b_kmajor) hipLaunchKernelGGL((gemm_xs_kernel<true, 2>), grid, dim3(512), 0, st, g);
                    else hipLaunchKernelGGL((gemm_xs_kernel<false, 2>), grid, dim3(512), 0, st, g);
                })) {
                g_last_class |= TWOG_GEMM_CLASS_XSPLIT;
                TWOG_CHECK_LAUNCH();
                done += n;
                continue;
            }
            dim3 grid(g.total_tiles, 1), block(512);
            if (x3s_ok(g, a_kmajor, 32)) {
                g_last_class |= TWOG_GEMM_CLASS_X3;
                if (g.total_tiles <= 256 && x3s_ku(g, 64) == 2) {
                    if (b_kmajor) hipLaunchKernelGGL((gemm_x3su_kernel<true, 2, 2>), grid, block, 0, st, g);
                    else hipLaunchKernelGGL((gemm_x3su_kernel<false, 2, 2>), grid, block, 0, st, g);
                } else if (b_kmajor) hipLaunchKernelGGL((gemm_x3s_kernel<true, 2>), grid, block, 0, st, g);
                else hipLaunchKernelGGL((gemm_x3s_kernel<false, 2>), grid, block, 0, st, g);
            } else if (b_kmajor) hipLaunchKernelGGL((gemm_ks_kernel<true, 2>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_ks_kernel<false, 2>), grid, block, 0, st, g);
            TWOG_CHECK_LAUNCH();
            done += n;
            continue;
        }
        static const int x3_try = getenv("TWOG_GEMM_X3") ? atoi(getenv("TWOG_GEMM_X3")) : 1;   // grouped rows: X3 has 8-wave variants
        if (big && w8 && (!grouped || x3_try)) rc = d128 == 2 ? launch<128, 128, 512, 2>(g, a_kmajor, b_kmajor, st) : launch<128, 128, 512, 1>(g, a_kmajor, b_kmajor, st);
        else if (big) rc = d128 == 2 ? launch<128, 128, 256, 2>(g, a_kmajor, b_kmajor, st) : launch<128, 128, 256, 1>(g, a_kmajor, b_kmajor, st);
        else if (!big && x3s_ok(g, a_kmajor, 16)) {
            g_last_class |= TWOG_GEMM_CLASS_X3;
            dim3 grid(g.total_tiles, 1), block(256);
            if (x3s_ku(g, 32) == 2) {
                if (b_kmajor) hipLaunchKernelGGL((gemm_x3su_kernel<true, 1, 2>), grid, block, 0, st, g);
                else hipLaunchKernelGGL((gemm_x3su_kernel<false, 1, 2>), grid, block, 0, st, g);
            } else if (b_kmajor) hipLaunchKernelGGL((gemm_x3s_kernel<true, 1>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((gemm_x3s_kernel<false, 1>), grid, block, 0, st, g);
            TWOG_CHECK_LAUNCH();
            rc = 0;
        }
        else rc = d64 == 2 ? launch<64, 64, 256, 2>(g, a_kmajor, b_kmajor, st) : launch<64, 64, 256, 1>(g, a_kmajor, b_kmajor, st);
        if (rc) return rc;
        if (g_last_class_x3) g_last_class |= TWOG_GEMM_CLASS_X3 | TWOG_GEMM_CLASS_WAVES8;
        done += n;
    }
    return 0;
}


// Gate-fused launch for the recurrent backward chains (see GateArgs): problems are C += A B with A row-major and B k-major;
// gates[i].rows > 0 asks for the gate backward `gates[i]` in the epilogue of problem i (its C must be the gate's dh_prev
// buffer; the gate's dh2 is taken from the accumulators, not from memory). Returns 1 without launching when this chunk
// is not served by the fused kernel (tile class, grouped rows, too many partial slots) -- dry_run only asks that.
int twog_internal_gemm_gate_bwd(const twog_gemm_t* pr, int n, const twog_gru_step_bwd_t* gates, float* const* du_part,
                                int dry_run, void* chain_ws, size_t chain_ws_bytes, void* stream) {
    if (n <= 0 || n > MAXP) return 1;
    Group g;
    int order[MAXP];
    bool big;
    int bm;
    prepare_group(pr, n, 0, 1, nullptr, 0, g, order, big, bm);
    if (big || g.splitk != 1) return 1;
    auto span31 = [](const twog_rows_t& m, int rows, int64_t width) {  // largest element offset fits 31 bits
        const int inner = m.inner > 1 ? m.inner : 1;
        const int64_t last = (int64_t)((rows - 1) / inner) * m.ld_outer + (int64_t)((rows - 1) % inner) * (m.inner > 1 ? m.ld_inner : 0) + width;
        return m.ld_outer >= 0 && m.ld_inner >= 0 && last < (int64_t(1) << 31);
    };
    for (int i = 0; i < n; ++i) {
        if (pr[i].B.inner > 1 || pr[i].batch > 1 || pr[i].bias || pr[i].act) return 1;
        const twog_gru_step_bwd_t& G = gates[i];
        if (G.rows <= 0) continue;
        if ((pr[i].N + 63) / 64 > 8 || G.rows != pr[i].M || G.hidden != pr[i].N || G.rows >= (1 << 22)) return 1;
        const int inner = G.dh.inner > 1 ? G.dh.inner : 1;
        auto same = [&](const twog_rows_t& m) { return (m.inner > 1 ? m.inner : 1) == inner; };
        if (!same(G.save) || !same(G.dgi) || !same(G.dgh) || (G.h_prev.ptr && !same(G.h_prev))) return 1;
        if (G.u && (G.u_inner > 1 ? G.u_inner : 1) != inner) return 1;
        if (G.dh_prev.inner > 1 || G.dh_prev.ptr != pr[i].C.ptr || G.dh_prev_accumulate) return 1;
        const int H3 = 3 * G.hidden;
        if (!span31(G.dh, G.rows, G.hidden) || !span31(G.save, G.rows, 4 * G.hidden) || !span31(G.dgi, G.rows, H3) ||
            !span31(G.dgh, G.rows, H3) || !span31(G.dh_prev, G.rows, G.hidden) ||
            (G.h_prev.ptr && !span31(G.h_prev, G.rows, G.hidden)))
            return 1;
        if (G.u && ((int64_t)((G.rows - 1) / inner) * G.u_ld_outer + (int64_t)((G.rows - 1) % inner) * G.u_ld_inner >= (int64_t(1) << 31)))
            return 1;
    }
    if (dry_run) return 0;
    GateArgs ga;
    for (int i = 0; i < n; ++i) {
        const int src = order[i];
        ga.gate_of[i] = gates[src].rows > 0 ? i : -1;
        ga.s[i] = gates[src];
        ga.du_part[i] = (gates[src].rows > 0 && gates[src].du && du_part) ? du_part[src] : nullptr;
    }
    for (int i = n; i < MAXP; ++i) { ga.gate_of[i] = -1; ga.du_part[i] = nullptr; }
    static const int depth = getenv("TWOG_GEMM_DEPTH") ? atoi(getenv("TWOG_GEMM_DEPTH")) : 0;
    const int d64 = depth ? ((depth >> 2) & 3) : 2;
    dim3 grid(g.total_tiles, 1), block(256);
    g_last_class = TWOG_GEMM_CLASS_GATE;
    static const int ks_on = getenv("TWOG_GEMM_KS") ? atoi(getenv("TWOG_GEMM_KS")) : 1;
    int kmax = 0;
    for (int i = 0; i < n; ++i) kmax = pr[i].K > kmax ? pr[i].K : kmax;
    if (bm == 64 && chain_ws && x3s_ok(g, 0, 32) && xl_setup(g, kmax, chain_ws, chain_ws_bytes)) {
        g_last_class |= TWOG_GEMM_CLASS_X3 | TWOG_GEMM_CLASS_XSPLIT;
        hipLaunchKernelGGL((gemm_gate_bwd_x3su_kernel<1, 2>), dim3(g.total_tiles, g.splitk), dim3(256), 0, (hipStream_t)stream, g, ga);
        TWOG_CHECK_LAUNCH();
        return 0;
    }
    if (bm == 32) {
        g_last_class |= TWOG_GEMM_CLASS_KSPLIT | TWOG_GEMM_CLASS_ROWS32;
        if (try_xsplit(g, 32, kmax, chain_ws, chain_ws_bytes, [&](dim3 xgrid) {
                hipLaunchKernelGGL(gemm_gate_bwd_xs32_kernel<2>, xgrid, dim3(256), 0, (hipStream_t)stream, g, ga);
            })) {
            g_last_class |= TWOG_GEMM_CLASS_XSPLIT;
            TWOG_CHECK_LAUNCH();
            return 0;
        }
        hipLaunchKernelGGL(gemm_gate_bwd_ks32_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, g, ga);
        TWOG_CHECK_LAUNCH();
        return 0;
    }
    if (ks_on && g.total_tiles <= 384 && kmax >= 256) {
        g_last_class |= TWOG_GEMM_CLASS_KSPLIT;
        if (try_xsplit(g, 64, kmax, chain_ws, chain_ws_bytes, [&](dim3 xgrid) {
                hipLaunchKernelGGL(gemm_gate_bwd_xs_kernel<2>, xgrid, dim3(512), 0, (hipStream_t)stream, g, ga);
            })) {
            g_last_class |= TWOG_GEMM_CLASS_XSPLIT;
            TWOG_CHECK_LAUNCH();
            return 0;
        }
        if (x3s_ok(g, 0, 32)) {
            g_last_class |= TWOG_GEMM_CLASS_X3;
            if (g.total_tiles <= 256 && x3s_ku(g, 64) == 2) hipLaunchKernelGGL((gemm_gate_bwd_x3su_kernel<2, 2>), grid, dim3(512), 0, (hipStream_t)stream, g, ga);
            else hipLaunchKernelGGL(gemm_gate_bwd_x3s_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, g, ga);
        } else
        hipLaunchKernelGGL(gemm_gate_bwd_ks_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, g, ga);
        TWOG_CHECK_LAUNCH();
        return 0;
    }
    if (x3s_ok(g, 0, 16)) {
        g_last_class |= TWOG_GEMM_CLASS_X3;
        hipLaunchKernelGGL(gemm_gate_bwd_x3s_kernel<1>, grid, block, 0, (hipStream_t)stream, g, ga);
    } else if (d64 == 2) hipLaunchKernelGGL(gemm_gate_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, g, ga);
    else hipLaunchKernelGGL(gemm_gate_bwd_kernel<1>, grid, block, 0, (hipStream_t)stream, g, ga);
    TWOG_CHECK_LAUNCH();
    return 0;
}

// Fused forward step of the recurrent chains (gemm_gru_fwd_kernel). Problem i: gh[i] is the launch the unfused path would
// make for W_hh h_prev + b_hh (A = previous states, B = W_hh [3h][h], bias = b_hh); gim[i] the one for the aggregated
// messages (M == 0: none; A = messages, B = W_ih[:, msg]); st[i] the gate step that would follow (its gh / gi2 operands
// are not read: the products stay in the accumulators). Returns 1 without launching when the shapes are not served
// (hidden size or message width not a multiple of the 32-wide k-tile, unaligned operands, mixed row groupings).
int twog_internal_gru_fwd_mode(void) {
    const char* e = getenv("TWOG_GRU_FWD_FUSION");
    return e ? atoi(e) : 3;
}

int twog_internal_gemm_gru_fwd(const twog_gemm_t* gh, const twog_gemm_t* gim, const twog_gru_step_t* st, int n,
                               int dry_run, void* stream) {
    // TWOG_GRU_FWD_FUSION: bit 0 = chains without message products (frame-level BiGRU), bit 1 = with (segment level),
    // bit 2 = always (skip the cost model below). Default 3: both levels, each launch decided by the model.
    const int mode = twog_internal_gru_fwd_mode();   // read per call: tests switch it inside one process
    static const int ksplit = getenv("TWOG_GRU_FWD_KS") ? atoi(getenv("TWOG_GRU_FWD_KS")) : 2;
    const bool with_msgs = gim != nullptr;
    if (!(mode & (with_msgs ? 2 : 1)) || n <= 0 || n > MAXP) return 1;
    const int h = st[0].hidden;
    if (h < 32 || h % BK) return 1;
    auto span31 = [](const twog_rows_t& m, int rows, int64_t width) {
        const int inner = m.inner > 1 ? m.inner : 1;
        const int64_t last = (int64_t)((rows - 1) / inner) * m.ld_outer + (int64_t)((rows - 1) % inner) * (m.inner > 1 ? m.ld_inner : 0) + width;
        return m.ld_outer >= 0 && m.ld_inner >= 0 && last < (int64_t(1) << 31);
    };
    GruFwdGroup g;
    g.n = n; g.hidden = h; g.tiles_n = (h + 63) / 64;
    int rt = 0;
    for (int i = 0; i < n; ++i) {
        const twog_gemm_t& q = gh[i];
        const twog_gru_step_t& S = st[i];
        if (S.hidden != h || S.rows <= 0 || S.rows != q.M || S.rows >= (1 << 22)) return 1;
        if (q.N != 3 * h || q.K != h || q.act || q.accumulate || q.batch > 1 || q.B.inner > 1) return 1;
        if (!vec_ok(q.A, 0, h, q.M) || !vec_ok(q.B, 0, h, 3 * h)) return 1;
        const bool m2 = gim && gim[i].M > 0;
        if (m2) {
            const twog_gemm_t& q2 = gim[i];
            if (q2.M != q.M || q2.N != 3 * h || q2.K <= 0 || q2.K % BK || q2.act || q2.batch > 1 || q2.B.inner > 1 || q2.bias) return 1;
            if (!vec_ok(q2.A, 0, q2.K, q2.M) || !vec_ok(q2.B, 0, q2.K, 3 * h)) return 1;
        }
        const int inner = S.gi.inner > 1 ? S.gi.inner : 1;
        auto same = [&](const twog_rows_t& m) { return (m.inner > 1 ? m.inner : 1) == inner; };
        if (!same(S.h_out) || !same(S.save) || (S.h_prev.ptr && !same(S.h_prev))) return 1;
        if (S.h_prev.ptr && (S.h_prev.ptr != q.A.ptr || S.h_prev.ld_outer != q.A.ld_outer || S.h_prev.ld_inner != q.A.ld_inner)) return 1;
        if (S.u && (S.u_inner > 1 ? S.u_inner : 1) != inner) return 1;
        if (!span31(S.gi, S.rows, 3 * h) || !span31(S.h_out, S.rows, h) || !span31(S.save, S.rows, 4 * h) ||
            (S.h_prev.ptr && !span31(S.h_prev, S.rows, h)))
            return 1;
        if (S.u && ((int64_t)((S.rows - 1) / inner) * S.u_ld_outer + (int64_t)((S.rows - 1) % inner) * S.u_ld_inner >= (int64_t(1) << 31)))
            return 1;
        GruFwdProb& P = g.p[i];
        P.A = q.A; P.B = q.B;
        if (m2) { P.A2 = gim[i].A; P.B2 = gim[i].B; P.K2 = gim[i].K; }
        else { P.A2 = q.A; P.B2 = q.B; P.K2 = 0; }
        P.gi = S.gi; P.h_out = S.h_out; P.save = S.save;
        P.b_hh = q.bias; P.u = S.u;
        P.u_ld_outer = (int)S.u_ld_outer; P.u_ld_inner = (int)S.u_ld_inner;
        P.rows = S.rows; P.has_prev = S.h_prev.ptr ? 1 : 0; P.rt_start = rt; P.inner = inner;
        rt += (S.rows + 63) / 64;
    }
    if (!(mode & 4)) {
        // Cost model (us of fp32 MFMA work per CU at 614 GFLOP/s, tiles dealt in rounds over 256 CUs; measured at BASELINE
        // size, profiles/HISTORY.md section 8): the fused launch runs rounds of 64 x 192 x K tiles, the pair it replaces rounds of
        // 64 x 64 x K tiles plus a gate launch (~7 us). bs64, h = 512: frame level 20.5 vs 3 x 6.8 + 7 -> fused (measured
        // 39 vs 43 us); segment level (K = 1 536, 160 tiles) 61 vs 2 x 20.5 + 7 -> unfused (measured +2.5 ms per step
        // fused); 8 clips: 20.5 vs 6.8 + 7 -> unfused (measured 31.5 vs 39.7 ms per step forced). Short reductions stay
        // unfused whatever the model says: at h = 64 (Bimanual layout) the 8-wave tile with its k-group exchange costs
        // more than the two small launches it replaces (measured 20.7 vs 19.3 ms per step).
        int kmax = h;
        for (int i = 0; i < n; ++i) kmax = (h + g.p[i].K2) > kmax ? (h + g.p[i].K2) : kmax;
        const double per_tile = 64.0 * 64.0 * kmax * 2.0 / 614.0e3;   // us for a 64 x 64 x K tile
        const int unfused_tiles = rt * ((3 * h + 63) / 64), fused_tiles = rt * g.tiles_n;
        const double fused = 3.0 * per_tile * ((fused_tiles + 255) / 256);
        const double unfused = per_tile * ((unfused_tiles + 255) / 256) + 7.0;
        if (fused > unfused || kmax < 256) return 1;
    }
    if (dry_run) return 0;
    for (int i = n; i < MAXP; ++i) { g.p[i] = g.p[0]; g.p[i].rt_start = 0x7fffffff; }
    g_last_class = TWOG_GEMM_CLASS_GRUFWD;
    static const int x3_on = (getenv("TWOG_GEMM_X3") ? atoi(getenv("TWOG_GEMM_X3")) : 1) && (getenv("TWOG_GEMM_X3S") ? atoi(getenv("TWOG_GEMM_X3S")) : 1) &&
                             (getenv("TWOG_GEMM_X3G") ? atoi(getenv("TWOG_GEMM_X3G")) : 1);
    bool x3 = x3_on && ksplit == 2 && h >= 256 && h % 32 == 0;   // whole 32-deep k-tiles of both products
    for (int i = 0; i < n; ++i) x3 = x3 && g.p[i].K2 % 32 == 0;
    if (x3) {
        g_last_class |= TWOG_GEMM_CLASS_X3;
        hipLaunchKernelGGL((gemm_gru_fwd_kernel<2, 2, true>), dim3(rt * g.tiles_n), dim3(512), 0, (hipStream_t)stream, g);
    } else if (ksplit == 2) hipLaunchKernelGGL((gemm_gru_fwd_kernel<2, 2>), dim3(rt * g.tiles_n), dim3(512), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((gemm_gru_fwd_kernel<2, 1>), dim3(rt * g.tiles_n), dim3(256), 0, (hipStream_t)stream, g);
    TWOG_CHECK_LAUNCH();
    return 0;
}

#ifdef TWOG_STAMPS
extern "C" int twog_debug_stamps(unsigned long long* out) {   // diagnostic builds only: [wave][8] cycle sums of workgroup 0
    return -(int)hipMemcpyFromSymbol(out, HIP_SYMBOL(twog_stamp_buf), sizeof(unsigned long long) * 16 * 8);
}
#endif

extern "C" const char* twog_version(void) {
    return "lib2ggcn_hip gfx950 fp32 gemm on bf16x3 MFMA(32x32x16) / fp32 MFMA(32x32x2), tiles 128x128 / 64x64 / 32x64";
}
