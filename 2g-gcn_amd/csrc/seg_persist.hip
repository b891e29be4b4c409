// Segment-level gated bidirectional recurrence with message passing as ONE persistent launch (small batches).
//
// Reference: the segment loop of TGGCN.forward (vhoi/models.py:785-880): per (tf, tb) pair every human / object gathers
// attention-weighted messages from the previously committed segment states of the other entities
// (_humans_to_human_segment_message :1051, _humans_to_object_segment_message :1145, _objects_to_human_segment_message
// :1239, _objects_to_object_segment_message :1334; compute_non_relational_message :1693, compute_attention_weights
// :1721), concatenates them to its frame-level input and takes one gated GRUCell step (_bidirectional_step :1535-1564).
// Same inputs, outputs and saved tensors as twog_segrnn_fwd (segrnn.hip), which issues three to four launches per time
// step: at 8 clips per GPU a step is a chain of launch boundaries around a handful of tiles (50 us per step, 58 % of the
// whole training step at BASELINE configs[1]).
//
// Here the chip is partitioned once for the whole sequence. The work of a (direction, clip chunk) GROUP is dealt to
// 4 x h/16 workgroups, one per compute unit: for every slice of 16 hidden units four ROLES
//   P1a  sender MLPs hh (on the humans' previous states) and oh (on the objects'), attention weights, the aggregated
//        messages received by HUMANS for its 16 columns of both blocks, and W_hh h_prev of the human cell (3 gates x 16);
//   P1b  the same for what OBJECTS receive: sender MLPs ho and oo, aggregated messages, W_hh h_prev of the object cell;
//   P2h  W_ih[:, messages] of the human cell on the complete aggregated-message rows (3 gates x 16 units), the gate
//        math, the blend with the hard segment gate u, the 16 columns of h_t;  P2o the same for objects.
// A step is two phases -- P1 needs every column of h_{t-1} of both kinds (A operand of its products and the attention
// scores, which every P1 workgroup computes redundantly for its chunk on the matrix cores: the Gram matrix of the
// chunk's state rows), P2 needs every column of the aggregated messages -- so two hand-offs per step and direction,
// each: write-through (sc1) 16-byte stores, every storing wave drains them (s_waitcnt vmcnt(0)), workgroup barrier, one
// lane adds to the role's agent-scope counter of the group; consumer: one lane polls the counter with sc1 loads, a
// workgroup barrier, sc1 loads of the handed-off bytes only (MI355X_MICROARCH.md, hand-off table row 1; one workgroup
// per compute unit, enforced by the LDS request). Every wait is bounded and fails soft (persist_common.h).
//
// Arithmetic: the exact 3 x bf16 split of both operands, six of nine products on v_mfma_f32_16x16x32_bf16, fp32
// accumulation with the h.h product and the five small ones in separate accumulators (as the 64x64 GEMM class and
// gru_persist.hip); the reduction of a product is split over the four waves of a workgroup by k-blocks and combined
// through LDS in wave order: fixed summation order, bit-reproducible. Weights are streamed from L2 / Infinity Cache
// every step (704 KB per slice and direction at h = 512: more than LDS holds), 160-192 KB per workgroup and step.
#include "twog_common.h"
#include "persist_common.h"

// The wave index is read through a SCALAR register: every per-wave condition (`kb = wave + 4 j < nkb`: is this k-block slot
// of the wave in use?) is then a scalar branch. Left in a vector register (round 5) the compiler has to treat those
// conditions as divergent and predicates the slot's code with EXEC -- which MFMAs ignore, and under which register copies /
// zero-initialisations do not execute: a variant of P2 then multiplied registers nobody had written.
// -DTWOG_SP_VECTOR_WAVE rebuilds that form (`make diag`, tools/persist_stress.py): never part of the shipped library.
#ifdef TWOG_SP_VECTOR_WAVE
#define TWOG_SP_WAVE_INDEX ((int)(threadIdx.x >> 6))
#else
#define TWOG_SP_WAVE_INDEX __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));

constexpr int SC1 = 16;          // aux bits of the buffer instructions: sc1
constexpr int RS = 20;           // floats per row of a 16 x 16 result tile in LDS (16-byte aligned, 4 banks apart)
constexpr int CNT_STRIDE = 32;   // uint32 between two counters (a 128-byte line each)
constexpr int SYNC_WORDS = 4096; // uint32 of the caller's sync buffer; the error word is the last line
constexpr int ERR_WORD = SYNC_WORDS - 32;
constexpr int MAXH = 4, MAXO = 12;

struct SegArgs {
    int bs, T, H, O, h;
    int cpc, n_chunks;           // clips per chunk, chunks per direction
    unsigned pair_mask;          // bit (i * 8 + j), i <= j: Gram tile (i, j) of the chunk's row tiles holds a same-clip pair
    float scale;
    int spin_limit;
    const float* gi[2];          // [kind] [bs][T][E][6h]
    const float* u[2];           // [bs][T][E]
    const float* mask;           // [bs][O]
    const float* w_hh[2][2];     // [kind][dir] [3h][h]
    const float* b_hh[2][2];
    const float* w_ihm[2][2];    // [kind][dir] &weight_ih[0][first message column]
    int64_t ld_ih[2];
    const float* w_s[4];         // sender MLPs hh, ho (on human states), oh, oo (on object states): [h][h]
    const float* b_s[4];
    float* hs[2];                // [bs][T][E][2h]
    float* save[2];              // [2][bs][T][E][4h]
    float* msrc[2];              // [2][bs][T][E][2h]  (hh | ho), (oh | oo)
    float* mg[2];                // [2][bs][T][E][2h]  (hh | oh), (ho | oo)
    float* gh[2];                // [2][bs*E][3h] scratch: W_hh h_prev + b_hh of the current step
    float* att;                  // [2][T][bs][natt]
    unsigned* cnt;
    unsigned* error;
};

__device__ __forceinline__ uint32_t pack_hi16(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302); }

// eight consecutive fp32 values -> the three bf16 planes of an MFMA fragment (element j = value j), by truncation: exact
struct Planes { bf16x8 h, m, l; };
__device__ __forceinline__ Planes split8(const f32x4 a, const f32x4 b) {
#ifdef TWOG_SP_PROBE_NOSPLIT   // timing probe only (wrong results): what the split of the streamed fragments costs a step
    { Planes q; q.h = __builtin_bit_cast(bf16x8, a); q.m = __builtin_bit_cast(bf16x8, b); q.l = q.h; return q; }
#endif
    uint32_t x[8], r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float f0 = i < 4 ? a[i] : b[i - 4];
        x[i] = __float_as_uint(f0);
        const float f1 = f0 - __uint_as_float(x[i] & 0xffff0000u);
        r1[i] = __float_as_uint(f1);
        r2[i] = __float_as_uint(f1 - __uint_as_float(r1[i] & 0xffff0000u));
    }
    Planes p;
    p.h = __builtin_bit_cast(bf16x8, i32x4{(int)pack_hi16(x[0], x[1]), (int)pack_hi16(x[2], x[3]), (int)pack_hi16(x[4], x[5]), (int)pack_hi16(x[6], x[7])});
    p.m = __builtin_bit_cast(bf16x8, i32x4{(int)pack_hi16(r1[0], r1[1]), (int)pack_hi16(r1[2], r1[3]), (int)pack_hi16(r1[4], r1[5]), (int)pack_hi16(r1[6], r1[7])});
    p.l = __builtin_bit_cast(bf16x8, i32x4{(int)pack_hi16(r2[0], r2[1]), (int)pack_hi16(r2[2], r2[3]), (int)pack_hi16(r2[4], r2[5]), (int)pack_hi16(r2[6], r2[7])});
    return p;
}

struct Acc { f32x4 hi, lo; };
__device__ __forceinline__ void acc_zero(Acc& c) { c.hi = f32x4{0.f, 0.f, 0.f, 0.f}; c.lo = f32x4{0.f, 0.f, 0.f, 0.f}; }
// c += A B with A, B given as planes: h.h into `hi`, the five small products into `lo`
__device__ __forceinline__ void mac6(Acc& c, const Planes& a, const Planes& b) {
    c.hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c.hi, 0, 0, 0);
    c.lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c.lo, 0, 0, 0);
    c.lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c.lo, 0, 0, 0);
    c.lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c.lo, 0, 0, 0);
    c.lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c.lo, 0, 0, 0);
    c.lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c.lo, 0, 0, 0);
}

// the same six products into ONE accumulator
__device__ __forceinline__ void mac6_one(f32x4& c, const Planes& a, const Planes& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld_sc1(const __amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, SC1));
}
__device__ __forceinline__ void st_sc1(const __amdgpu_buffer_rsrc_t rs, uint32_t byte_off, const f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)byte_off, 0, SC1);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0xffffffff, 0x00020000);
}

// weight fragment: lane l holds W[row0 + (l & 15)][k0 + 8 (l >> 4) + j], j = 0..7
__device__ __forceinline__ Planes load_w(const float* w, int64_t ld, int row0, int k0, int lane) {
    const float* p = w + (int64_t)(row0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4);
    return split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4));
}

// raw (fp32) weight fragments, kept in registers for the whole sequence: lane l holds W[row0 + (l & 15)][k0 + 8 (l >> 4) + j]
// (row-major operand) or W[k0 + 8 (l >> 4) + j][col0 + (l & 15)] (k-major operand), j = 0..7
struct WFrag { f32x4 a, b; };
__device__ __forceinline__ WFrag load_w_raw(const float* w, int64_t ld, int row0, int k0, int lane) {
    const float* p = w + (int64_t)(row0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4);
    WFrag f;
    f.a = *reinterpret_cast<const f32x4*>(p);
    f.b = *reinterpret_cast<const f32x4*>(p + 4);
    return f;
}
__device__ __forceinline__ WFrag load_wk_raw(const float* w, int64_t ld, int k0, int col0, int lane) {
    const float* p = w + (int64_t)(k0 + 8 * (lane >> 4)) * ld + col0 + (lane & 15);
    WFrag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.a[j] = p[(int64_t)j * ld]; f.b[j] = p[(int64_t)(j + 4) * ld]; }
    return f;
}
__device__ __forceinline__ WFrag wfrag_zero() { WFrag f; f.a = f32x4{0.f, 0.f, 0.f, 0.f}; f.b = f.a; return f; }

// c += A B on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32 products, fp32 accumulate): the 32 k values of a
// fragment pair are consumed 4 at a time, lane group g4 supplying k = 8 g4 + j in step j for BOTH operands (any pairing of
// k values to (step, lane group) slots is a valid dot product as long as A and B agree). No operand split: where a weight
// fragment meets only one or two row tiles (P2, Q1, Q2) the 3 x bf16 form spends more vector cycles splitting the fragment
// than the matrix pipe saves, and it needs 36 more registers per lane for the planes.
__device__ __forceinline__ void mac_f32(f32x4& c, const f32x4 a0, const f32x4 a1, const WFrag& w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], w.a[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], w.b[j], c, 0, 0, 0);
}

// The reduction loop of one product, this wave's k-blocks (kb = wave, wave + 4, ...; at most NJ of them): the A fragments of
// MK row tiles come from a handed-off buffer (sc1 loads, byte offsets off[i] + 128 kb), four k-blocks in flight beside the
// four being multiplied (a round trip to the other XCDs' L2 is ~2 us: with two in flight the loop waited for every pair); body(j, A) multiplies k-block number j of this wave with the weights the caller keeps in registers.
template <int NJ, int MK, int CHMAX = 4, class F>
__device__ __forceinline__ void k_stream(const __amdgpu_buffer_rsrc_t rs, const uint32_t (&off)[MK], int nkb, int wave, F&& body) {
    constexpr int CH = NJ < CHMAX ? NJ : CHMAX, NG = (NJ + CH - 1) / CH;
    typedef f32x4 Buf[CH][MK][2];
#ifdef TWOG_SP_P2_X3_ALLJ   // root-cause build: every k-block slot multiplied (zero weights beyond the reduction), see below
    Buf b0 = {}, b1 = {};
#else
    Buf b0, b1;   // two named buffers (no run-time buffer index: that would put the fragments into scratch memory)
#endif
    auto load = [&](int g, Buf& buf) {
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int j = g * CH + jj, kb = wave + 4 * j;
            if (j < NJ && kb < nkb) {
#pragma unroll
                for (int i = 0; i < MK; ++i) {
                    buf[jj][i][0] = ld_sc1(rs, off[i] + 128u * kb);
                    buf[jj][i][1] = ld_sc1(rs, off[i] + 128u * kb + 16u);
                }
            }
        }
    };
    auto use = [&](int g, const Buf& buf) {
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int j = g * CH + jj, kb = wave + 4 * j;
#ifdef TWOG_SP_P2_X3_ALLJ
            (void)kb;
            if (j < NJ) body(j, buf[jj]);
#else
            if (j < NJ && kb < nkb) body(j, buf[jj]);
#endif
        }
    };
    load(0, b0);
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) load(g + 1, b1);
        use(g, b0);
        if (g + 1 < NG) {
            if (g + 2 < NG) load(g + 2, b0);
            use(g + 1, b1);
        }
    }
}

constexpr int KW1 = 4;    // k-blocks per wave at most: a reduction over h (h = 512: 16 k-blocks, 4 waves)
constexpr int KW2 = 8;    // over 2h
constexpr int KW3 = 12;   // over 3h

// one wave's partial tile -> part[wave][tile][lane][4]
__device__ __forceinline__ void put_part(float* part, int n_tiles, int wave, int tile, int lane, const Acc& c) {
    *reinterpret_cast<f32x4*>(part + ((size_t)(wave * n_tiles + tile) * 64 + lane) * 4) = c.hi + c.lo;
}

__device__ __forceinline__ void put_part1(float* part, int n_tiles, int wave, int tile, int lane, const f32x4 c) {
    *reinterpret_cast<f32x4*>(part + ((size_t)(wave * n_tiles + tile) * 64 + lane) * 4) = c;
}

// sums the waves' partial tiles in wave order into res[tile][16][RS]; all 256 threads; ends with a barrier
__device__ __forceinline__ void combine_parts(const float* part, float* res, int n_tiles, int n_waves_used) {
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX;
    for (int tile = wave; tile < n_tiles; tile += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(part + ((size_t)tile * 64 + lane) * 4);
        for (int w = 1; w < n_waves_used; ++w)
            v += *reinterpret_cast<const f32x4*>(part + ((size_t)(w * n_tiles + tile) * 64 + lane) * 4);
        float* r = res + (size_t)tile * 16 * RS + (4 * (lane >> 4)) * RS + (lane & 15);
        r[0] = v[0]; r[RS] = v[1]; r[2 * RS] = v[2]; r[3 * RS] = v[3];
    }
    __syncthreads();
}

// wave 0 waits for (up to) two counters; the whole workgroup learns the outcome. Returns false -> everybody leaves.
__device__ __forceinline__ bool group_wait(const unsigned* c0, unsigned want0, const unsigned* c1, unsigned want1,
                                           unsigned* error, int spin_limit, int* flag) {
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX;
    if (wave == 0) {
        bool ok = twog_wait_counter(c0, want0, error, spin_limit, lane);
        if (ok && c1) ok = twog_wait_counter(c1, want1, error, spin_limit, lane);
        if (lane == 0) *flag = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    return ok;
}

// every wave has drained its stores -> barrier -> one lane signals
__device__ __forceinline__ void group_signal(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) twog_jitter();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Diagnostic build only (-DTWOG_SP_STAMPS, tools/seg_persist_stamps.sh): thread 0 of the workgroup (direction 0, chunk 0,
// slice 0) of every role sums the wall-clock ticks (10 ns) between the phase boundaries of its steps and leaves the sums in
// the sync buffer at word 2048 + 16 role + k. The shipped library carries none of this.
#ifdef TWOG_SP_STAMPS
#define SP_STAMP(k)                                                                                         \
    do {                                                                                                    \
        if (threadIdx.x == 0 && G.dir == 0 && G.chunk == 0 && G.slice == 0) {                               \
            const long long now_ = wall_clock64();                                                          \
            atomicAdd(P.cnt + 2048 + 16 * G.role + (k), (unsigned)(now_ - stamp_));                         \
            stamp_ = now_;                                                                                  \
        }                                                                                                   \
    } while (0)
#define SP_STAMP_BEGIN() long long stamp_ = wall_clock64()
#else
#define SP_STAMP(k) do {} while (0)
#define SP_STAMP_BEGIN() do {} while (0)
#endif

struct Geo {   // what a workgroup knows about its place
    int dir, chunk, slice, role, b0, nb, RH, RO;
};

// layout of the saved attention weights of one (direction, step, clip): hh | oh | ho | oo (attn.hip)
__device__ __forceinline__ int att_hh(int H, int O) { return 0; }
__device__ __forceinline__ int att_oh(int H, int O) { return H * H; }
__device__ __forceinline__ int att_ho(int H, int O) { return H * H + H * O; }
__device__ __forceinline__ int att_oo(int H, int O) { return H * H + 2 * H * O; }

// masked softmax over the senders of one receiver (attn.hip, softmax_row): w[0..S) <- softmax over the valid senders,
// 0 elsewhere (no valid sender: all zeros -- the reference's NaN -> 0, models.py:1750-1753)
__device__ __forceinline__ void softmax_row(const float* score, int sstride, float* w, int S, int excluded, const float* smask) {
    float sc[MAXO];
    bool ok[MAXO];
    float m = -INFINITY;
#pragma unroll
    for (int s = 0; s < MAXO; ++s) {
        ok[s] = s < S && s != excluded && (!smask || smask[s < S ? s : 0] != 0.f);
        sc[s] = ok[s] ? score[(s < S ? s : 0) * sstride] : -INFINITY;
        m = fmaxf(m, sc[s]);
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < MAXO; ++s) {
        sc[s] = ok[s] ? expf(sc[s] - m) : 0.f;
        sum += sc[s];
    }
#pragma unroll
    for (int s = 0; s < MAXO; ++s)
        if (s < S) w[s] = ok[s] ? sc[s] / sum : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// P1 of receiver kind RK (0: humans receive -> role P1a; 1: objects receive -> P1b), one step.
// Tiles of `res`: [0, MH) sender MLP on human states (hh | ho), [MH, MH+MO) sender MLP on object states (oh | oo),
// then MK x 3 W_hh tiles of kind RK (row tile major), then the Gram tiles (i <= j over the MH + MO row tiles).
// ---------------------------------------------------------------------------------------------------------------------
template <int MH, int MO, int RK>
__device__ __forceinline__ bool p1_step(const SegArgs& P, const Geo& G, int s, float* part, float* res, float* sG, float* sW,
                                        const float* sMask, const float* sBias, int* flag, const WFrag (&Wr)[KW1][5]) {
    constexpr int MK = RK == 0 ? MH : MO, MT = MH + MO;
    constexpr int NPAIR = MT * (MT + 1) / 2;
    constexpr int T_SH = 0, T_SO = MH, T_G = MH + MO, T_GRAM = T_G + 3 * MK, NTILES = T_GRAM + NPAIR;
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX, i16 = lane & 15, g4 = lane >> 4;
    const int H = P.H, O = P.O, h = P.h, T = P.T, E_K = RK == 0 ? H : O;
    const int dir = G.dir, t = dir == 0 ? s : T - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
    const int nkb = h / 32;
    const int group = dir * P.n_chunks + G.chunk;
    unsigned* cnt = P.cnt + (size_t)group * 4 * CNT_STRIDE;
    const int ns = h / 16;

    SP_STAMP_BEGIN();
    Acc a_sh[MH], a_so[MO], a_g[MK][3];
    f32x4 a_gram[NPAIR];
#pragma unroll
    for (int i = 0; i < MH; ++i) acc_zero(a_sh[i]);
#pragma unroll
    for (int i = 0; i < MO; ++i) acc_zero(a_so[i]);
#pragma unroll
    for (int i = 0; i < MK; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc_zero(a_g[i][c]);
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) a_gram[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (s > 0) {
        // every column of h_{t-1}: all slices of P2h and P2o have published step s - 1
        if (!group_wait(cnt + 2 * CNT_STRIDE, (unsigned)s * ns, cnt + 3 * CNT_STRIDE, (unsigned)s * ns, P.error, P.spin_limit, flag))
            return false;
        SP_STAMP(0);   // waited for h_{t-1}
        const __amdgpu_buffer_rsrc_t rs_h = rsrc_of(P.hs[0]), rs_o = rsrc_of(P.hs[1]);
        // byte offset of (chunk row r of a kind, column 8 g4 of this direction) at time tp
        uint32_t off_h[MH], off_o[MO];
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            const int r = min(i * 16 + i16, G.RH - 1), b = G.b0 + r / H, e = r % H;
            off_h[i] = 4u * (uint32_t)((((int64_t)b * T + tp) * H + e) * (2 * h) + dir * h + 8 * g4);
        }
#pragma unroll
        for (int i = 0; i < MO; ++i) {
            const int r = min(i * 16 + i16, G.RO - 1), b = G.b0 + r / O, e = r % O;
            off_o[i] = 4u * (uint32_t)((((int64_t)b * T + tp) * O + e) * (2 * h) + dir * h + 8 * g4);
        }
        // the A fragments of this wave's k-blocks, two k-blocks in flight beside the two being multiplied; the weights are
        // in registers already
        typedef f32x4 Buf[2][MT][2];
        Buf b0, b1;
        auto load = [&](int g, Buf& buf) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int kb = wave + 4 * (2 * g + jj);
                if (kb < nkb) {
#pragma unroll
                    for (int i = 0; i < MH; ++i) { buf[jj][i][0] = ld_sc1(rs_h, off_h[i] + 128u * kb); buf[jj][i][1] = ld_sc1(rs_h, off_h[i] + 128u * kb + 16u); }
#pragma unroll
                    for (int i = 0; i < MO; ++i) { buf[jj][MH + i][0] = ld_sc1(rs_o, off_o[i] + 128u * kb); buf[jj][MH + i][1] = ld_sc1(rs_o, off_o[i] + 128u * kb + 16u); }
                }
            }
        };
        auto use = [&](int g, const Buf& buf) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * g + jj, kb = wave + 4 * j;
                if (kb < nkb) {
                    Planes A[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) A[i] = split8(buf[jj][i][0], buf[jj][i][1]);
                    {
                        const Planes B = split8(Wr[j][0].a, Wr[j][0].b);
#pragma unroll
                        for (int i = 0; i < MH; ++i) mac6(a_sh[i], A[i], B);
                    }
                    {
                        const Planes B = split8(Wr[j][1].a, Wr[j][1].b);
#pragma unroll
                        for (int i = 0; i < MO; ++i) mac6(a_so[i], A[MH + i], B);
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const Planes B = split8(Wr[j][2 + c].a, Wr[j][2 + c].b);
#pragma unroll
                        for (int i = 0; i < MK; ++i) mac6(a_g[i][c], A[(RK == 0 ? 0 : MH) + i], B);
                    }
                    // Gram tiles of the chunk's state rows: the B fragment of row tile j IS its A fragment. One accumulator:
                    // the scores go through a softmax, the split accumulators' last bits are not needed here
                    int p = 0;
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int i2 = i; i2 < MT; ++i2) {
                            if (P.pair_mask & (1u << (i * 8 + i2))) mac6_one(a_gram[p], A[i], A[i2]);
                            ++p;
                        }
                }
            }
        };
        static_assert(KW1 == 4, "two groups of two k-blocks");
        load(0, b0);
        load(1, b1);
        use(0, b0);
        use(1, b1);
    }
    SP_STAMP(1);   // loads + products
    // ---- partial tiles -> LDS, combined in wave order
#pragma unroll
    for (int i = 0; i < MH; ++i) put_part(part, NTILES, wave, T_SH + i, lane, a_sh[i]);
#pragma unroll
    for (int i = 0; i < MO; ++i) put_part(part, NTILES, wave, T_SO + i, lane, a_so[i]);
#pragma unroll
    for (int i = 0; i < MK; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) put_part(part, NTILES, wave, T_G + i * 3 + c, lane, a_g[i][c]);
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) put_part1(part, NTILES, wave, T_GRAM + p, lane, a_gram[p]);
    combine_parts(part, res, NTILES, min(4, nkb));
    SP_STAMP(2);   // combine

    const int tid = threadIdx.x, q = tid & 3, ur = tid >> 2;   // (row of a 64-row pass, quad of 4 units)
    const int col = G.slice * 16 + 4 * q;
    // ---- sender messages: relu(. + bias), kept in `res` for the weighted sums and saved for the backward pass
    {
        const int blk = RK == 0 ? 0 : 1;   // column block inside msrc_h (hh | ho) and msrc_o (oh | oo)
        if (ur < MH * 16) {
            float* r = res + (size_t)(T_SH + ur / 16) * 16 * RS + (ur % 16) * RS + 4 * q;
            f32x4 v = *reinterpret_cast<f32x4*>(r);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k] + bias[k], 0.f);
            *reinterpret_cast<f32x4*>(r) = v;
            if (ur < G.RH) {
                const int b = G.b0 + ur / H, e = ur % H;
                float* dst = P.msrc[0] + ((((int64_t)dir * P.bs + b) * T + t) * H + e) * (2 * h) + blk * h + col;
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        for (int x = ur; x < MO * 16; x += 64) {
            float* r = res + (size_t)(T_SO + x / 16) * 16 * RS + (x % 16) * RS + 4 * q;
            f32x4 v = *reinterpret_cast<f32x4*>(r);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(sBias + 16 + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k] + bias[k], 0.f);
            *reinterpret_cast<f32x4*>(r) = v;
            if (x < G.RO) {
                const int b = G.b0 + x / O, e = x % O;
                float* dst = P.msrc[1] + ((((int64_t)dir * P.bs + b) * T + t) * O + e) * (2 * h) + blk * h + col;
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    }
    // ---- scores of the pairs this role's receivers need: sG[clip][receiver of kind RK][sender in E] = scale <h_r, h_s>
    const int E = H + O;
    {
        const int n = G.nb * E_K * E;
        for (int x = tid; x < n; x += 256) {
            const int bl = x / (E_K * E), rem = x - bl * (E_K * E), r = rem / E, sdr = rem - r * E;
            // unified rows: humans at bl * H + e, objects at 16 MH + bl * O + e
            const int ra = RK == 0 ? bl * H + r : 16 * MH + bl * O + r;
            const int rb = sdr < H ? bl * H + sdr : 16 * MH + bl * O + (sdr - H);
            int ta = ra / 16, tb = rb / 16, ia = ra % 16, ib = rb % 16;
            if (ta > tb) { const int k1 = ta; ta = tb; tb = k1; const int k2 = ia; ia = ib; ib = k2; }
            const int p = ta * MT - ta * (ta - 1) / 2 + (tb - ta);   // index of (ta, tb), ta <= tb, in the unrolled order
            sG[x] = P.scale * res[(size_t)(T_GRAM + p) * 16 * RS + ia * RS + ib];
        }
    }
    __syncthreads();
    // ---- the two masked softmaxes of this role: one thread per (clip, relation, receiver)
    const int natt = H * H + 2 * H * O + O * O;
    {
        const int n = G.nb * 2 * E_K;
        for (int x = tid; x < n; x += 256) {
            const int bl = x / (2 * E_K), rem = x - bl * 2 * E_K, rel = rem / E_K, r = rem - rel * E_K;
            const float* sc = sG + (size_t)(bl * E_K + r) * E + (rel == 0 ? 0 : H);
            float* w;
            int S, excl = -1;
            const float* mk = nullptr;
            if (RK == 0) {
                if (rel == 0) { w = sW + bl * natt + att_hh(H, O) + r * H; S = H; excl = r; }
                else { w = sW + bl * natt + att_oh(H, O) + r * O; S = O; mk = sMask + bl * O; }
            } else {
                if (rel == 0) { w = sW + bl * natt + att_ho(H, O) + r * H; S = H; }
                else { w = sW + bl * natt + att_oo(H, O) + r * O; S = O; excl = r; mk = sMask + bl * O; }
            }
            softmax_row(sc, 1, w, S, excl, mk);
        }
    }
    __syncthreads();
    SP_STAMP(3);   // relu, scores, softmax
    if (G.slice == 0) {   // one workgroup per (group, receiver kind) saves its half of the weights for the backward pass
        const int o0 = RK == 0 ? 0 : att_ho(H, O), o1 = RK == 0 ? att_ho(H, O) : natt;
        const int n = G.nb * (o1 - o0);
        for (int x = tid; x < n; x += 256) {
            const int bl = x / (o1 - o0), i = o0 + x - bl * (o1 - o0);
            P.att[(((int64_t)dir * T + t) * P.bs + G.b0 + bl) * natt + i] = sW[bl * natt + i];
        }
    }
    // ---- aggregated messages of the receivers (two blocks) and W_hh h_prev + b_hh: write-through, then the signal
    {
        const __amdgpu_buffer_rsrc_t rs_mg = rsrc_of(P.mg[RK]), rs_gh = rsrc_of(P.gh[RK]);
        const int R_K = RK == 0 ? G.RH : G.RO;
        for (int x = ur; x < MK * 16; x += 64) {
            if (x >= R_K) continue;
            const int bl = x / E_K, e = x - bl * E_K, b = G.b0 + bl;
            // block 0: senders humans (hh / ho); block 1: senders objects (oh / oo)
            f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};
            const float* w0 = sW + bl * natt + (RK == 0 ? att_hh(H, O) + e * H : att_ho(H, O) + e * H);
            const float* w1 = sW + bl * natt + (RK == 0 ? att_oh(H, O) + e * O : att_oo(H, O) + e * O);
            for (int sd = 0; sd < H; ++sd) {
                const int r = bl * H + sd;
                const f32x4 v = *reinterpret_cast<const f32x4*>(res + (size_t)(T_SH + r / 16) * 16 * RS + (r % 16) * RS + 4 * q);
                const float w = w0[sd];
#pragma unroll
                for (int k = 0; k < 4; ++k) m0[k] = fmaf(w, v[k], m0[k]);
            }
            for (int sd = 0; sd < O; ++sd) {
                const int r = bl * O + sd;
                const f32x4 v = *reinterpret_cast<const f32x4*>(res + (size_t)(T_SO + r / 16) * 16 * RS + (r % 16) * RS + 4 * q);
                const float w = w1[sd];
#pragma unroll
                for (int k = 0; k < 4; ++k) m1[k] = fmaf(w, v[k], m1[k]);
            }
            const uint32_t o = 4u * (uint32_t)(((((int64_t)dir * P.bs + b) * T + t) * E_K + e) * (2 * h) + col);
            st_sc1(rs_mg, o, m0);
            st_sc1(rs_mg, o + 4u * (uint32_t)h, m1);
            // W_hh h_prev + b_hh of this row, three gates
            const int64_t grow = (int64_t)dir * P.bs * E_K + (int64_t)b * E_K + e;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x4 v = *reinterpret_cast<const f32x4*>(res + (size_t)(T_G + (x / 16) * 3 + c) * 16 * RS + (x % 16) * RS + 4 * q);
                v += *reinterpret_cast<const f32x4*>(sBias + 32 + 16 * c + 4 * q);
                st_sc1(rs_gh, 4u * (uint32_t)(grow * 3 * h + c * h + col), v);
            }
        }
    }
    SP_STAMP(4);   // weighted sums, stores issued
    group_signal(cnt + RK * CNT_STRIDE);
    SP_STAMP(5);   // drained + signalled
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// P2 of kind K (0 humans, 1 objects), one step: W_ih[:, messages] on the complete aggregated-message rows, the gates, the
// blend with the hard gate, the slice's 16 columns of h_t. h_own: this thread's own (row, 4 units) of the previous step.
// ---------------------------------------------------------------------------------------------------------------------
template <int MK, int K>
__device__ __forceinline__ bool p2_step(const SegArgs& P, const Geo& G, int s, float* part, float* res, int* flag,
                                        f32x4 (&h_own)[(MK * 16 + 63) / 64], const WFrag (&Wr)[KW2][3]) {
    constexpr int NTILES = 3 * MK, NPASS = (MK * 16 + 63) / 64;
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX, i16 = lane & 15, g4 = lane >> 4;
    const int h = P.h, T = P.T, E_K = K == 0 ? P.H : P.O, R_K = K == 0 ? G.RH : G.RO;
    const int dir = G.dir, t = dir == 0 ? s : T - 1 - s;
    const int ns = h / 16, nkb = 2 * h / 32;
    const int group = dir * P.n_chunks + G.chunk;
    unsigned* cnt = P.cnt + (size_t)group * 4 * CNT_STRIDE;
    const int tid = threadIdx.x, q = tid & 3, ur = tid >> 2;
    const int col = G.slice * 16 + 4 * q;

    SP_STAMP_BEGIN();
    // what does not depend on the chain: the frame part of W_ih x + b_ih and the hard gate of this thread's rows
    f32x4 gi[NPASS][3];
    float uu[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int x = min(ps * 64 + ur, R_K - 1), bl = x / E_K, e = x - bl * E_K, b = G.b0 + bl;
        const float* p = P.gi[K] + ((((int64_t)b * T + t) * E_K + e) * 6 + dir * 3) * h + col;
#pragma unroll
        for (int c = 0; c < 3; ++c) gi[ps][c] = *reinterpret_cast<const f32x4*>(p + c * h);
        uu[ps] = P.u[K][((int64_t)b * T + t) * E_K + e];
    }
    // the aggregated messages (every column) and W_hh h_prev of this kind: P1 of receiver kind K, all slices
    if (!group_wait(cnt + K * CNT_STRIDE, (unsigned)(s + 1) * ns, nullptr, 0u, P.error, P.spin_limit, flag)) return false;
    SP_STAMP(0);   // waited for the messages
    const __amdgpu_buffer_rsrc_t rs_mg = rsrc_of(P.mg[K]), rs_gh = rsrc_of(P.gh[K]);
    f32x4 acc[MK][3];
#pragma unroll
    for (int i = 0; i < MK; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t off[MK];
#pragma unroll
    for (int i = 0; i < MK; ++i) {
        const int r = min(i * 16 + i16, R_K - 1), bl = r / E_K, e = r - bl * E_K, b = G.b0 + bl;
        off[i] = 4u * (uint32_t)(((((int64_t)dir * P.bs + b) * T + t) * E_K + e) * (2 * h) + 8 * g4);
    }
    // 3 x bf16 products, one accumulator per tile (round 6): the compiler keeps the weight slice as bf16 PLANES across the
    // time loop (288 registers instead of 192 of raw fp32: the split is hoisted, nothing is split per step), and with three
    // k-blocks of messages in flight instead of four the kernel fits its 512 registers without scratch: 27.2 -> 26.1 us per
    // step at 8 clips x h 512, 18.2 -> 17.2 at one clip (profiles/r06_seg_persist_p2_x3.txt). Round 5 had measured this loop
    // faster too but got wrong results from it and blamed its scratch use -- the cause was the vector-register wave index
    // (see the top of this file). -DTWOG_SP_P2_F32 rebuilds the fp32-pipe loop (v_mfma_f32_16x16x4_f32, exact fp32 products).
#ifndef TWOG_SP_P2_F32
#ifndef TWOG_SP_P2_CH
#define TWOG_SP_P2_CH 3
#endif
    k_stream<KW2, MK, TWOG_SP_P2_CH>(rs_mg, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
        Planes Ap[MK];
#pragma unroll
        for (int i = 0; i < MK; ++i) Ap[i] = split8(A[i][0], A[i][1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const Planes B = split8(Wr[j][c].a, Wr[j][c].b);
#pragma unroll
            for (int i = 0; i < MK; ++i) mac6_one(acc[i][c], Ap[i], B);
        }
    });
#else
    k_stream<KW2, MK>(rs_mg, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < MK; ++i) mac_f32(acc[i][c], A[i][0], A[i][1], Wr[j][c]);
    });
#endif
    f32x4 gh[NPASS][3];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int x = min(ps * 64 + ur, R_K - 1), bl = x / E_K, e = x - bl * E_K, b = G.b0 + bl;
        const int64_t grow = (int64_t)dir * P.bs * E_K + (int64_t)b * E_K + e;
#pragma unroll
        for (int c = 0; c < 3; ++c) gh[ps][c] = ld_sc1(rs_gh, 4u * (uint32_t)(grow * 3 * h + c * h + col));
    }

    SP_STAMP(1);   // loads + products
#pragma unroll
    for (int i = 0; i < MK; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) put_part1(part, NTILES, wave, i * 3 + c, lane, acc[i][c]);
    combine_parts(part, res, NTILES, min(4, nkb));
    SP_STAMP(2);   // combine
    // ---- gates (gru.hip, gru_step_fwd_kernel: same arithmetic), states write-through, then what the backward pass reads
    const __amdgpu_buffer_rsrc_t rs_hs = rsrc_of(P.hs[K]);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int x = ps * 64 + ur;
        if (x < R_K) {
            const int bl = x / E_K, e = x - bl * E_K, b = G.b0 + bl;
            f32x4 gim[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                gim[c] = *reinterpret_cast<const f32x4*>(res + (size_t)((x / 16) * 3 + c) * 16 * RS + (x % 16) * RS + 4 * q);
            f32x4 rg, zz, nn, hnew;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ir = gi[ps][0][k] + gim[0][k], iz = gi[ps][1][k] + gim[1][k], in_ = gi[ps][2][k] + gim[2][k];
                const float hr = gh[ps][0][k], hz = gh[ps][1][k], hn = gh[ps][2][k];
                rg[k] = 1.0f / (1.0f + expf(-(ir + hr)));
                zz[k] = 1.0f / (1.0f + expf(-(iz + hz)));
                nn[k] = tanhf(in_ + rg[k] * hn);
                const float h0 = h_own[ps][k];
                const float gnew = (1.0f - zz[k]) * nn[k] + zz[k] * h0;
                hnew[k] = uu[ps] * gnew + (1.0f - uu[ps]) * h0;
            }
            h_own[ps] = hnew;
            st_sc1(rs_hs, 4u * (uint32_t)((((int64_t)b * T + t) * E_K + e) * (2 * h) + dir * h + col), hnew);
            float* sv = P.save[K] + ((((int64_t)dir * P.bs + b) * T + t) * E_K + e) * (4 * h) + col;
            *reinterpret_cast<f32x4*>(sv) = rg;
            *reinterpret_cast<f32x4*>(sv + h) = zz;
            *reinterpret_cast<f32x4*>(sv + 2 * h) = nn;
            *reinterpret_cast<f32x4*>(sv + 3 * h) = gh[ps][2];
        }
    }
    SP_STAMP(4);   // gates, stores issued
    group_signal(cnt + (2 + K) * CNT_STRIDE);
    SP_STAMP(5);   // drained + signalled
    return true;
}

// One function per role, inlined into the kernel (as a real call the kernel's argument block would be passed through
// scratch memory and re-read from there every step). Register note: the fragment buffers of the reduction loops are NAMED
// arrays -- indexing them with a run-time buffer number put them (250-500 registers per lane) into scratch memory.
struct FwdLds { float *part, *res, *sG, *sW, *sMask, *sBias; int* flag; };

// The weights of a workgroup are the same every step: its slice is loaded ONCE, as raw fp32 MFMA fragments, into the
// registers of the wave that multiplies it (k-block kb = wave + 4 j is fragment j) -- 160 / 192 VGPRs at h = 512 -- and split
// into the bf16 planes at every use. Nothing but states and messages moves per step.
template <int MH, int MO, int RK>
__device__ __forceinline__ void role_p1(const SegArgs& P, const Geo& G, const FwdLds& M) {
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX;
    const int nkb = P.h / 32;
    WFrag Wr[KW1][5];
#pragma unroll
    for (int j = 0; j < KW1; ++j) {
        const int kb = wave + 4 * j;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            if (kb < nkb) {
                const float* w = n == 0 ? P.w_s[RK == 0 ? 0 : 1] : n == 1 ? P.w_s[RK == 0 ? 2 : 3] : P.w_hh[RK][G.dir];
                Wr[j][n] = load_w_raw(w, P.h, (n < 2 ? 0 : (n - 2) * P.h) + G.slice * 16, kb * 32, lane);
            } else {
                Wr[j][n] = wfrag_zero();
            }
        }
    }
    // the slice's biases, once: [0, 16) sender MLP on human states, [16, 32) on object states, [32, 80) b_hh r | z | n
    if (threadIdx.x < 80) {
        const int i = threadIdx.x, c16 = G.slice * 16 + (i & 15);
        const float* b = i < 16 ? P.b_s[RK == 0 ? 0 : 1] : i < 32 ? P.b_s[RK == 0 ? 2 : 3] : P.b_hh[RK][G.dir];
        M.sBias[i] = b ? b[(i < 32 ? 0 : ((i - 32) / 16) * P.h) + c16] : 0.f;
    }
    __syncthreads();
    for (int s = 0; s < P.T; ++s)
        if (!p1_step<MH, MO, RK>(P, G, s, M.part, M.res, M.sG, M.sW, M.sMask, M.sBias, M.flag, Wr)) return;
}

template <int MK, int K>
__device__ __forceinline__ void role_p2(const SegArgs& P, const Geo& G, const FwdLds& M) {
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX;
    const int nkb = 2 * P.h / 32;
    WFrag Wr[KW2][3];
#pragma unroll
    for (int j = 0; j < KW2; ++j) {
        const int kb = wave + 4 * j;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Wr[j][c] = kb < nkb ? load_w_raw(P.w_ihm[K][G.dir], P.ld_ih[K], c * P.h + G.slice * 16, kb * 32, lane) : wfrag_zero();
    }
    f32x4 h_own[(MK * 16 + 63) / 64];
#pragma unroll
    for (int i = 0; i < (MK * 16 + 63) / 64; ++i) h_own[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < P.T; ++s)
        if (!p2_step<MK, K>(P, G, s, M.part, M.res, M.flag, h_own, Wr)) return;
}

template <int MH, int MO>
__global__ __launch_bounds__(256, 1) void seg_persist_fwd_kernel(const SegArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = MH + MO, NPAIR = MT * (MT + 1) / 2;
    constexpr int NT_MAX = MH + MO + 3 * (MH > MO ? MH : MO) + NPAIR;
    const int ns = P.h / 16;
    const int W = (int)gridDim.x;
    // consecutive linear indices on one XCD (blocks are dealt round-robin over the 8 XCDs): speed only
    const int L = ((int)blockIdx.x % 8) * (W / 8) + (int)blockIdx.x / 8;
    const int group = L / (4 * ns), within = L - group * 4 * ns;
    Geo G;
    G.dir = group / P.n_chunks; G.chunk = group - G.dir * P.n_chunks;
    G.slice = within / 4; G.role = within - G.slice * 4;
    G.b0 = G.chunk * P.cpc; G.nb = min(P.cpc, P.bs - G.b0);
    G.RH = G.nb * P.H; G.RO = G.nb * P.O;
    float* part = reinterpret_cast<float*>(smem);                       // [4][NT_MAX][64][4]
    float* res = part + 4 * NT_MAX * 256;                               // [NT_MAX][16][RS]
    float* sG = res + NT_MAX * 16 * RS;                                 // [cpc][E_K][E]
    const int E = P.H + P.O, natt = P.H * P.H + 2 * P.H * P.O + P.O * P.O;
    float* sW = sG + P.cpc * (P.H > P.O ? P.H : P.O) * E;               // [cpc][natt]
    float* sMask = sW + P.cpc * natt;                                   // [cpc][O]
    float* sBias = sMask + ((P.cpc * P.O + 3) & ~3);                    // [80]
    int* flag = reinterpret_cast<int*>(sBias + 80);
    for (int i = threadIdx.x; i < G.nb * P.O; i += 256) sMask[i] = P.mask ? P.mask[(int64_t)G.b0 * P.O + i] : 1.f;
    __syncthreads();
    FwdLds M{part, res, sG, sW, sMask, sBias, flag};
    if (G.role == 0) role_p1<MH, MO, 0>(P, G, M);
    else if (G.role == 1) role_p1<MH, MO, 1>(P, G, M);
    else if (G.role == 2) role_p2<MH, 0>(P, G, M);
    else role_p2<MO, 1>(P, G, M);
}

// =====================================================================================================================
// Backward through time as ONE persistent launch. Reference: autograd through the segment loop (vhoi/models.py:785-880);
// same outputs as twog_segrnn_bwd (segrnn.hip): d_gi, d_gh (gradients wrt the two GRUCell projections of every step),
// d_pre (wrt the pre-ReLU sender-MLP activations), d_u (+= gradient wrt the hard gates).
//
// Per (direction, clip chunk) and slice of 16 columns four roles again, two hand-offs per step:
//   Q2h / Q2o  own 16 hidden units of the rows of one kind. Keep the carried state gradient of their units in registers.
//        Per step: carry <- direct part + W_hh part + sender-MLP part + attention-score part; gate backward of the own
//        units -> d_gi, d_gh columns (write-through) -> SIGNAL X1; then, beside the Q1 workgroups' work, the W_hh part of
//        the NEXT carry: the complete d_gh rows (all slices of the kind: X1) times W_hh[:, own units].
//   Q1h / Q1o  own 16 columns of both message blocks a kind RECEIVES. Per step (after X1): d_mg = complete d_gi rows
//        times W_ih[:, message columns]; from it, with the SAVED attention weights (no softmax backward needed for
//        this), the gradient of the sender messages of its two relations -> ReLU mask -> d_pre columns (write-through;
//        they are also an output), and the slice's share of the score gradients dL/dw[r][s] = <d_mg[r], msg[s]>
//        (16 of the h columns) -> SIGNAL X2.
//   Q2 (after X2 of both Q1 kinds): sender-MLP part = complete d_pre rows times W_s[:, own units]; dL/dw summed over
//        the slices in slice order, softmax backward, the attention-score part of the carry from the saved states.
// Everything else as in the forward launch: exact 3 x bf16 products, k-blocks dealt to the four waves and combined in
// wave order, write-through hand-offs, bounded waits that fail soft. d_u: every Q2 workgroup parks the sum over its 16
// units per (step, row); a second launch adds the h / 16 partials in slice order (bit-reproducible).
// =====================================================================================================================
struct SegBwdArgs {
    int bs, T, H, O, h;
    int cpc, n_chunks;
    int dw_pad;                  // floats of one (direction, step, slice, receiver kind, chunk) block of dL/dw shares
    float scale;
    int spin_limit;
    const float* u[2];
    const float* w_hh[2][2];     // [kind][dir] [3h][h]
    const float* w_ihm[2][2];
    int64_t ld_ih[2];
    const float* w_sp[2];        // packed sender MLPs on human / object states: [2h][h]
    const float* hs[2];          // forward states [bs][T][E][2h]
    const float* save[2];        // [2][bs][T][E][4h]
    const float* msrc[2];        // [2][bs][T][E][2h]
    const float* att;            // [2][T][bs][natt]
    const float* d_hs[2];        // [bs][T][E][2h]
    float* d_gi[2];              // [bs][T][E][6h]
    float* d_gh[2];
    float* d_pre[2];             // [2][bs][T][E][2h]
    float* dwpart;               // [2][T][ns][2][n_chunks][dw_pad]
    float* du_part[2];           // [2][T][ns][bs*E]
    unsigned* cnt;
    unsigned* error;
};

// k-major weight fragment: lane l holds W[k0 + 8 (l >> 4) + j][col0 + (l & 15)], j = 0..7 (W row-major, row stride ld)
__device__ __forceinline__ Planes load_wk(const float* w, int64_t ld, int k0, int col0, int lane) {
    const float* p = w + (int64_t)(k0 + 8 * (lane >> 4)) * ld + col0 + (lane & 15);
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = p[(int64_t)j * ld]; b[j] = p[(int64_t)(j + 4) * ld]; }
    return split8(a, b);
}

// ---- Q1 of receiver kind RK: one step
template <int MH, int MO, int RK, bool X3Q1>
__device__ __forceinline__ bool q1_step(const SegBwdArgs& P, const Geo& G, int s, float* part, float* res, float* msT,
                                        float* sW, float* sDW, int* flag, const WFrag (&Wr)[KW3][2]) {
    constexpr int MK = RK == 0 ? MH : MO, NTILES = 2 * MK;
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX, i16 = lane & 15, g4 = lane >> 4;
    const int H = P.H, O = P.O, h = P.h, T = P.T, E_K = RK == 0 ? H : O, R_K = RK == 0 ? G.RH : G.RO;
    const int dir = G.dir, t = dir == 0 ? s : T - 1 - s;
    const int ns = h / 16, nkb = 3 * h / 32;
    const int group = dir * P.n_chunks + G.chunk;
    unsigned* cnt = P.cnt + (size_t)group * 4 * CNT_STRIDE;
    const int tid = threadIdx.x, q = tid & 3, ur = tid >> 2;
    const int col = G.slice * 16 + 4 * q;
    const int natt = H * H + 2 * H * O + O * O;
    SP_STAMP_BEGIN();
    // chain-independent inputs first: the saved attention weights of (dir, t) and the sender-message columns of this slice
    for (int x = tid; x < G.nb * natt; x += 256) {
        const int bl = x / natt, i = x - bl * natt;
        sW[x] = P.att[(((int64_t)dir * T + t) * P.bs + G.b0 + bl) * natt + i];
    }
    // msT: [RH + RO][RS] sender messages (post-ReLU) of the two relations this kind receives, 16 columns
    const int sblk = RK == 0 ? 0 : 1;   // block inside msrc_h (hh | ho) and msrc_o (oh | oo)
    for (int x = ur; x < G.RH + G.RO; x += 64) {
        const bool hum = x < G.RH;
        const int r = hum ? x : x - G.RH, Es = hum ? H : O, bl = r / Es, e = r - bl * Es, b = G.b0 + bl;
        const float* src = P.msrc[hum ? 0 : 1] + ((((int64_t)dir * P.bs + b) * T + t) * Es + e) * (2 * h) + sblk * h + col;
        *reinterpret_cast<f32x4*>(msT + (size_t)x * RS + 4 * q) = *reinterpret_cast<const f32x4*>(src);
    }
    // every column of d_gi of this kind at step s
    if (!group_wait(cnt + (2 + RK) * CNT_STRIDE, (unsigned)(T - s) * ns, nullptr, 0u, P.error, P.spin_limit, flag)) return false;
    SP_STAMP(0);   // waited for d_gi
    const __amdgpu_buffer_rsrc_t rs_gi = rsrc_of(P.d_gi[RK]);
    f32x4 acc[MK][2];
#pragma unroll
    for (int i = 0; i < MK; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
    uint32_t off[MK];
#pragma unroll
    for (int i = 0; i < MK; ++i) {
        const int r = min(i * 16 + i16, R_K - 1), bl = r / E_K, e = r - bl * E_K, b = G.b0 + bl;
        off[i] = 4u * (uint32_t)((((int64_t)b * T + t) * E_K + e) * (6 * h) + dir * 3 * h + 8 * g4);
    }
    // d_mg = d_gi W_ih[:, messages]: 3 x bf16 products with one accumulator per tile where the reduction is long (X3Q1, the host
    // picks it for h >= 256: 28.1 -> 26.2 us per step at 8 clips x h 512; at h = 64 -- one or two k-blocks per wave -- the
    // fp32 pipe without the split is faster: 10.3 against 10.7 us; profiles/r06_seg_persist_p2_x3.txt), else exact fp32 products
    if constexpr (X3Q1) {
        k_stream<KW3, MK, 3>(rs_gi, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
            Planes Ap[MK];
#pragma unroll
            for (int i = 0; i < MK; ++i) Ap[i] = split8(A[i][0], A[i][1]);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const Planes B = split8(Wr[j][blk].a, Wr[j][blk].b);
#pragma unroll
                for (int i = 0; i < MK; ++i) mac6_one(acc[i][blk], Ap[i], B);
            }
        });
    } else {
        k_stream<KW3, MK>(rs_gi, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int i = 0; i < MK; ++i) mac_f32(acc[i][blk], A[i][0], A[i][1], Wr[j][blk]);
        });
    }
#pragma unroll
    for (int i = 0; i < MK; ++i) { put_part1(part, NTILES, wave, i * 2, lane, acc[i][0]); put_part1(part, NTILES, wave, i * 2 + 1, lane, acc[i][1]); }
    SP_STAMP(1);   // loads + products
    combine_parts(part, res, NTILES, min(4, nkb));
    SP_STAMP(2);   // combine
    // res tile (i * 2 + blk): d_mg[receiver rows of kind RK][message block blk][16 columns]
    // ---- gradient of the sender messages (saved weights), ReLU mask, d_pre columns: write-through
    {
        const __amdgpu_buffer_rsrc_t rs_ph = rsrc_of(P.d_pre[0]), rs_po = rsrc_of(P.d_pre[1]);
        for (int x = ur; x < G.RH + G.RO; x += 64) {
            const bool hum = x < G.RH;
            const int r = hum ? x : x - G.RH, Es = hum ? H : O, bl = r / Es, sd = r - bl * Es, b = G.b0 + bl;
            const int blk = hum ? 0 : 1;
            // weights of the receivers of kind RK towards sender sd: hh / ho (senders humans), oh / oo (senders objects)
            const float* w = sW + bl * natt + (RK == 0 ? (hum ? att_hh(H, O) : att_oh(H, O)) : (hum ? att_ho(H, O) : att_oo(H, O))) + sd;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            for (int rc = 0; rc < E_K; ++rc) {
                const int rr = bl * E_K + rc;
                const f32x4 d = *reinterpret_cast<const f32x4*>(res + (size_t)((rr / 16) * 2 + blk) * 16 * RS + (rr % 16) * RS + 4 * q);
                const float wv = w[rc * Es];
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = fmaf(wv, d[k], g[k]);
            }
            const f32x4 m = *reinterpret_cast<const f32x4*>(msT + (size_t)x * RS + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!(m[k] > 0.f)) g[k] = 0.f;
            const uint32_t o = 4u * (uint32_t)(((((int64_t)dir * P.bs + b) * T + t) * Es + sd) * (2 * h) + sblk * h + col);
            st_sc1(hum ? rs_ph : rs_po, o, g);
        }
    }
    SP_STAMP(3);   // d_pre
    // ---- this slice's share of dL/dw[r][s] = <d_mg[r], msg[s]> (16 of the h columns), both relations; not needed at the
    // chain start (no previous state, no score gradient)
    const int nr = RK == 0 ? H * H + H * O : H * O + O * O;
    if (s > 0) {
        for (int x = tid; x < P.dw_pad; x += 256) {
            float v = 0.f;
            if (x < G.nb * nr) {
                const int bl = x / nr, i = x - bl * nr;
                int rc, sd, blk;
                bool diag;
                const int nfirst = E_K * H;   // slots of the relation with human senders (hh / ho)
                if (i < nfirst) { rc = i / H; sd = i - rc * H; blk = 0; diag = RK == 0 && rc == sd; }
                else { const int j = i - nfirst; rc = j / O; sd = j - rc * O; blk = 1; diag = RK == 1 && rc == sd; }
                if (!diag) {
                    const int rr = bl * E_K + rc;
                    const float* d = res + (size_t)((rr / 16) * 2 + blk) * 16 * RS + (rr % 16) * RS;
                    const float* m = msT + (size_t)(blk == 0 ? bl * H + sd : G.RH + bl * O + sd) * RS;
#pragma unroll
                    for (int k = 0; k < 16; ++k) v = fmaf(d[k], m[k], v);
                }
            }
            sDW[x] = v;
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rs_dw = rsrc_of(P.dwpart);
        const int64_t blk_off = ((((int64_t)dir * T + t) * ns + G.slice) * 2 + RK) * P.n_chunks + G.chunk;
        for (int x = tid; x < P.dw_pad / 4; x += 256)
            st_sc1(rs_dw, 4u * (uint32_t)(blk_off * P.dw_pad + 4 * x), *reinterpret_cast<const f32x4*>(sDW + 4 * x));
    }
    SP_STAMP(4);   // dL/dw shares
    group_signal(cnt + RK * CNT_STRIDE);
    SP_STAMP(5);   // drain + signal
    return true;
}

// ---- Q2 of kind K: one step. carry / c_hh: this thread's (row, 4 units) of the carried gradient and of its W_hh part.
template <int MK, int K>
__device__ __forceinline__ bool q2_step(const SegBwdArgs& P, const Geo& G, int s, float* part, float* res, float* fT, float* sW,
                                        float* sDW, float* sC, int* flag, f32x4 (&direct)[(MK * 16 + 63) / 64],
                                        f32x4 (&c_hh)[(MK * 16 + 63) / 64], const WFrag (&We)[KW2], const WFrag (&Wh)[KW3]) {
    constexpr int NPASS = (MK * 16 + 63) / 64;
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX, i16 = lane & 15, g4 = lane >> 4;
    const int H = P.H, O = P.O, E = H + O, h = P.h, T = P.T, E_K = K == 0 ? H : O, R_K = K == 0 ? G.RH : G.RO;
    const int dir = G.dir, t = dir == 0 ? s : T - 1 - s, tp = dir == 0 ? t - 1 : t + 1;
    const int tn = dir == 0 ? t + 1 : t - 1;   // the time of chain step s + 1 (processed before this one)
    const int ns = h / 16;
    const int group = dir * P.n_chunks + G.chunk;
    unsigned* cnt = P.cnt + (size_t)group * 4 * CNT_STRIDE;
    const int tid = threadIdx.x, q = tid & 3, ur = tid >> 2;
    const int col = G.slice * 16 + 4 * q;
    const int natt = H * H + 2 * H * O + O * O;
    const bool first = s == 0, last = s == T - 1;
    SP_STAMP_BEGIN();

    // chain-independent inputs of the gate backward, requested before any wait
    f32x4 dout[NPASS], sr[NPASS], sz[NPASS], sn[NPASS], shn[NPASS], h0[NPASS];
    float uu[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int x = min(ps * 64 + ur, R_K - 1), bl = x / E_K, e = x - bl * E_K, b = G.b0 + bl;
        const int64_t row = ((int64_t)b * T + t) * E_K + e;
        dout[ps] = *reinterpret_cast<const f32x4*>(P.d_hs[K] + row * (2 * h) + dir * h + col);
        const float* sv = P.save[K] + (((int64_t)dir * P.bs + b) * T * E_K + (int64_t)t * E_K + e) * (4 * h) + col;
        sr[ps] = *reinterpret_cast<const f32x4*>(sv);
        sz[ps] = *reinterpret_cast<const f32x4*>(sv + h);
        sn[ps] = *reinterpret_cast<const f32x4*>(sv + 2 * h);
        shn[ps] = *reinterpret_cast<const f32x4*>(sv + 3 * h);
        h0[ps] = first ? f32x4{0.f, 0.f, 0.f, 0.f}
                       : *reinterpret_cast<const f32x4*>(P.hs[K] + (((int64_t)b * T + tp) * E_K + e) * (2 * h) + dir * h + col);
        uu[ps] = P.u[K][row];
    }
    f32x4 carry[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) carry[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!last) {
        // saved attention weights of chain step s + 1 (time tn): chain-independent
        for (int x = tid; x < G.nb * natt; x += 256) {
            const int bl = x / natt, i = x - bl * natt;
            sW[x] = P.att[(((int64_t)dir * T + tn) * P.bs + G.b0 + bl) * natt + i];
        }
        // fT: [RH + RO][RS] the features of step s + 1 = the forward states at time t, this slice's 16 columns (chain-independent)
        for (int x = ur; x < G.RH + G.RO; x += 64) {
            const bool hum = x < G.RH;
            const int r = hum ? x : x - G.RH, Es = hum ? H : O, bl = r / Es, e = r - bl * Es, b = G.b0 + bl;
            *reinterpret_cast<f32x4*>(fT + (size_t)x * RS + 4 * q) =
                *reinterpret_cast<const f32x4*>(P.hs[hum ? 0 : 1] + (((int64_t)b * T + t) * Es + e) * (2 * h) + dir * h + col);
        }
        // d_pre columns and dL/dw shares of step s + 1: all slices of Q1h and Q1o
        if (!group_wait(cnt + 0 * CNT_STRIDE, (unsigned)(T - 1 - s) * ns, cnt + 1 * CNT_STRIDE, (unsigned)(T - 1 - s) * ns, P.error,
                        P.spin_limit, flag))
            return false;
        SP_STAMP(0);   // waited for d_pre / dL/dw shares
        // ---- dL/dw shares of step s + 1 (all four relations, every slice): requested first, ALL at once, 16 bytes per load
        // -- thread (c, half) takes float4 column c of the two receiver kinds' blocks for the slices of its half, 16 loads in
        // flight (one scalar after the other they cost a round trip each: 12 us of a 38 us step; as 13 000 scalar loads per
        // workgroup they saturated the L2 request rate: 8 us) -- and added after the product below: slices in order within a
        // half, then half 0 + half 1
        constexpr int NSH = 16;
        const __amdgpu_buffer_rsrc_t rs_dw = rsrc_of(P.dwpart);
        const int nr0 = H * H + H * O, n_dw = G.nb * natt;
        const int n4 = P.dw_pad / 4;                       // float4 columns per receiver kind
        const bool dw_fast = 4 * n4 <= 256 && ns <= 2 * NSH;   // (else: the plain loop below)
        f32x4 dwv[NSH];
        const int dw_c = tid % (2 * n4 > 0 ? 2 * n4 : 1), dw_half = tid / (2 * n4 > 0 ? 2 * n4 : 1);
        if (dw_fast) {
            const int rk = dw_c / n4, c4 = dw_c - rk * n4;
            const uint32_t o0 = 4u * (uint32_t)((((((int64_t)dir * T + tn) * ns) * 2 + rk) * P.n_chunks + G.chunk) * P.dw_pad + 4 * c4);
            const uint32_t st = 4u * (uint32_t)(2 * P.n_chunks * P.dw_pad);
#pragma unroll
            for (int k = 0; k < NSH; ++k) {
                const int sl = dw_half * NSH + k;
                dwv[k] = (dw_half < 2 && sl < ns) ? ld_sc1(rs_dw, o0 + st * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // ---- sender-MLP part: complete d_pre rows of this kind (time tn) x packed W_s[:, own units]
        {
            const __amdgpu_buffer_rsrc_t rs_p = rsrc_of(P.d_pre[K]);
            const int nkb = 2 * h / 32;
            f32x4 acc[MK];
            uint32_t off[MK];
#pragma unroll
            for (int i = 0; i < MK; ++i) {
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int r = min(i * 16 + i16, R_K - 1), bl = r / E_K, e = r - bl * E_K, b = G.b0 + bl;
                off[i] = 4u * (uint32_t)(((((int64_t)dir * P.bs + b) * T + tn) * E_K + e) * (2 * h) + 8 * g4);
            }
#ifndef TWOG_SP_Q2_CH
#define TWOG_SP_Q2_CH 3
#endif
#ifdef TWOG_SP_Q2_X3   // (experiment, as in P2: 3 x bf16 products with one accumulator)
            k_stream<KW2, MK, TWOG_SP_Q2_CH>(rs_p, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
                const Planes B = split8(We[j].a, We[j].b);
#pragma unroll
                for (int i = 0; i < MK; ++i) mac6_one(acc[i], split8(A[i][0], A[i][1]), B);
            });
#else
            k_stream<KW2, MK>(rs_p, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
#pragma unroll
                for (int i = 0; i < MK; ++i) mac_f32(acc[i], A[i][0], A[i][1], We[j]);
            });
#endif
#pragma unroll
            for (int i = 0; i < MK; ++i) put_part1(part, MK, wave, i, lane, acc[i]);
        }
        if (dw_fast) {
            f32x4 v = dwv[0];
#pragma unroll
            for (int k = 1; k < NSH; ++k) v += dwv[k];   // (slices beyond ns hold zeros)
            // res is free until the combine below: [half][kind][dw_pad] partial sums
            if (dw_half < 2) *reinterpret_cast<f32x4*>(sC + (size_t)(dw_half * 2 * n4 + dw_c) * 4) = v;
            __syncthreads();
            for (int x = tid; x < n_dw; x += 256) {
                const int bl = x / natt, i = x - bl * natt;
                const int rk = i < nr0 ? 0 : 1, nr = rk == 0 ? nr0 : natt - nr0, j = rk == 0 ? i : i - nr0;
                const int e = rk * P.dw_pad + bl * nr + j;
                sDW[x] = sC[e] + sC[2 * P.dw_pad + e];
            }
        } else {
            for (int x = tid; x < n_dw; x += 256) {
                const int bl = x / natt, i = x - bl * natt;
                const int rk = i < nr0 ? 0 : 1, nr = rk == 0 ? nr0 : natt - nr0, j = rk == 0 ? i : i - nr0;
                const int64_t base = ((((int64_t)dir * T + tn) * ns) * 2 + rk) * P.n_chunks + G.chunk;
                float v = 0.f;
                for (int sl = 0; sl < ns; ++sl)
                    v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                             rs_dw, (int)(4u * (uint32_t)((base + (int64_t)sl * 2 * P.n_chunks) * P.dw_pad + bl * nr + j)), 0, SC1));
                sDW[x] = v;
            }
        }
        SP_STAMP(1);   // sender-MLP product + dL/dw sums
        combine_parts(part, res, MK, min(4, 2 * h / 32));   // (its barriers also order sW / sDW)
        // softmax backward per (clip, relation, receiver): dscore = w (dw - sum_s w dw) scale
        for (int x = tid; x < G.nb * (2 * H + 2 * O); x += 256) {
            const int bl = x / (2 * H + 2 * O), i = x - bl * (2 * H + 2 * O);
            int off, S;
            if (i < H) { off = att_hh(H, O) + i * H; S = H; }
            else if (i < 2 * H) { off = att_oh(H, O) + (i - H) * O; S = O; }
            else if (i < 2 * H + O) { off = att_ho(H, O) + (i - 2 * H) * H; S = H; }
            else { off = att_oo(H, O) + (i - 2 * H - O) * O; S = O; }
            const float* w = sW + bl * natt + off;
            float* d = sDW + bl * natt + off;
            float tt = 0.f;
            for (int k = 0; k < S; ++k) tt = fmaf(w[k], d[k], tt);
            for (int k = 0; k < S; ++k) d[k] = w[k] * (d[k] - tt) * P.scale;
        }
        __syncthreads();
        // coefficient of F[b] in dF[a], a of kind K (attn.hip: every score sends its dscore to receiver and sender)
        for (int x = tid; x < G.nb * E_K * E; x += 256) {
            const int bl = x / (E_K * E), rem = x - bl * E_K * E, a = rem / E, b = rem - a * E;
            const float* d = sDW + bl * natt;
            float v;
            if (K == 0) {
                if (b < H) v = d[att_hh(H, O) + a * H + b] + d[att_hh(H, O) + b * H + a];
                else v = d[att_oh(H, O) + a * O + (b - H)] + d[att_ho(H, O) + (b - H) * H + a];
            } else {
                if (b < H) v = d[att_ho(H, O) + a * H + b] + d[att_oh(H, O) + b * O + a];
                else v = d[att_oo(H, O) + a * O + (b - H)] + d[att_oo(H, O) + (b - H) * O + a];
            }
            sC[x] = v;
        }
        __syncthreads();
        // carry = direct + W_hh part + sender-MLP part + score part (features: the forward states at time t, both kinds)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int x = ps * 64 + ur;
            if (x < R_K) {
                const int bl = x / E_K, a = x - bl * E_K;
                f32x4 df = {0.f, 0.f, 0.f, 0.f};
                const float* c = sC + (size_t)(bl * E_K + a) * E;
                for (int eb = 0; eb < E; ++eb) {
                    const f32x4 f = *reinterpret_cast<const f32x4*>(fT + (size_t)(eb < H ? bl * H + eb : G.RH + bl * O + (eb - H)) * RS + 4 * q);
                    const float cv = c[eb];
#pragma unroll
                    for (int k = 0; k < 4; ++k) df[k] = fmaf(cv, f[k], df[k]);
                }
                const f32x4 cs = *reinterpret_cast<const f32x4*>(res + (size_t)(x / 16) * 16 * RS + (x % 16) * RS + 4 * q);
                carry[ps] = ((direct[ps] + c_hh[ps]) + cs) + df;
            }
        }
    }
    SP_STAMP(2);   // softmax backward, coefficients, carry
    // ---- gate backward of the own units (gru.hip, gru_step_bwd_kernel: same arithmetic)
    const __amdgpu_buffer_rsrc_t rs_gi = rsrc_of(P.d_gi[K]), rs_gh = rsrc_of(P.d_gh[K]);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int x = ps * 64 + ur;
        float du = 0.f;
        f32x4 gi_[3], gh_[3], dprev;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = dout[ps][k] + carry[ps][k];
            const float rg = sr[ps][k], z = sz[ps][k], n = sn[ps][k], hn = shn[ps][k], hp = h0[ps][k];
            const float gnew = (1.0f - z) * n + z * hp;
            du += d * (gnew - hp);
            const float dg = uu[ps] * d;
            float dp = (1.0f - uu[ps]) * d;
            const float dn = dg * (1.0f - z);
            const float dz = dg * (hp - n);
            dp += dg * z;
            const float dn_pre = dn * (1.0f - n * n);
            const float dr_pre = dn_pre * hn * rg * (1.0f - rg);
            const float dz_pre = dz * z * (1.0f - z);
            gi_[0][k] = dr_pre; gi_[1][k] = dz_pre; gi_[2][k] = dn_pre;
            gh_[0][k] = dr_pre; gh_[1][k] = dz_pre; gh_[2][k] = dn_pre * rg;
            dprev[k] = dp;
        }
        direct[ps] = dprev;
        // the sum over this slice's 16 units of a row: the four quads of a row are adjacent lanes
        du += __shfl_xor(du, 1, 64);
        du += __shfl_xor(du, 2, 64);
        if (x < R_K) {
            const int bl = x / E_K, e = x - bl * E_K, b = G.b0 + bl;
            const uint32_t o = 4u * (uint32_t)((((int64_t)b * T + t) * E_K + e) * (6 * h) + dir * 3 * h + col);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                st_sc1(rs_gh, o + 4u * (uint32_t)(c * h), gh_[c]);
                st_sc1(rs_gi, o + 4u * (uint32_t)(c * h), gi_[c]);
            }
            if (q == 0)
                P.du_part[K][(((int64_t)dir * T + t) * ns + G.slice) * ((int64_t)P.bs * E_K) + (int64_t)b * E_K + e] = du;
        }
    }
    SP_STAMP(3);   // gate backward, stores issued
    group_signal(cnt + (2 + K) * CNT_STRIDE);
    SP_STAMP(4);   // drain + signal
    if (first) return true;
    // ---- W_hh part of the next carry: complete d_gh rows of this kind at step s (all slices) x W_hh[:, own units]
    if (!group_wait(cnt + (2 + K) * CNT_STRIDE, (unsigned)(T - s) * ns, nullptr, 0u, P.error, P.spin_limit, flag)) return false;
    SP_STAMP(5);   // waited for the kind's d_gh
    {
        const int nkb = 3 * h / 32;
        f32x4 acc[MK];
        uint32_t off[MK];
#pragma unroll
        for (int i = 0; i < MK; ++i) {
            acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int r = min(i * 16 + i16, R_K - 1), bl = r / E_K, e = r - bl * E_K, b = G.b0 + bl;
            off[i] = 4u * (uint32_t)((((int64_t)b * T + t) * E_K + e) * (6 * h) + dir * 3 * h + 8 * g4);
        }
#ifdef TWOG_SP_Q2_X3
        k_stream<KW3, MK, TWOG_SP_Q2_CH>(rs_gh, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
            const Planes B = split8(Wh[j].a, Wh[j].b);
#pragma unroll
            for (int i = 0; i < MK; ++i) mac6_one(acc[i], split8(A[i][0], A[i][1]), B);
        });
#else
        k_stream<KW3, MK>(rs_gh, off, nkb, wave, [&](int j, const f32x4 (&A)[MK][2]) {
#pragma unroll
            for (int i = 0; i < MK; ++i) mac_f32(acc[i], A[i][0], A[i][1], Wh[j]);
        });
#endif
#pragma unroll
        for (int i = 0; i < MK; ++i) put_part1(part, MK, wave, i, lane, acc[i]);
        combine_parts(part, res, MK, min(4, nkb));
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int x = min(ps * 64 + ur, MK * 16 - 1);
            c_hh[ps] = *reinterpret_cast<const f32x4*>(res + (size_t)(x / 16) * 16 * RS + (x % 16) * RS + 4 * q);
        }
        __syncthreads();   // res is rewritten by the next step's first combine
    }
    SP_STAMP(6);   // W_hh part of the next carry
    return true;
}

struct BwdLds { float *part, *res, *msT, *sW, *sDW, *sC; int* flag; };

// weights once, into registers (see the forward roles): k-major operands here
template <int MH, int MO, int RK, bool X3Q1>
__device__ __forceinline__ void role_q1(const SegBwdArgs& P, const Geo& G, const BwdLds& M) {
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX;
    const int nkb = 3 * P.h / 32;
    WFrag Wr[KW3][2];
#pragma unroll
    for (int j = 0; j < KW3; ++j) {
        const int kb = wave + 4 * j;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
            Wr[j][blk] = kb < nkb ? load_wk_raw(P.w_ihm[RK][G.dir], P.ld_ih[RK], kb * 32, blk * P.h + G.slice * 16, lane) : wfrag_zero();
    }
    for (int s = P.T - 1; s >= 0; --s)
        if (!q1_step<MH, MO, RK, X3Q1>(P, G, s, M.part, M.res, M.msT, M.sW, M.sDW, M.flag, Wr)) return;
}

template <int MK, int K>
__device__ __forceinline__ void role_q2(const SegBwdArgs& P, const Geo& G, const BwdLds& M) {
    const int lane = threadIdx.x & 63, wave = TWOG_SP_WAVE_INDEX;
    WFrag We[KW2], Wh[KW3];
#pragma unroll
    for (int j = 0; j < KW2; ++j) {
        const int kb = wave + 4 * j;
        We[j] = kb < 2 * P.h / 32 ? load_wk_raw(P.w_sp[K], P.h, kb * 32, G.slice * 16, lane) : wfrag_zero();
    }
#pragma unroll
    for (int j = 0; j < KW3; ++j) {
        const int kb = wave + 4 * j;
        Wh[j] = kb < 3 * P.h / 32 ? load_wk_raw(P.w_hh[K][G.dir], P.h, kb * 32, G.slice * 16, lane) : wfrag_zero();
    }
    constexpr int NP = (MK * 16 + 63) / 64;
    f32x4 direct[NP], c_hh[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { direct[i] = f32x4{0.f, 0.f, 0.f, 0.f}; c_hh[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int s = P.T - 1; s >= 0; --s)
        if (!q2_step<MK, K>(P, G, s, M.part, M.res, M.msT, M.sW, M.sDW, M.sC, M.flag, direct, c_hh, We, Wh)) return;
}

template <int MH, int MO, bool X3Q1>
__global__ __launch_bounds__(256, 1) void seg_persist_bwd_kernel(const SegBwdArgs P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT_MAX = 2 * (MH > MO ? MH : MO);
    const int ns = P.h / 16;
    const int W = (int)gridDim.x;
    const int L = ((int)blockIdx.x % 8) * (W / 8) + (int)blockIdx.x / 8;
    const int group = L / (4 * ns), within = L - group * 4 * ns;
    Geo G;
    G.dir = group / P.n_chunks; G.chunk = group - G.dir * P.n_chunks;
    G.slice = within / 4; G.role = within - G.slice * 4;
    G.b0 = G.chunk * P.cpc; G.nb = min(P.cpc, P.bs - G.b0);
    G.RH = G.nb * P.H; G.RO = G.nb * P.O;
    const int E = P.H + P.O, natt = P.H * P.H + 2 * P.H * P.O + P.O * P.O;
    float* part = reinterpret_cast<float*>(smem);                       // [4][NT_MAX][64][4]
    float* res = part + 4 * NT_MAX * 256;                               // [NT_MAX][16][RS]
    float* msT = res + NT_MAX * 16 * RS;                                // [16 (MH + MO)][RS]
    float* sW = msT + 16 * (MH + MO) * RS;                              // [cpc][natt]
    float* sDW = sW + P.cpc * natt;                                     // [max(cpc natt, dw_pad)]
    float* sC = sDW + (P.cpc * natt > P.dw_pad ? P.cpc * natt : P.dw_pad);   // [cpc][E_K][E]
    const int n_sc = P.cpc * (P.H > P.O ? P.H : P.O) * E;
    int* flag = reinterpret_cast<int*>(sC + (n_sc > 4 * P.dw_pad ? n_sc : 4 * P.dw_pad));   // (sC also holds the dL/dw half sums)
    BwdLds M{part, res, msT, sW, sDW, sC, flag};
    if (G.role == 0) role_q1<MH, MO, 0, X3Q1>(P, G, M);
    else if (G.role == 1) role_q1<MH, MO, 1, X3Q1>(P, G, M);
    else if (G.role == 2) role_q2<MH, 0>(P, G, M);
    else role_q2<MO, 1>(P, G, M);
}

// d_u[b][t][e] += the slices' parked sums, both directions, in fixed order: one thread per (clip, time, entity)
__global__ __launch_bounds__(256) void seg_du_reduce_kernel(const float* part, float* du, int bs, int T, int E, int ns,
                                                            const unsigned* error) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bs * T * E || *error != 0u) return;   // a launch that gave up parked nothing complete: d_u stays as it was
    const int e = i % E, t = (i / E) % T, b = i / (E * T);
    const int64_t rows = (int64_t)bs * E, row = (int64_t)b * E + e;
    float acc = 0.f;
    for (int dir = 0; dir < 2; ++dir) {
        const float* p = part + (((int64_t)dir * T + t) * ns) * rows + row;
        for (int k = 0; k < ns; ++k) acc += p[(int64_t)k * rows];
    }
    du[i] += acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct SegPlan { int cpc, n_chunks, mh, mo; unsigned pair_mask; int grid; size_t lds; };

// chunk size, tile counts and grid for a shape on n_cus compute units; false if the shape is not served
bool make_plan(const twog_segrnn_t& S, int n_cus, SegPlan& pl) {
    const int h = S.hidden, H = S.H, O = S.O, bs = S.bs;
    if (h != 64 && h != 128 && h != 256 && h != 512) return false;
    if (!S.msg_segment || !S.rel_hh || !S.rel_ho || !S.rel_oh || !S.rel_oo) return false;
    if (H < 1 || O < 1 || H > MAXH || O > MAXO || bs < 1 || S.T < 1) return false;
    if (!S.gi_h || !S.gi_o || !S.u_h || !S.u_o) return false;
    const int ns = h / 16;
    // as many chunks as the device holds (more chunks = fewer rows per workgroup, more compute units busy)
    int max_chunks = n_cus / (2 * 4 * ns);
    if (max_chunks < 1) return false;
    if (max_chunks > 16) max_chunks = 16;
#ifdef TWOG_PERSIST_ALLOW_SCRATCH
    // root-cause build only: a smaller grid for the same kernel instance (tools/persist_stress.py)
    if (const char* e = getenv("TWOG_SP_MAX_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v < max_chunks) max_chunks = v; }
#endif
    int n_chunks = max_chunks < bs ? max_chunks : bs;
    int cpc = (bs + n_chunks - 1) / n_chunks;
    n_chunks = (bs + cpc - 1) / cpc;
    const int mh = (cpc * H + 15) / 16, mo = (cpc * O + 15) / 16;
    if (mh != 1 || mo < 1 || mo > 2) return false;   // instantiated: (1, 1), (1, 2)
    pl.cpc = cpc; pl.n_chunks = n_chunks; pl.mh = mh; pl.mo = mo;
    // Gram tiles that hold a same-clip pair (unified rows: humans at bl * H + e, objects at 16 mh + bl * O + e)
    unsigned mask = 0;
    for (int bl = 0; bl < cpc; ++bl)
        for (int a = 0; a < H + O; ++a)
            for (int b = 0; b < H + O; ++b) {
                const int ra = a < H ? bl * H + a : 16 * mh + bl * O + (a - H);
                const int rb = b < H ? bl * H + b : 16 * mh + bl * O + (b - H);
                int ta = ra / 16, tb = rb / 16;
                if (ta > tb) { const int k = ta; ta = tb; tb = k; }
                mask |= 1u << (ta * 8 + tb);
            }
    pl.pair_mask = mask;
    pl.grid = 2 * n_chunks * 4 * ns;
    const int MT = mh + mo, NPAIR = MT * (MT + 1) / 2, NT = mh + mo + 3 * (mh > mo ? mh : mo) + NPAIR;
    const int E = H + O, natt = H * H + 2 * H * O + O * O;
    size_t lds = (size_t)4 * NT * 256 * 4 + (size_t)NT * 16 * RS * 4 + (size_t)cpc * (H > O ? H : O) * E * 4 +
                 (size_t)cpc * natt * 4 + (size_t)cpc * O * 4 + 80 * 4 + 128;
    if (lds < 84 * 1024) lds = 84 * 1024;   // more than half of the 160 KB: one workgroup per compute unit
    if (lds > 160 * 1024) return false;
    pl.lds = lds;
    return pl.grid <= n_cus && pl.grid % 8 == 0;
}

int device_cus(int& n_cus) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return -(int)e;
    e = hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev);
    return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace

// 2 if twog_segrnn_fwd_persistent serves this shape on the current device (and is the faster path: it is only served
// where a chunk's rows fit three 16-row tiles, i.e. small batches), 0 if not.
extern "C" int twog_segrnn_persistent_supported(const twog_segrnn_t* desc) {
    int n_cus = 0;
    if (!desc || device_cus(n_cus) != 0) return 0;
    SegPlan pl;
    return make_plan(*desc, n_cus, pl) ? 2 : 0;
}

extern "C" size_t twog_segrnn_persistent_sync_bytes(void) { return (size_t)SYNC_WORDS * 4; }

// Same contract as twog_segrnn_fwd (tmp_gim_* and zeros unused; tmp_gh_* carries W_hh h_prev between the roles).
// sync: device memory, twog_segrnn_persistent_sync_bytes(), ZERO at launch; the last 128-byte line is the error word
// (uint32 index 4064): non-zero after the launch = a wait ran out, the outputs are incomplete, re-run with twog_segrnn_fwd.
extern "C" int twog_segrnn_fwd_persistent(const twog_segrnn_t* desc, void* sync, void* stream) {
    if (!desc || !sync) return -2;
    const twog_segrnn_t& S = *desc;
    int n_cus = 0;
    int rc = device_cus(n_cus);
    if (rc) return rc;
    SegPlan pl;
    if (!make_plan(S, n_cus, pl)) return -2;
    SegArgs P;
    P.bs = S.bs; P.T = S.T; P.H = S.H; P.O = S.O; P.h = S.hidden;
    P.cpc = pl.cpc; P.n_chunks = pl.n_chunks; P.pair_mask = pl.pair_mask;
    P.scale = S.att_scale;
    P.spin_limit = twog_persist_spin_limit();
    twog_jitter_configure();   // (no-op in the shipped library)
    P.gi[0] = S.gi_h; P.gi[1] = S.gi_o; P.u[0] = S.u_h; P.u[1] = S.u_o; P.mask = S.obj_mask;
    for (int d = 0; d < 2; ++d) {
        P.w_hh[0][d] = S.w_hh_h[d]; P.w_hh[1][d] = S.w_hh_o[d];
        P.b_hh[0][d] = S.b_hh_h[d]; P.b_hh[1][d] = S.b_hh_o[d];
        P.w_ihm[0][d] = S.w_ihm_h[d]; P.w_ihm[1][d] = S.w_ihm_o[d];
    }
    P.ld_ih[0] = S.ld_ih_h; P.ld_ih[1] = S.ld_ih_o;
    const int64_t hh = (int64_t)S.hidden * S.hidden;
    P.w_s[0] = S.w_smsg_h; P.w_s[1] = S.w_smsg_h + hh; P.w_s[2] = S.w_smsg_o; P.w_s[3] = S.w_smsg_o + hh;
    P.b_s[0] = S.b_smsg_h; P.b_s[1] = S.b_smsg_h ? S.b_smsg_h + S.hidden : nullptr;
    P.b_s[2] = S.b_smsg_o; P.b_s[3] = S.b_smsg_o ? S.b_smsg_o + S.hidden : nullptr;
    P.hs[0] = S.hs_h; P.hs[1] = S.hs_o; P.save[0] = S.save_h; P.save[1] = S.save_o;
    P.msrc[0] = S.msrc_h; P.msrc[1] = S.msrc_o; P.mg[0] = S.mg_h; P.mg[1] = S.mg_o;
    P.gh[0] = S.tmp_gh_h; P.gh[1] = S.tmp_gh_o;
    P.att = S.att;
    P.cnt = static_cast<unsigned*>(sync);
    P.error = P.cnt + ERR_WORD;
    if ((size_t)2 * pl.n_chunks * 4 * CNT_STRIDE > (size_t)ERR_WORD) return -2;
    hipStream_t st = (hipStream_t)stream;
#define TWOG_SP_LAUNCH(MH_, MO_)                                                                              \
    do {                                                                                                      \
        static std::atomic<uint32_t> done{0};                                                                 \
        twog_allow_dynamic_lds(seg_persist_fwd_kernel<MH_, MO_>, 160 * 1024, done);                           \
        if (!twog_persist_grid_fits(seg_persist_fwd_kernel<MH_, MO_>, pl.grid, pl.lds, n_cus))               \
            return TWOG_PERSIST_NOT_RESIDENT;                                                                 \
        hipLaunchKernelGGL((seg_persist_fwd_kernel<MH_, MO_>), dim3(pl.grid), dim3(256), pl.lds, st, P);      \
    } while (0)
    if (pl.mo == 1) TWOG_SP_LAUNCH(1, 1);
    else TWOG_SP_LAUNCH(1, 2);
#undef TWOG_SP_LAUNCH
    TWOG_CHECK_LAUNCH();
    return 0;
}

namespace {
int bwd_dw_pad(const twog_segrnn_t& S, int cpc) {
    const int nr0 = S.H * S.H + S.H * S.O, nr1 = S.H * S.O + S.O * S.O;
    const int n = cpc * (nr0 > nr1 ? nr0 : nr1);
    return (n + 3) / 4 * 4;
}
size_t bwd_lds(const twog_segrnn_t& S, const SegPlan& pl) {
    const int mk = pl.mh > pl.mo ? pl.mh : pl.mo, NT = 2 * mk, E = S.H + S.O, natt = S.H * S.H + 2 * S.H * S.O + S.O * S.O;
    const int dw_pad = bwd_dw_pad(S, pl.cpc);
    size_t lds = (size_t)4 * NT * 256 * 4 + (size_t)NT * 16 * RS * 4 + (size_t)16 * (pl.mh + pl.mo) * RS * 4 +
                 (size_t)pl.cpc * natt * 4 + (size_t)(pl.cpc * natt > dw_pad ? pl.cpc * natt : dw_pad) * 4 +
                 (size_t)(pl.cpc * (S.H > S.O ? S.H : S.O) * E > 4 * dw_pad ? pl.cpc * (S.H > S.O ? S.H : S.O) * E : 4 * dw_pad) * 4 + 64;
    if (lds < 84 * 1024) lds = 84 * 1024;
    return lds;
}
}  // namespace

// bytes of the caller-owned scratch of twog_segrnn_bwd_persistent (the slices' dL/dw shares and d_u partial sums); 0 if the
// shape is not served
extern "C" size_t twog_segrnn_bwd_persistent_scratch_bytes(const twog_segrnn_t* desc) {
    int n_cus = 0;
    if (!desc || device_cus(n_cus) != 0) return 0;
    SegPlan pl;
    if (!make_plan(*desc, n_cus, pl)) return 0;
    const twog_segrnn_t& S = *desc;
    const int ns = S.hidden / 16;
    const size_t dw = (size_t)2 * S.T * ns * 2 * pl.n_chunks * bwd_dw_pad(S, pl.cpc);
    const size_t du = (size_t)2 * S.T * ns * ((size_t)S.bs * S.H + (size_t)S.bs * S.O);
    return (dw + du) * 4 + 256;
}

// Same outputs as twog_segrnn_bwd (d_gi_*, d_gh_*, d_pre_*, d_u_* +=; carry_*, tmp_dmg_*, trash, du_part_* unused).
// scratch: twog_segrnn_bwd_persistent_scratch_bytes(desc) of device memory (contents undefined); sync as for the forward.
extern "C" int twog_segrnn_bwd_persistent(const twog_segrnn_t* desc, const twog_segrnn_bwd_t* bdesc, void* scratch,
                                          size_t scratch_bytes, void* sync, void* stream) {
    if (!desc || !bdesc || !sync || !scratch) return -2;
    const twog_segrnn_t& S = *desc;
    const twog_segrnn_bwd_t& B = *bdesc;
    int n_cus = 0;
    int rc = device_cus(n_cus);
    if (rc) return rc;
    SegPlan pl;
    if (!make_plan(S, n_cus, pl)) return -2;
    if (scratch_bytes < twog_segrnn_bwd_persistent_scratch_bytes(desc)) return -2;
    if (!B.d_hs_h || !B.d_hs_o || !B.d_gi_h || !B.d_gi_o || !B.d_gh_h || !B.d_gh_o || !B.d_pre_h || !B.d_pre_o || !B.d_u_h || !B.d_u_o)
        return -2;
    const int ns = S.hidden / 16;
    SegBwdArgs P;
    P.bs = S.bs; P.T = S.T; P.H = S.H; P.O = S.O; P.h = S.hidden;
    P.cpc = pl.cpc; P.n_chunks = pl.n_chunks; P.dw_pad = bwd_dw_pad(S, pl.cpc);
    P.scale = S.att_scale;
    P.spin_limit = twog_persist_spin_limit();
    twog_jitter_configure();   // (no-op in the shipped library)
    P.u[0] = S.u_h; P.u[1] = S.u_o;
    for (int d = 0; d < 2; ++d) {
        P.w_hh[0][d] = S.w_hh_h[d]; P.w_hh[1][d] = S.w_hh_o[d];
        P.w_ihm[0][d] = S.w_ihm_h[d]; P.w_ihm[1][d] = S.w_ihm_o[d];
    }
    P.ld_ih[0] = S.ld_ih_h; P.ld_ih[1] = S.ld_ih_o;
    P.w_sp[0] = S.w_smsg_h; P.w_sp[1] = S.w_smsg_o;
    P.hs[0] = S.hs_h; P.hs[1] = S.hs_o; P.save[0] = S.save_h; P.save[1] = S.save_o;
    P.msrc[0] = S.msrc_h; P.msrc[1] = S.msrc_o; P.att = S.att;
    P.d_hs[0] = B.d_hs_h; P.d_hs[1] = B.d_hs_o;
    P.d_gi[0] = B.d_gi_h; P.d_gi[1] = B.d_gi_o; P.d_gh[0] = B.d_gh_h; P.d_gh[1] = B.d_gh_o;
    P.d_pre[0] = B.d_pre_h; P.d_pre[1] = B.d_pre_o;
    float* sc = static_cast<float*>(scratch);
    P.dwpart = sc;
    const size_t dw = (size_t)2 * S.T * ns * 2 * pl.n_chunks * P.dw_pad;
    P.du_part[0] = sc + dw;
    P.du_part[1] = P.du_part[0] + (size_t)2 * S.T * ns * S.bs * S.H;
    P.cnt = static_cast<unsigned*>(sync);
    P.error = P.cnt + ERR_WORD;
    if ((size_t)2 * pl.n_chunks * 4 * CNT_STRIDE > (size_t)ERR_WORD) return -2;
    const size_t lds = bwd_lds(S, pl);
    if (lds > 160 * 1024) return -2;
    hipStream_t st = (hipStream_t)stream;
#define TWOG_SPB_LAUNCH(MH_, MO_, X3_)                                                                        \
    do {                                                                                                      \
        static std::atomic<uint32_t> done{0};                                                                 \
        twog_allow_dynamic_lds(seg_persist_bwd_kernel<MH_, MO_, X3_>, 160 * 1024, done);                      \
        if (!twog_persist_grid_fits(seg_persist_bwd_kernel<MH_, MO_, X3_>, pl.grid, lds, n_cus))             \
            return TWOG_PERSIST_NOT_RESIDENT;                                                                 \
        hipLaunchKernelGGL((seg_persist_bwd_kernel<MH_, MO_, X3_>), dim3(pl.grid), dim3(256), lds, st, P);    \
    } while (0)
    const bool x3q1 = S.hidden >= 256;   // (see q1_step)
    if (pl.mo == 1) { if (x3q1) TWOG_SPB_LAUNCH(1, 1, true); else TWOG_SPB_LAUNCH(1, 1, false); }
    else { if (x3q1) TWOG_SPB_LAUNCH(1, 2, true); else TWOG_SPB_LAUNCH(1, 2, false); }
#undef TWOG_SPB_LAUNCH
    TWOG_CHECK_LAUNCH();
    // the parked d_u sums (skipped after a launch that gave up: the caller re-runs the pass, which adds into d_u itself)
    hipLaunchKernelGGL(seg_du_reduce_kernel, dim3((S.bs * S.T * S.H + 255) / 256), dim3(256), 0, st, P.du_part[0], B.d_u_h, S.bs,
                       S.T, S.H, ns, P.error);
    hipLaunchKernelGGL(seg_du_reduce_kernel, dim3((S.bs * S.T * S.O + 255) / 256), dim3(256), 0, st, P.du_part[1], B.d_u_o, S.bs,
                       S.T, S.O, ns, P.error);
    TWOG_CHECK_LAUNCH();
    return 0;
}
