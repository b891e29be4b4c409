// Replays a recorded sequence of library calls for the further steps of a loop whose descriptors are AFFINE in the step
// index -- the host-composed general segment-level loop (ops.segment_recurrence_general_*): every operand of chain step
// s lives in slot s of a per-step buffer, or is the same for all steps, so each 64-bit word of each descriptor (a device
// pointer, a stride, a pair of int32 sizes, a float) is  w(s) = w(a) + k * (w(b) - w(a))  for the two consecutive steps
// a, b the host composed and recorded. The host then composes three steps of a 120-step chain instead of all of them
// (its Python time per step was 10x the GPU's); the launches are exactly those the host would have issued.
#include <vector>

#include "twog_common.h"

namespace {

size_t desc_bytes(int kind) {
    switch (kind) {
        case TWOG_TAPE_GEMM: return sizeof(twog_gemm_t);
        case TWOG_TAPE_RELATION_FWD: return sizeof(twog_relation_t);
        case TWOG_TAPE_RELATION_BWD: return sizeof(twog_relation_bwd_t);
        case TWOG_TAPE_GRU_STEP_FWD: return sizeof(twog_gru_step_t);
        case TWOG_TAPE_GRU_STEP_BWD: return sizeof(twog_gru_step_bwd_t);
        case TWOG_TAPE_ROWOPS: return sizeof(twog_rowop_t);
        default: return 0;
    }
}

}  // namespace

extern "C" int twog_tape_run(const twog_tape_entry_t* step_a, const twog_tape_entry_t* step_b, int n_entries, int k_begin,
                             int k_end, void* workspace, size_t workspace_bytes, void* stream) {
    static_assert(sizeof(twog_gemm_t) % 8 == 0 && sizeof(twog_relation_t) % 8 == 0 && sizeof(twog_relation_bwd_t) % 8 == 0 &&
                      sizeof(twog_gru_step_t) % 8 == 0 && sizeof(twog_gru_step_bwd_t) % 8 == 0 && sizeof(twog_rowop_t) % 8 == 0,
                  "descriptors are sequences of 64-bit words");
    if (n_entries < 0 || (n_entries > 0 && (!step_a || !step_b))) return -2;
    size_t most = 0;
    for (int i = 0; i < n_entries; ++i) {
        const twog_tape_entry_t &a = step_a[i], &b = step_b[i];
        const size_t sz = desc_bytes(a.kind);
        if (!sz || a.kind != b.kind || a.n != b.n || a.flags != b.flags || a.n < 0 || (a.n > 0 && (!a.desc || !b.desc)))
            return -2;   // the two steps are not the same program
        if (sz * a.n > most) most = sz * a.n;
    }
    std::vector<uint64_t> words(most / 8 + 1);
    for (int k = k_begin; k < k_end; ++k)
        for (int i = 0; i < n_entries; ++i) {
            const twog_tape_entry_t &a = step_a[i], &b = step_b[i];
            if (a.n == 0) continue;
            const size_t nw = desc_bytes(a.kind) * a.n / 8;
            const uint64_t* wa = static_cast<const uint64_t*>(a.desc);
            const uint64_t* wb = static_cast<const uint64_t*>(b.desc);
            for (size_t j = 0; j < nw; ++j) words[j] = wa[j] + (uint64_t)(int64_t)k * (wb[j] - wa[j]);
            const void* d = words.data();
            int rc = 0;
            switch (a.kind) {
                case TWOG_TAPE_GEMM:
                    rc = twog_gemm_f32(static_cast<const twog_gemm_t*>(d), a.n, a.flags & 1, (a.flags >> 1) & 1, workspace,
                                       workspace_bytes, stream);
                    break;
                case TWOG_TAPE_RELATION_FWD: rc = twog_relation_fwd_n(static_cast<const twog_relation_t*>(d), a.n, stream); break;
                case TWOG_TAPE_RELATION_BWD: rc = twog_relation_bwd_n(static_cast<const twog_relation_bwd_t*>(d), a.n, stream); break;
                case TWOG_TAPE_GRU_STEP_FWD: rc = twog_gru_step_fwd(static_cast<const twog_gru_step_t*>(d), a.n, stream); break;
                case TWOG_TAPE_GRU_STEP_BWD: rc = twog_gru_step_bwd(static_cast<const twog_gru_step_bwd_t*>(d), a.n, stream); break;
                case TWOG_TAPE_ROWOPS: rc = twog_rowops(static_cast<const twog_rowop_t*>(d), a.n, stream); break;
            }
            if (rc) return rc;
        }
    return 0;
}
