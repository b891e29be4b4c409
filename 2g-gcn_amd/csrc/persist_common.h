// Shared by the persistent launches (gru_persist.hip, seg_persist.hip): kernels whose workgroups order their steps among
// themselves through agent-scope counters and therefore need EVERY workgroup of the grid resident at once.
//
//  * before the launch the host asks the runtime how many workgroups of this kernel (its registers, its LDS) fit a compute
//    unit (hipOccupancyMaxActiveBlocksPerMultiprocessor) and refuses a grid the device cannot hold
//    (TWOG_PERSIST_NOT_RESIDENT: the caller runs the launch-per-step path instead);
//  * what the runtime cannot know -- another tenant holding compute units (a second process, another stream's long
//    kernel, a CU mask) -- is caught inside the launch: every wait is bounded; the first wave whose bound runs out sets the
//    launch's error word and LEAVES, every other waiting wave sees the word within 256 spins and leaves too. No trap: the
//    context survives, the host reads the word after the launch and re-runs the pass on the launch-per-step path (outputs
//    are written in place, so the re-run is idempotent).
#pragma once
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "twog_common.h"

// polls (each an L2 round trip plus an s_sleep, ~1 us) before a wait gives up: 2^16, i.e. several times the ~20 ms of a
// whole training step at the batches these launches serve (a healthy hand-off takes microseconds; round 5 waited 2^24 polls,
// ~16 s per timed-out wait: ADVICE r05); TWOG_PERSIST_SPIN_LIMIT overrides. 0 is the test hook of the recovery path: every wait gives up at once, whether or not
// its counter has arrived (an idle device hands over within a poll or two, so a small positive limit proves nothing).
inline int twog_persist_spin_limit() {
    const char* e = getenv("TWOG_PERSIST_SPIN_LIMIT");
    if (e && *e) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 0) return (int)(v > (1L << 30) ? (1L << 30) : v);
    }
    return 1 << 16;
}

// true if `grid` workgroups of `kernel` (256 threads, `lds` bytes of dynamic LDS) can be resident at once on a device with
// n_cus compute units that this process has to itself. The runtime's answers (occupancy, scratch use) depend only on
// (kernel, lds): asked once per kernel instance and LDS size, not on every launch of the latency-critical path.
template <class K>
inline bool twog_persist_grid_fits(K kernel, int grid, size_t lds, int n_cus) {
    struct Answer { size_t lds; int per_cu; bool scratch; };
    static std::mutex mu;                 // (one static per template instance = per kernel)
    static std::vector<Answer> known;
    int per_cu = -1;
    bool scratch = false;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Answer& a : known)
            if (a.lds == lds) { per_cu = a.per_cu; scratch = a.scratch; }
    }
    if (per_cu < 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        scratch = attr.localSizeBytes != 0;
        std::lock_guard<std::mutex> lock(mu);
        known.push_back(Answer{lds, per_cu, scratch});
    }
    // A persistent kernel that needs scratch (private) memory is refused: its reduction loops keep up to 512 registers per
    // lane busy, and a spill in them is a performance cliff (every reload waits for ALL loads in flight: vmcnt is in order).
    // (Round 5 also blamed scratch for WRONG results of one build. Round 6 found the cause elsewhere -- the wave index sat in
    // a vector register, the compiler predicated the unused k-block slots of short reductions with EXEC, and MFMAs, which
    // ignore EXEC, consumed registers the masked-off code never wrote: DESIGN.md "persistent launches", profiles/r06_persist_
    // stress_*.txt. The wave index is scalar now; the spilling variant computes right results with it.)
#if !defined(TWOG_SP_STAMPS) && !defined(TWOG_PERSIST_ALLOW_SCRATCH)
    // (the diagnostic build with phase stamps may spill a few registers, its numbers are read as shares; ALLOW_SCRATCH is the
    // root-cause build of tools/persist_stress.py, which runs the spilling variant on purpose)
    if (scratch) return false;
#endif
    return per_cu >= 1 && (int64_t)per_cu * n_cus >= grid;
}

// Diagnostic build only (-DTWOG_PERSIST_JITTER, `make jitter` -> lib2ggcn_hip_jitter.so; tests/test_kernels_gpu.py::
// test_persistent_hand_offs_hold_under_jitter): a pseudo-random pause of 0 ... ~4 us per wave in front of EVERY publish and
// EVERY poll of the persistent launches, so that the order in which workgroups reach their hand-offs differs from launch to
// launch and from step to step -- a hand-off that only holds by timing shows as a wrong word. The shipped library has none.
#ifdef TWOG_PERSIST_JITTER
// pause length mask (0 ... mask units of ~0.27 us); TWOG_JITTER_MASK=0 runs the SAME binary without pauses (the control of
// tools/persist_stress.py: is a wrong result a matter of timing or of the generated code?)
static __device__ unsigned twog_jitter_mask_dev = 15u;
inline void twog_jitter_configure() {
    unsigned m = 15u;
    if (const char* e = getenv("TWOG_JITTER_MASK")) m = (unsigned)strtoul(e, nullptr, 10);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(twog_jitter_mask_dev), &m, sizeof(m));
}
__device__ __forceinline__ void twog_jitter() {
    // wave-uniform: the cycle counter read through a scalar register, mixed with the workgroup and wave ids
    unsigned x = (unsigned)__builtin_readcyclecounter() ^ (blockIdx.x * 0x9E3779B9u) ^ ((threadIdx.x >> 6) * 0x85EBCA6Bu);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    const int n = __builtin_amdgcn_readfirstlane((int)(x & twog_jitter_mask_dev));
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(10);   // 10 x 64 cycles each
}
#else
inline void twog_jitter_configure() {}
__device__ __forceinline__ void twog_jitter() {}
#endif

// Waits until *counter >= want (agent scope). Returns false when the wave has to leave the kernel (see above).
__device__ __forceinline__ bool twog_wait_counter(const unsigned* counter, unsigned want, unsigned* error, int spin_limit,
                                                  int lane) {
    int ok = 1;
    twog_jitter();
    if (lane == 0) {
        int spins = 0;
        if (spin_limit <= 0) {   // test hook: give up at once
            __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
        } else
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if (spins > spin_limit) {
                __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            if ((spins & 255) == 0 && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                ok = 0;
                break;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}
