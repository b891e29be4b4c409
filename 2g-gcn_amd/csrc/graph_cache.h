// hipGraph cache for the launch-bound time loops (frame-level BiGRU and segment-level recurrence, forward and
// backward: 240-720 dependent launches per call). OPT-IN since round 3 (TWOG_GRAPHS=1; TWOG_NO_GRAPHS=1 still forces it
// off): the loops' kernels are no longer shorter than the C++ launch path that issues them (fused gate epilogues, fewer
// and longer launches), and a replayed node costs 0.8-1.6 us MORE than the same kernel launched directly on ROCm 7.2
// (bench, graphs on -> off: bs64 84.8 -> 83.9 ms, 8 clips hs512 26.1 -> 24.7 ms, 16 clips h=64 20.9 -> 18.9 ms per step).
// The graphs remain the remedy for a host-bound deployment (slow or shared host cores): then a loop is captured once
// (stream capture of exactly the launches the loop issues) and replayed with one hipGraphLaunch while its descriptor --
// every pointer, shape and stride -- stays the same, the steady state of a training loop under a caching allocator. The 64-bit hash of the
// descriptor only picks the bucket: every entry keeps the descriptor BYTES and a hit is confirmed with a compare, so two
// descriptors that collide can never replay each other's pointers (a mismatch probes on under a salted key and
// captures its own graph). The cache is small (64 loops) and flushed when full. TWOG_NO_GRAPHS=1 turns it off;
// TWOG_GRAPH_HASH_BITS=k (tests) truncates the hash to k bits to force collisions.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>

namespace twog_graph {

inline uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

struct Entry { hipGraphExec_t exec; int uses; std::string desc; };

// descriptor bytes of a loop: the caller appends every struct / array its launches read
struct Desc {
    std::string bytes;
    Desc& add(const void* p, size_t n) { bytes.append(static_cast<const char*>(p), n); return *this; }
    template <class T> Desc& pod(const T& v) { return add(&v, sizeof(T)); }
};

// One instance per process (inline function static: shared by every translation unit of the library). The mutex makes
// the entry points callable from several host threads (calls are serialised: capture is thread-local and a replay is
// one launch); the per-device side stream and the device ordinal mixed into the key keep a process that drives more
// than one GPU correct, although the intended deployment is one process per GPU.
constexpr int MAX_DEVICES = 16;
struct State {
    std::mutex mu;
    std::unordered_map<uint64_t, Entry> cache;
    std::unordered_set<std::string> seen;            // a descriptor is captured the second time it shows up
    uint64_t collisions = 0;                         // bucket hits whose descriptor bytes differed
    hipStream_t side[MAX_DEVICES] = {};
    hipEvent_t ev_in[MAX_DEVICES] = {}, ev_out[MAX_DEVICES] = {};
};
inline State& state() {
    static State s;
    return s;
}

inline uint64_t collisions() { return state().collisions; }

// enqueue(stream): issues the loop's launches on `stream`, returns 0 on success. desc: everything the launches read.
// Graphs are captured and replayed on a stream owned by the library (the caller's stream may be the legacy default
// stream, which cannot be captured), fenced against the caller's stream with two events.
template <class F>
int run(const Desc& d, hipStream_t user, F enqueue) {
    // read per call (two getenv calls against hundreds of launches): tests switch it inside one process
    const char* on_env = getenv("TWOG_GRAPHS");
    if (!on_env || atoi(on_env) == 0 || getenv("TWOG_NO_GRAPHS") != nullptr) return enqueue(user);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        (void)hipGetLastError();
        return enqueue(user);
    }
    State& S = state();
    std::lock_guard<std::mutex> lock(S.mu);
    static const int hash_bits = getenv("TWOG_GRAPH_HASH_BITS") ? atoi(getenv("TWOG_GRAPH_HASH_BITS")) : 64;
    uint64_t key = fnv1a(d.bytes.data(), d.bytes.size());
    if (hash_bits < 64) key = hash_bits <= 0 ? 0 : (key & ((1ull << hash_bits) - 1));
    key ^= 0x9e3779b97f4a7c15ull * (uint64_t)(dev + 1);
    hipStream_t& side = S.side[dev];
    hipEvent_t &ev_in = S.ev_in[dev], &ev_out = S.ev_out[dev];
    auto& cache = S.cache;
    auto& seen = S.seen;
    if (!side) {
        if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            side = nullptr;
            return enqueue(user);
        }
    }
    // the hash picks the bucket, the bytes decide: an entry with other bytes is a collision -> probe on (salted key)
    auto it = cache.find(key);
    while (it != cache.end() && it->second.desc != d.bytes) {
        ++S.collisions;
        key = key * 6364136223846793005ull + 1442695040888963407ull;
        it = cache.find(key);
    }
    if (it == cache.end()) {
        if (seen.find(d.bytes) == seen.end()) {   // keyed by the bytes themselves: immune to bucket collisions
            if (seen.size() > 256) seen.clear();
            seen.insert(d.bytes);
            return enqueue(user);  // first sighting (also covers lazy one-time setup inside the launch paths)
        }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(side, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            return enqueue(user);
        }
        const int rc = enqueue(side);
        const hipError_t ec = hipStreamEndCapture(side, &graph);
        hipGraphExec_t exec = nullptr;
        if (rc != 0 || ec != hipSuccess || !graph ||
            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            return rc != 0 ? rc : enqueue(user);  // nothing ran during the capture: issue the launches directly
        }
        (void)hipGraphDestroy(graph);
        if (cache.size() >= 64) {
            // flush: wait for replays still in flight on the side streams before their executables go away
            for (int i = 0; i < MAX_DEVICES; ++i)
                if (S.side[i]) (void)hipStreamSynchronize(S.side[i]);
            for (auto& kv : cache) (void)hipGraphExecDestroy(kv.second.exec);
            cache.clear();
        }
        it = cache.emplace(key, Entry{exec, 0, d.bytes}).first;
    }
    ++it->second.uses;
    if (hipEventRecord(ev_in, user) != hipSuccess || hipStreamWaitEvent(side, ev_in, 0) != hipSuccess) return -101;
    if (hipGraphLaunch(it->second.exec, side) != hipSuccess) return -100;
    if (hipEventRecord(ev_out, side) != hipSuccess || hipStreamWaitEvent(user, ev_out, 0) != hipSuccess) return -102;
    return 0;
}

}  // namespace twog_graph
