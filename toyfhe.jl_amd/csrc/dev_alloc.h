// dev_alloc.h -- the library's device allocator behind tfhe_malloc / tfhe_free.
//
// The reference allocates a fresh Julia array per ring operation and leaves reclamation to the GC; the mirror does the same
// with device buffers, i.e. thousands of short-lived allocations of a handful of sizes per homomorphic circuit.  hipMalloc /
// hipFree per buffer costs microseconds each AND hipFree drains the device, so the host could never run ahead of the GPU
// (the encrypted-MNIST circuit was host-bound for that reason).  This allocator recycles blocks by exact size:
//
//   tfhe_free(p)  : no synchronisation.  An event is recorded on the stream of every live context (all product work runs
//                   on context streams); the block is parked until those events have completed.
//   tfhe_malloc(n): polls the parked blocks in FIFO order (events complete in order), then hands out a ready block of the
//                   same size, or falls back to hipMalloc; on out-of-memory the cache is drained and the call retried.
//
// So a recycled block is never handed out while a kernel enqueued before its tfhe_free can still touch it, whatever stream
// the next user runs on.  TFHE_ALLOC_CACHE=0 in the environment restores plain hipMalloc / hipFree.  Per device; a mutex
// guards the tables (contexts on different host threads share the allocator).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <deque>
#include <mutex>
#include <unordered_map>
#include <vector>

struct tfhe_ctx;

namespace devalloc {

struct parked_t {
    void* p;
    size_t bytes;
    std::vector<hipEvent_t> evs;
};

struct state_t {
    std::mutex mu;
    bool enabled = true, init = false;
    std::vector<hipStream_t*> streams;                           // &ctx->stream of every live context
    std::unordered_map<void*, size_t> live;                      // block -> size (handed out)
    std::unordered_map<size_t, std::vector<void*>> ready;        // size -> recyclable blocks
    std::deque<parked_t> parked;                                 // freed, waiting for their events
    std::vector<hipEvent_t> ev_pool;
    size_t cached_bytes = 0, live_bytes = 0, max_cached = (size_t)96 << 30;
    long n_hip_malloc = 0, n_reuse = 0;
};
inline state_t& S() {
    static state_t* s = new state_t();   // intentionally leaked: contexts may be finalised during interpreter shutdown
    return *s;
}
inline void lazy_init(state_t& s) {
    if (s.init) return;
    s.init = true;
    const char* e = getenv("TFHE_ALLOC_CACHE");
    if (e && e[0] == '0') s.enabled = false;
}

inline void register_stream(hipStream_t* sp) {
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    s.streams.push_back(sp);
}
inline void unregister_stream(hipStream_t* sp) {
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    for (size_t i = 0; i < s.streams.size(); i++)
        if (s.streams[i] == sp) { s.streams.erase(s.streams.begin() + i); break; }
}

// move parked blocks whose events have all completed to the ready lists (front of the queue first; stop at the first busy one)
inline void poll_locked(state_t& s) {
    while (!s.parked.empty()) {
        parked_t& b = s.parked.front();
        bool done = true;
        for (hipEvent_t ev : b.evs)
            if (hipEventQuery(ev) != hipSuccess) { done = false; break; }
        if (!done) { (void)hipGetLastError(); break; }
        for (hipEvent_t ev : b.evs) s.ev_pool.push_back(ev);
        s.ready[b.bytes].push_back(b.p);
        s.parked.pop_front();
    }
}
// give everything cached back to the driver (after draining the device)
inline void trim_locked(state_t& s) {
    (void)hipDeviceSynchronize();
    poll_locked(s);
    for (auto& kv : s.ready)
        for (void* p : kv.second) (void)hipFree(p);
    s.ready.clear();
    for (auto& b : s.parked) {
        for (hipEvent_t ev : b.evs) s.ev_pool.push_back(ev);
        (void)hipFree(b.p);
    }
    s.parked.clear();
    s.cached_bytes = 0;
}

inline hipError_t alloc(size_t bytes, void** out) {
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    lazy_init(s);
    if (bytes == 0) bytes = 8;
    if (!s.enabled) return hipMalloc(out, bytes);
    poll_locked(s);
    auto it = s.ready.find(bytes);
    if (it != s.ready.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        s.cached_bytes -= bytes;
        s.n_reuse++;
    } else {
        hipError_t e = hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            trim_locked(s);
            e = hipMalloc(out, bytes);
        }
        if (e != hipSuccess) return e;
        s.n_hip_malloc++;
    }
    s.live[*out] = bytes;
    s.live_bytes += bytes;
    return hipSuccess;
}

inline hipError_t release(void* p) {
    if (!p) return hipSuccess;
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    lazy_init(s);
    auto it = s.live.find(p);
    if (it == s.live.end()) return hipFree(p);                  // not ours (allocated while the cache was off)
    const size_t bytes = it->second;
    s.live.erase(it);
    s.live_bytes -= bytes;
    if (!s.enabled) return hipFree(p);
    parked_t b{p, bytes, {}};
    for (hipStream_t* sp : s.streams) {
        hipEvent_t ev;
        if (!s.ev_pool.empty()) { ev = s.ev_pool.back(); s.ev_pool.pop_back(); }
        else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return hipFree(p); }
        if (hipEventRecord(ev, *sp) != hipSuccess) {           // a stream the caller has destroyed: fall back to the safe path
            (void)hipGetLastError();
            s.ev_pool.push_back(ev);
            for (hipEvent_t e2 : b.evs) s.ev_pool.push_back(e2);
            return hipFree(p);
        }
        b.evs.push_back(ev);
    }
    s.parked.push_back(std::move(b));
    s.cached_bytes += bytes;
    if (s.cached_bytes > s.max_cached) trim_locked(s);
    return hipSuccess;
}

}  // namespace devalloc
