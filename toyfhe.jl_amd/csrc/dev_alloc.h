// dev_alloc.h -- the library's device allocator behind tfhe_malloc / tfhe_free.
//
// The reference allocates a fresh Julia array per ring operation and leaves reclamation to the GC; the mirror does the same
// with device buffers, i.e. thousands of short-lived allocations of a handful of sizes per homomorphic circuit.  hipMalloc /
// hipFree per buffer costs microseconds each AND hipFree drains the device, so the host could never run ahead of the GPU
// (the encrypted-MNIST circuit was host-bound for that reason).  This allocator recycles blocks by exact size:
//
//   tfhe_free(p)  : no synchronisation.  An event is recorded on the stream of every live context (all product work runs
//                   on context streams); the block is parked until those events have completed.  A cache that has grown past
//                   its bound is trimmed by the NEXT tfhe_malloc (want_trim), never here.  The one waiting path left is the
//                   fallback when an event cannot be created or recorded (a stream the caller destroyed): a plain hipFree.
//   tfhe_malloc(n): polls the parked blocks in FIFO order (events complete in order), then hands out a ready block of the
//                   same size, or falls back to hipMalloc; on out-of-memory the cache is drained and the call retried.
//                   Once the cache holds more than a soft threshold (a quarter of its bound; TFHE_ALLOC_SOFT_GIB) a request
//                   that finds nothing ready waits for the oldest PARKED block of its size instead of allocating: a host
//                   far ahead of the device is held back by memory, it does not run the cache into its bound.
//
// So a recycled block is never handed out while a kernel enqueued before its tfhe_free can still touch it, whatever stream
// the next user runs on.  TFHE_ALLOC_CACHE=0 in the environment restores plain hipMalloc / hipFree.
//
// Per device: a block is cached, parked and handed out again only on the device it was allocated on (the current device of
// the calling thread, tfhe_set_device), and its release events are recorded only on the streams of contexts of that device.
// One mutex guards the tables (contexts on different host threads share the allocator); a context's stream slot is read
// and re-pointed under it (set_stream).  The cache is bounded by a third of the device's memory (at most 96 GiB), and every
// other device allocation of the library goes through malloc_retry, which gives the cache back before reporting
// out-of-memory.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
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
struct live_t {
    size_t bytes;
    int dev;
};
struct stream_slot_t {
    hipStream_t* sp;  // &ctx->stream
    int dev;
};
struct dev_state_t {  // the cache of one device
    std::unordered_map<size_t, std::vector<void*>> ready;        // size -> recyclable blocks
    std::deque<parked_t> parked;                                 // freed, waiting for their events
    std::vector<hipEvent_t> ev_pool;                             // events of this device
    std::unordered_map<size_t, unsigned long> last_use;          // size -> sequence number of the last request for it
    unsigned long seq = 0;
    size_t cached_bytes = 0, max_cached = 0, soft_cached = 0;
    bool want_trim = false;                                      // the cache passed its bound in release(): trimmed by the next alloc()
};

struct state_t {
    std::mutex mu;
    bool enabled = true, init = false;
    std::vector<stream_slot_t> streams;                          // stream slot of every live context, with its device
    std::unordered_map<void*, live_t> live;                      // block -> size, device (handed out)
    std::map<int, dev_state_t> devs;
    size_t cached_bytes = 0, live_bytes = 0;
    long n_hip_malloc = 0, n_reuse = 0;
    // TFHE_ALLOC_DEBUG=1: where alloc() spends its time, printed at exit (design aid)
    bool debug = false;
    double t_wait = 0, t_stale = 0, t_malloc = 0, t_trim = 0;
    long n_wait = 0, n_stale_free = 0, n_trim = 0;
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
    const char* dbg = getenv("TFHE_ALLOC_DEBUG");
    if (dbg && dbg[0] == '1') {
        s.debug = true;
        atexit([] {
            state_t& t = S();
            fprintf(stderr, "[tfhe alloc] hipMalloc %ld (%.1f ms)  reuse %ld  waits %ld (%.1f ms)  stale blocks freed %ld (%.1f ms)  trims %ld (%.1f ms)\n",
                    t.n_hip_malloc, t.t_malloc * 1e3, t.n_reuse, t.n_wait, t.t_wait * 1e3, t.n_stale_free, t.t_stale * 1e3, t.n_trim, t.t_trim * 1e3);
        });
    }
}
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}
inline dev_state_t& dev_locked(state_t& s, int dev) {
    dev_state_t& d = s.devs[dev];
    if (d.max_cached == 0) {
        size_t fr = 0, tot = 0;
        d.max_cached = (size_t)96 << 30;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) d.max_cached = std::min(d.max_cached, tot / 3);
        else (void)hipGetLastError();
        // back-pressure threshold (alloc): a quarter of the bound, TFHE_ALLOC_SOFT_GIB overrides (fractions allowed)
        d.soft_cached = d.max_cached / 4;
        if (const char* e = getenv("TFHE_ALLOC_SOFT_GIB")) {
            const double g = atof(e);
            if (g > 0.0) d.soft_cached = std::min(d.max_cached, (size_t)(g * (double)((size_t)1 << 30)));
        }
    }
    return d;
}

inline void register_stream(hipStream_t* sp) {
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    s.streams.push_back(stream_slot_t{sp, current_device()});
}
inline void unregister_stream(hipStream_t* sp) {
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    for (size_t i = 0; i < s.streams.size(); i++)
        if (s.streams[i].sp == sp) { s.streams.erase(s.streams.begin() + i); break; }
}
// re-point a registered stream slot (tfhe_ctx_set_stream): release() reads the slot under the same mutex
inline void set_stream(hipStream_t* sp, hipStream_t v) {
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    *sp = v;
}

// move parked blocks whose events have all completed to the ready lists (front of the queue first; stop at the first busy one).
// Only hipErrorNotReady means "busy": any other status is a device error (fault, reset, a destroyed stream) and comes back to the
// caller -- a waiting alloc() must not spin on it (r06).
inline hipError_t poll_locked(dev_state_t& d) {
    while (!d.parked.empty()) {
        parked_t& b = d.parked.front();
        for (hipEvent_t ev : b.evs) {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) continue;
            (void)hipGetLastError();
            return q == hipErrorNotReady ? hipSuccess : q;
        }
        for (hipEvent_t ev : b.evs) d.ev_pool.push_back(ev);
        if (b.p) d.ready[b.bytes].push_back(b.p);                // (p == nullptr: the events of a block alloc_ws took while parked)
        d.parked.pop_front();
    }
    return hipSuccess;
}
// the oldest parked block of exactly `bytes`, queried directly (a busy, unrelated block at the front of the queue does not hide it):
// *found = one is parked; returns hipSuccess with *out set when its events have completed (the block leaves the queue, its
// events go back to the pool), hipErrorNotReady while they have not, any other status on a device error
inline hipError_t take_parked_locked(dev_state_t& d, size_t bytes, void** out, bool* found) {
    *found = false;
    for (auto it = d.parked.begin(); it != d.parked.end(); ++it) {
        if (!it->p || it->bytes != bytes) continue;
        *found = true;
        for (hipEvent_t ev : it->evs) {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) continue;
            (void)hipGetLastError();
            return q;
        }
        for (hipEvent_t ev : it->evs) d.ev_pool.push_back(ev);
        *out = it->p;
        d.parked.erase(it);
        return hipSuccess;
    }
    return hipErrorNotReady;
}
// give everything cached on the CURRENT device back to the driver (after draining it)
inline void trim_locked(state_t& s) {
    dev_state_t& d = dev_locked(s, current_device());
    (void)hipDeviceSynchronize();
    (void)poll_locked(d);
    for (auto& kv : d.ready)
        for (void* p : kv.second) (void)hipFree(p);
    d.ready.clear();
    for (auto& b : d.parked) {
        for (hipEvent_t ev : b.evs) d.ev_pool.push_back(ev);
        if (b.p) (void)hipFree(b.p);
    }
    d.parked.clear();
    s.cached_bytes -= d.cached_bytes;
    d.cached_bytes = 0;
}
// hipMalloc for the library's long-lived allocations (workspaces, tables): on out-of-memory the cache is given back first
template <class T>
inline hipError_t malloc_retry(T** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        state_t& s = S();
        std::lock_guard<std::mutex> g(s.mu);
        trim_locked(s);
        e = hipMalloc(out, bytes);
    }
    return e;
}

// ready blocks of sizes nobody has asked for during the last STALE_AFTER requests (left by an earlier phase of the program) back
// to the driver (hipFree drains the device; parked blocks and the sizes in use stay)
constexpr unsigned long STALE_AFTER = 4096;
inline void free_stale_ready_locked(state_t& s, dev_state_t& d) {
    for (auto it = d.ready.begin(); it != d.ready.end();) {
        auto lu = d.last_use.find(it->first);
        if (lu != d.last_use.end() && lu->second + STALE_AFTER >= d.seq) { ++it; continue; }
        const double t0 = s.debug ? now_s() : 0.0;
        for (void* p : it->second) {
            (void)hipFree(p);
            d.cached_bytes -= it->first;
            s.cached_bytes -= it->first;
            s.n_stale_free++;
        }
        if (s.debug) s.t_stale += now_s() - t0;
        if (lu != d.last_use.end()) d.last_use.erase(lu);
        it = d.ready.erase(it);
    }
}

inline hipError_t alloc(size_t bytes, void** out) {
    state_t& s = S();
    std::unique_lock<std::mutex> g(s.mu);
    lazy_init(s);
    if (bytes == 0) bytes = 8;
    if (!s.enabled) return hipMalloc(out, bytes);
    const int dev = current_device();
    dev_state_t& d = dev_locked(s, dev);
    if (d.want_trim) {                                           // deferred from release(), which must never wait (GC finalizer threads)
        d.want_trim = false;
        if (d.cached_bytes > d.max_cached) {
            const double t0 = s.debug ? now_s() : 0.0;
            trim_locked(s);
            s.n_trim++;
            if (s.debug) s.t_trim += now_s() - t0;
        }
    }
    (void)poll_locked(d);
    d.last_use[bytes] = ++d.seq;
    auto it = d.ready.find(bytes);
    if ((it == d.ready.end() || it->second.empty()) && d.cached_bytes > d.soft_cached) {
        // Back-pressure (r04).  A host that enqueues far ahead of the device frees blocks long before their release events complete:
        // nothing is ready when the same sizes are asked for again, every request becomes a hipMalloc, the cache runs into its bound
        // and is trimmed wholesale (device drain + hipFree + hipMalloc of everything, once per pass: the chained-rotation MNIST
        // circuit went from 0.29 s to 2.5 s per pass that way).  Past the soft threshold: first give back ready blocks of sizes
        // that have gone out of use (left by an earlier phase), then WAIT for the oldest parked block of this size instead of
        // allocating -- the host stays ahead of the device by the blocks it already owns, not by ever more memory.
        free_stale_ready_locked(s, d);
        if (d.cached_bytes > d.soft_cached) {
            // Wait by POLLING under the lock (r05): the events of a parked block belong to the tables -- poll_locked on
            // another thread may return them to the pool and release() re-record them on a busy stream while this thread is
            // unlocked, so synchronising on copied handles could block on unrelated later work.  Between polls the mutex is free
            // (release() on finalizer threads never waits behind this); the loop ends when a block of this size is ready or none
            // of this size is parked any more (a trim on another thread).
            // r06: the block waited for is the oldest parked one of THIS size, queried directly; a device error or a wait
            // beyond MAX_WAIT_SPINS (about two seconds: a wedged stream) ends the loop and the request falls through to hipMalloc,
            // which reports what the device has to say.
            constexpr unsigned MAX_WAIT_SPINS = 100000;
            const double tw0 = s.debug ? now_s() : 0.0;
            s.n_wait++;
            for (unsigned spin = 0; spin < MAX_WAIT_SPINS; spin++) {
                it = d.ready.find(bytes);
                if (it != d.ready.end() && !it->second.empty()) break;
                void* got = nullptr;
                bool parked_one = false;
                const hipError_t q = take_parked_locked(d, bytes, &got, &parked_one);
                if (q == hipSuccess && got) { d.ready[bytes].push_back(got); break; }
                if (!parked_one || q != hipErrorNotReady) break;
                g.unlock();
                if (spin < 256) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(20));
                g.lock();
                if (poll_locked(d) != hipSuccess) break;
            }
            if (s.debug) s.t_wait += now_s() - tw0;
        }
        it = d.ready.find(bytes);
    }
    if (it != d.ready.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        d.cached_bytes -= bytes;
        s.cached_bytes -= bytes;
        s.n_reuse++;
    } else {
        const double tm0 = s.debug ? now_s() : 0.0;
        hipError_t e = hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            trim_locked(s);
            e = hipMalloc(out, bytes);
        }
        if (s.debug) s.t_malloc += now_s() - tm0;
        if (e != hipSuccess) return e;
        s.n_hip_malloc++;
    }
    s.live[*out] = live_t{bytes, dev};
    s.live_bytes += bytes;
    return hipSuccess;
}

// A large per-call workspace for work on ONE stream (ensure_ws(pooled)): any cached block of `bytes` .. 2 x `bytes` serves --
// a ready one as it is, a PARKED one after `stream` has been made to wait for its release events (the device orders the
// reuse; the host does not wait), so successive calls of similar shape share one block even while the host runs ahead of
// the GPU.  *got = the block's real size.
inline hipError_t alloc_ws(size_t bytes, hipStream_t stream, void** out, size_t* got) {
    {
        state_t& s = S();
        std::lock_guard<std::mutex> g(s.mu);
        lazy_init(s);
        if (s.enabled) {
            const int dev = current_device();
            dev_state_t& d = dev_locked(s, dev);
            (void)poll_locked(d);
            size_t best = 0;
            for (auto& kv : d.ready)
                if (kv.first >= bytes && kv.first <= 2 * bytes && !kv.second.empty() && (best == 0 || kv.first < best)) best = kv.first;
            if (best) {
                *out = d.ready[best].back();
                d.ready[best].pop_back();
            } else {
                for (auto it = d.parked.begin(); it != d.parked.end(); ++it) {
                    if (it->bytes < bytes || it->bytes > 2 * bytes) continue;
                    bool ok = true;
                    for (hipEvent_t ev : it->evs)
                        if (hipStreamWaitEvent(stream, ev, 0) != hipSuccess) { (void)hipGetLastError(); ok = false; }
                    if (!ok) continue;
                    // the events stay out of the pool until they have completed: recycle them through a zero-byte parked entry
                    best = it->bytes;
                    *out = it->p;
                    parked_t rest{nullptr, 0, std::move(it->evs)};
                    *it = std::move(rest);
                    break;
                }
            }
            if (best) {
                d.cached_bytes -= best;
                s.cached_bytes -= best;
                s.n_reuse++;
                s.live[*out] = live_t{best, dev};
                s.live_bytes += best;
                *got = best;
                return hipSuccess;
            }
        }
    }
    *got = bytes;
    return alloc(bytes, out);
}

inline hipError_t release(void* p) {
    if (!p) return hipSuccess;
    state_t& s = S();
    std::lock_guard<std::mutex> g(s.mu);
    lazy_init(s);
    auto it = s.live.find(p);
    if (it == s.live.end()) return hipFree(p);                  // not ours (allocated while the cache was off)
    const size_t bytes = it->second.bytes;
    const int dev = it->second.dev;
    s.live.erase(it);
    s.live_bytes -= bytes;
    if (!s.enabled) return hipFree(p);
    // events and the cache of the block's own device (the caller may have switched devices since the allocation)
    const int cur = current_device();
    if (cur != dev && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return hipFree(p); }
    dev_state_t& d = dev_locked(s, dev);
    parked_t b{p, bytes, {}};
    hipError_t rc = hipSuccess;
    bool park = true;
    for (const stream_slot_t& sl : s.streams) {
        if (sl.dev != dev) continue;                             // work on other devices cannot touch this block
        hipEvent_t ev;
        if (!d.ev_pool.empty()) { ev = d.ev_pool.back(); d.ev_pool.pop_back(); }
        else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); park = false; break; }
        if (hipEventRecord(ev, *sl.sp) != hipSuccess) {         // a stream the caller has destroyed: fall back to the safe path
            (void)hipGetLastError();
            d.ev_pool.push_back(ev);
            park = false;
            break;
        }
        b.evs.push_back(ev);
    }
    if (park) {
        d.parked.push_back(std::move(b));
        d.cached_bytes += bytes;
        s.cached_bytes += bytes;
        if (d.cached_bytes > d.max_cached) d.want_trim = true;   // no device synchronisation here: the next alloc() trims
    } else {
        for (hipEvent_t e2 : b.evs) d.ev_pool.push_back(e2);
        rc = hipFree(p);                                         // synchronising free
    }
    if (cur != dev) (void)hipSetDevice(cur);
    return rc;
}

}  // namespace devalloc
