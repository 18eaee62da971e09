#include "devpool.hpp"

#include <cstdlib>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace kvx {
namespace {
std::mutex g_mu;
struct Key {
    int dev; size_t bytes;
    bool operator<(const Key &o) const { return dev != o.dev ? dev < o.dev : bytes < o.bytes; }
};
std::multimap<Key, void *> g_free;               // cached blocks by (device, size)
std::unordered_map<void *, Key> g_live;          // blocks handed out -> (device, size)
size_t g_cached = 0;
constexpr size_t MAX_BLOCK = (size_t)1 << 30;

std::multimap<int, hipStream_t> g_streams;      // by device
std::multimap<int, hipStream_t> g_streams_hi;   // ... of the highest priority
std::multimap<int, hipEvent_t> g_events[2];      // [timing] by device

size_t cache_cap()
{
    static const size_t cap = [] {
        const char *e = getenv("KVX_POOL_MAX_MB");
        return (size_t)(e ? atoll(e) : 8192) << 20;
    }();
    return cap;
}

// hipStreamCreate costs ~3 ms on MI355X; a factor object holds three to five streams, so the first call on a new pattern while the
// streams of earlier objects are still in use paid 10-15 ms for them (klu.linsolve on ACTIVSg2000: 15 ms against 5.6 with streams at
// hand).  When a request finds the pool empty, a background thread tops it up with spare streams beside the caller's own work; the
// next objects find them there.  KVX_SPARE_STREAMS=<count per refill> (default 12, 0 = off).
// fork() after the first device use is not supported (as for the HIP runtime itself): the child would inherit the pool's lock in
// whatever state the refill thread left it.
std::mutex g_refill_mu;                          // guards the thread object (the thread itself only takes g_mu)
std::thread g_refill;
std::atomic<bool> g_refill_busy{false};
bool g_refill_atexit = false;

void join_refill()
{
    std::lock_guard<std::mutex> lk(g_refill_mu);
    if (g_refill.joinable()) g_refill.join();
}

void start_refill(int dev)
{
    static const int spare = [] { const char *e = getenv("KVX_SPARE_STREAMS"); return e ? atoi(e) : 12; }();
    if (spare <= 0) return;
    bool expected = false;
    if (!g_refill_busy.compare_exchange_strong(expected, true)) return;
    std::lock_guard<std::mutex> lk(g_refill_mu);
    if (g_refill.joinable()) g_refill.join();    // (a finished earlier refill)
    if (!g_refill_atexit) { g_refill_atexit = true; (void)atexit(join_refill); }   // runs before the HIP runtime's own exit handlers
    try {
        g_refill = std::thread([dev] {
            if (hipSetDevice(dev) == hipSuccess) {
                for (int i = 0; i < spare; i++) {
                    hipStream_t s = nullptr;
                    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
                    std::lock_guard<std::mutex> lk2(g_mu);
                    if (g_streams.count(dev) >= 64) { (void)hipStreamDestroy(s); break; }   // (per device: one device's spares do not block another's)
                    g_streams.emplace(dev, s);
                }
            }
            g_refill_busy.store(false);
        });
    } catch (...) {
        g_refill_busy.store(false);
    }
}

void release_all_locked()
{
    for (auto &kv : g_free) (void)hipFree(kv.second);
    g_free.clear();
    g_cached = 0;
}
}  // namespace

static hipError_t pool_malloc_impl(void **p, size_t bytes);
// KVX_DBG_POISON=1: every block handed out is filled with 0xFF bytes (NaN as doubles, -1 as integers) -- a read of memory the
// library never wrote shows up in the results whatever the allocator happened to leave there
hipError_t pool_malloc(void **p, size_t bytes)
{
    static const bool poison = [] { const char *e = getenv("KVX_DBG_POISON"); return e && e[0] == '1'; }();
    hipError_t e = pool_malloc_impl(p, bytes);
    if (e == hipSuccess && poison) {
        (void)hipDeviceSynchronize();
        e = hipMemset(*p, 0xFF, std::max<size_t>((bytes + 255) & ~(size_t)255, 256));
        (void)hipDeviceSynchronize();
    }
    return e;
}

static hipError_t pool_malloc_impl(void **p, size_t bytes)
{
    if (!p) return hipErrorInvalidValue;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    int dev = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 != hipSuccess) return e0;
    std::lock_guard<std::mutex> lk(g_mu);
    if (bytes <= MAX_BLOCK) {
        auto it = g_free.lower_bound(Key{dev, bytes});
        if (it != g_free.end() && it->first.dev == dev && it->first.bytes <= bytes + bytes / 8) {
            *p = it->second;
            g_live[*p] = it->first;
            g_cached -= it->first.bytes;
            g_free.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && !g_free.empty()) {       // out of memory with blocks cached: give them back and retry
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        release_all_locked();
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) g_live[*p] = Key{dev, bytes};
    return e;
}

hipError_t pool_free(void *p)
{
    if (!p) return hipSuccess;
    std::unique_lock<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) { lk.unlock(); return hipFree(p); }      // not ours (never happens inside the library)
    const Key key = it->second;
    const size_t bytes = key.bytes;
    g_live.erase(it);
    if (bytes > MAX_BLOCK || cache_cap() == 0 || g_cached + bytes > cache_cap()) { lk.unlock(); return hipFree(p); }
    lk.unlock();
    hipError_t e;
    {                                               // what hipFree guarantees: nothing in flight on the BLOCK's device still touches it
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != key.dev) (void)hipSetDevice(key.dev);
        e = hipDeviceSynchronize();
        if (cur != key.dev) (void)hipSetDevice(cur);
    }
    lk.lock();
    g_free.emplace(key, p);
    g_cached += bytes;
    return e;
}

// high = true: a stream of the highest priority the device offers (the pivot chain of a factorisation: its kernels are
// dispatched ahead of the side streams' when both have workgroups waiting for a CU); kept in a pool of its own
hipError_t pool_stream_get(hipStream_t *s, bool high)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &pool = high ? g_streams_hi : g_streams;
        auto it = pool.find(dev);
        if (it != pool.end()) { *s = it->second; pool.erase(it); return hipSuccess; }
    }
    if (!high) start_refill(dev);
    if (high) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

void pool_stream_put(hipStream_t s, bool high)
{
    if (!s) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipStreamDestroy(s); return; }
    std::lock_guard<std::mutex> lk(g_mu);
    auto &pool = high ? g_streams_hi : g_streams;
    if (pool.size() >= 64) { (void)hipStreamDestroy(s); return; }
    pool.emplace(dev, s);
}

hipError_t pool_event_get(hipEvent_t *ev, bool timing)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &m = g_events[timing ? 1 : 0];
        auto it = m.find(dev);
        if (it != m.end()) { *ev = it->second; m.erase(it); return hipSuccess; }
    }
    return timing ? hipEventCreate(ev) : hipEventCreateWithFlags(ev, hipEventDisableTiming);
}

void pool_event_put(hipEvent_t ev, bool timing)
{
    if (!ev) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipEventDestroy(ev); return; }
    std::lock_guard<std::mutex> lk(g_mu);
    auto &m = g_events[timing ? 1 : 0];
    if (m.size() >= 4096) { (void)hipEventDestroy(ev); return; }
    m.emplace(dev, ev);
}

void pool_release_all()
{
    join_refill();
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_mu);
    release_all_locked();
    for (auto &kv : g_streams) (void)hipStreamDestroy(kv.second);
    g_streams.clear();
    for (auto &kv : g_streams_hi) (void)hipStreamDestroy(kv.second);
    g_streams_hi.clear();
    for (auto &m : g_events) { for (auto &kv : m) (void)hipEventDestroy(kv.second); m.clear(); }
}
}  // namespace kvx
