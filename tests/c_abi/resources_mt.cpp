// resources_mt.cpp -- TEST INFRASTRUCTURE.  csrc/host_resources.hpp (the device-keyed caches of the host layer) compiled against
// tests/c_abi/fake_hip.h: four "devices", eight host threads.  What a one-GPU box cannot exercise:
//   * DeviceGuard switches and restores the per-thread device, nested and across a throw;
//   * a cached block only ever goes back to a caller on ITS device; caps, LRU eviction and the bounded deferral hold per device;
//   * a second free of a cached block never reaches hipFree; an out-of-memory allocation empties the device's cache and retries,
//     and a refused one is reported as HipFail{oom};
//   * every device gets its own priority stream set, exactly once, also when eight threads ask at the same time;
//   * ANOFOX_HIP_DEVICES parsing; nothing leaks after the release calls;
//   * parallel_shares runs every share exactly once and joins what it started -- also when the host refuses threads (EAGAIN after 0, 1,
//     3 grants, injected through the ANOFOX_TEST_HOOKS switch of the header) and when the caller's own share throws.
// Built by tests/test_abi_cpu.py with g++ -fsanitize=thread (and once with address,undefined); exit code 0 = all checks passed.
#define ANOFOX_TEST_HOOKS 1
#include "fake_hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../anofox-forecast_amd/csrc/host_semantics.hpp"
using namespace anofox;
namespace {
#include "../../anofox-forecast_amd/csrc/host_resources.hpp"
}

static int failures = 0;
#define CHECK(cond)                                                                       \
    do {                                                                                  \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static int device_of(void *p)
{
    FakeHip &f = fake_hip();
    std::lock_guard<std::mutex> lock(f.mu);
    auto it = f.dev_allocs.find(p);
    return it == f.dev_allocs.end() ? -1 : it->second.first;
}
static size_t idle_bytes_of(int dev)
{
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    return c.idle_bytes[dev];
}
static size_t cap_of(int dev)
{
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    return c.cap.count(dev) ? c.cap[dev] : 0;
}

int main()
{
    FakeHip &F = fake_hip();
    const int G = F.n_devices;
    CHECK(G >= 2);

    // ---- DeviceGuard -------------------------------------------------------------------------------------
    CHECK(hipSetDevice(1) == hipSuccess);
    {
        DeviceGuard a(3 % G);
        int d = -1; hipGetDevice(&d); CHECK(d == 3 % G);
        { DeviceGuard b(0); hipGetDevice(&d); CHECK(d == 0); { DeviceGuard same(0); hipGetDevice(&d); CHECK(d == 0); } hipGetDevice(&d); CHECK(d == 0); }
        hipGetDevice(&d); CHECK(d == 3 % G);
        try { DeviceGuard c(0); throw std::runtime_error("x"); } catch (...) {}
        hipGetDevice(&d); CHECK(d == 3 % G);
    }
    { int d = -1; hipGetDevice(&d); CHECK(d == 1); }

    // ---- the caching allocator is keyed by device ---------------------------------------------------------
    hipSetDevice(0);
    void *a0 = dev_alloc_bytes(1 << 20);
    CHECK(device_of(a0) == 0);
    dev_free(a0, true);                                   // cached on device 0
    CHECK(idle_bytes_of(0) == dev_round(1 << 20));
    hipSetDevice(1);
    void *a1 = dev_alloc_bytes(1 << 20);
    CHECK(a1 != a0 && device_of(a1) == 1);                // the idle block of device 0 is NOT handed to a caller on device 1
    hipSetDevice(0);
    const long mallocs = F.n_malloc;
    void *a0b = dev_alloc_bytes(1 << 20);
    CHECK(a0b == a0 && F.n_malloc == mallocs);            // ... and comes back to device 0 without a hipMalloc
    // a block is freed on ITS device's books whatever the current device is
    hipSetDevice(2 % G);
    dev_free(a0b, true);
    dev_free(a1, true);
    CHECK(idle_bytes_of(0) == dev_round(1 << 20) && idle_bytes_of(1) == dev_round(1 << 20) && idle_bytes_of(2 % G) == (2 % G <= 1 ? dev_round(1 << 20) : 0));
    // a second free of a cached block never reaches hipFree
    const long frees = F.n_free;
    dev_free(a1, true);
    CHECK(F.n_free == frees && F.n_bad_free == 0);

    // ---- per-device cap, oldest first ---------------------------------------------------------------------
    hipSetDevice(3 % G);
    const int D = 3 % G;
    std::vector<void *> blocks;
    for (int i = 0; i < 6; i++) blocks.push_back(dev_alloc_bytes((size_t)4 << 20));          // 24 MB live on a 64 MB device (cap = a quarter = 16 MB)
    CHECK(cap_of(D) == F.capacity / 4);
    for (void *p : blocks) dev_free(p, true);
    CHECK(idle_bytes_of(D) <= cap_of(D) && idle_bytes_of(D) >= cap_of(D) - ((size_t)4 << 20));
    CHECK(device_of(blocks[0]) == -1 && device_of(blocks[5]) == D);                        // the oldest went back to the runtime, the newest is cached
    // a block larger than the cap is never cached
    void *big = dev_alloc_bytes((size_t)20 << 20);
    dev_free(big, true);
    CHECK(device_of(big) == -1);

    // ---- out of memory: the device's cache is emptied and the allocation retried; a refusal is HipFail{oom} ----------
    {
        std::vector<void *> live;
        const size_t idle_before = idle_bytes_of(D);
        CHECK(idle_before > 0);
        // 64 MB device: with idle_before cached, 56 MB more only fit once the cache has been flushed
        for (int i = 0; i < 7; i++) live.push_back(dev_alloc_bytes((size_t)8 << 20));
        CHECK(idle_bytes_of(D) == 0 && F.n_oom >= 1);
        bool refused = false, oom_flag = false;
        try { (void)dev_alloc_bytes((size_t)16 << 20); } catch (const HipFail &f) { refused = true; oom_flag = f.oom; }
        CHECK(refused && oom_flag);
        (void)hipGetLastError();
        for (void *p : live) dev_free(p, true);
    }

    // ---- deferred evictions stay under the cap ---------------------------------------------------------------
    {
        EvictionDeferral hold;
        std::vector<void *> live;
        for (int i = 0; i < 12; i++) live.push_back(dev_alloc_bytes((size_t)4 << 20));     // 48 MB
        for (void *p : live) dev_free(p, true);
        DevCache &c = dev_cache();
        std::lock_guard<std::mutex> lock(c.mu);
        CHECK(c.deferred_bytes <= c.cap[D]);
        CHECK(c.idle_bytes[D] <= c.cap[D]);
    }
    { DevCache &c = dev_cache(); std::lock_guard<std::mutex> lock(c.mu); CHECK(c.deferred.empty() && c.deferred_bytes == 0); }

    // ---- pinned cache -----------------------------------------------------------------------------------------
    void *h0 = pin_alloc_bytes(3 << 20);
    pin_free(h0);
    void *h1 = pin_alloc_bytes(3 << 20);
    CHECK(h1 == h0);
    pin_free(h1);
    pin_free(h1);                                          // second free: ignored
    CHECK(F.n_bad_free == 0);

    // ---- stream sets: one priority set per device, oldest idle set of the device first --------------------------
    setenv("GPU_MAX_HW_QUEUES", "16", 1);
    hipSetDevice(0);
    StreamSet *s0 = stream_set_take();
    StreamSet *s0b = stream_set_take();
    CHECK(s0->dev == 0 && s0->prio && !s0b->prio && s0->aux[0]->prio != 0 && s0->aux[7]->prio == 0 && s0->own->dev == 0);
    hipSetDevice(1);
    StreamSet *s1 = stream_set_take();
    CHECK(s1->dev == 1 && s1->prio && s1->aux[3]->dev == 1 && s1->ev_join[5]->dev == 1);
    stream_set_give(s0b); stream_set_give(s0); stream_set_give(s1);
    hipSetDevice(0);
    StreamSet *again = stream_set_take();
    CHECK(again == s0);                                    // the OLDEST idle set of this device
    hipSetDevice(1);
    StreamSet *again1 = stream_set_take();
    CHECK(again1 == s1);                                   // never another device's
    stream_set_give(again); stream_set_give(again1);

    // ---- device list ------------------------------------------------------------------------------------------
    {
        std::vector<int> v;
        CHECK(parse_device_list("0,2", v) && v.size() == 2 && v[1] == 2);
        CHECK(parse_device_list("all", v) && (int)v.size() == G);
        CHECK(!parse_device_list("0,99", v));
        CHECK(parse_device_list(" 1 , 0 ", v) && v.size() == 2 && v[0] == 1);
    }

    // ---- eight threads over the devices ---------------------------------------------------------------------------
    {
        std::atomic<int> wrong_device{0}, thread_fail{0};
        auto worker = [&](int t) {
            try {
                const int dev = t % G;
                if (hipSetDevice(dev) != hipSuccess) { thread_fail++; return; }
                std::mt19937 rng(1234 + t);
                std::vector<std::pair<void *, int>> mine;
                StreamSet *ss = stream_set_take();
                if (ss->dev != dev) wrong_device++;
                for (int it = 0; it < 3000; it++) {
                    const int what = rng() % 10;
                    if (what < 5) {
                        const size_t sz = (size_t)(1 + rng() % 6) << 18;                         // 0.25 .. 1.5 MB
                        int cur = dev;
                        if (what == 0) { cur = (dev + 1) % G; DeviceGuard g(cur); void *p = dev_alloc_bytes(sz); if (device_of(p) != cur) wrong_device++; mine.push_back({p, cur}); }
                        else { void *p = dev_alloc_bytes(sz); if (device_of(p) != dev) wrong_device++; mine.push_back({p, dev}); }
                    } else if (what < 9 && !mine.empty()) {
                        const size_t k = rng() % mine.size();
                        if (device_of(mine[k].first) != mine[k].second) wrong_device++;
                        dev_free(mine[k].first, (rng() & 1) != 0);
                        mine[k] = mine.back(); mine.pop_back();
                    } else if (what == 9) {
                        if (rng() % 8 == 0) { EvictionDeferral hold; void *p = dev_alloc_bytes((size_t)3 << 20); dev_free(p, true); }
                        else { void *h = pin_alloc_bytes((size_t)(1 + rng() % 3) << 20); pin_free(h); }
                    }
                    int d = -1; hipGetDevice(&d);
                    if (d != dev) wrong_device++;                                       // every guard restored this thread's device
                    if (mine.size() > 8) { dev_free(mine.back().first, true); mine.pop_back(); }
                }
                for (auto &m : mine) dev_free(m.first, true);
                stream_set_give(ss);
            } catch (const HipFail &f) {
                std::fprintf(stderr, "thread %d: HipFail %s\n", t, f.msg.c_str());
                thread_fail++;
            } catch (...) { thread_fail++; }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < 8; t++) pool.emplace_back(worker, t);
        for (auto &th : pool) th.join();
        CHECK(wrong_device == 0 && thread_fail == 0);
        for (int d = 0; d < G; d++) CHECK(idle_bytes_of(d) <= cap_of(d) || cap_of(d) == 0);
    }
    // eight threads asking for their first stream set of a fresh process state: still one priority set per device
    stream_pool_release_all();
    {
        std::vector<StreamSet *> got(8, nullptr);
        std::vector<std::thread> pool;
        for (int t = 0; t < 8; t++) pool.emplace_back([&, t] { hipSetDevice(t % G); got[t] = stream_set_take(); });
        for (auto &th : pool) th.join();
        std::map<int, int> prio_sets;
        for (StreamSet *s : got) { CHECK(s != nullptr); if (s && s->prio) prio_sets[s->dev]++; }
        for (int d = 0; d < G && d < 8; d++) { if (prio_sets[d] != 1) std::fprintf(stderr, "device %d: %d priority sets\n", d, prio_sets[d]); CHECK(prio_sets[d] == 1); }
        for (StreamSet *s : got) stream_set_give(s);
    }

    // ---- shares of a host loop on threads --------------------------------------------------------------------------
    for (int refuse : {-1, 0, 1, 3}) {
        parallel_shares_refuse_after() = refuse;
        for (unsigned n : {0u, 1u, 2u, 8u, 33u}) {
            std::vector<std::atomic<int>> hits(n ? n : 1);
            for (auto &h : hits) h = 0;
            std::atomic<int> off_thread{0};
            const auto me = std::this_thread::get_id();
            parallel_shares(n, [&](unsigned k) { hits[k]++; if (std::this_thread::get_id() != me) off_thread++; });
            for (unsigned k = 0; k < n; k++) CHECK(hits[k] == 1);
            if (refuse >= 0) CHECK(off_thread <= refuse);
            if (refuse < 0 && n > 1) CHECK(off_thread == (int)n - 1);
        }
        // the caller's share throws: the started threads are joined (a joinable std::thread's destructor would end the process)
        std::atomic<int> done{0};
        bool caught = false;
        try { parallel_shares(6, [&](unsigned k) { if (k == 0) throw std::runtime_error("share 0"); std::this_thread::sleep_for(std::chrono::milliseconds(5)); done++; }); }
        catch (const std::runtime_error &) { caught = true; }
        CHECK(caught && done == (refuse < 0 ? 5 : std::min(refuse, 5)));
    }
    parallel_shares_refuse_after() = -1;

    // ---- nothing leaks -------------------------------------------------------------------------------------------
    dev_cache_release_all();
    pin_cache_release_all();
    stream_pool_release_all();
    {
        std::lock_guard<std::mutex> lock(F.mu);
        if (!F.dev_allocs.empty()) std::fprintf(stderr, "leaked device blocks: %zu (first on device %d, %zu bytes); host %zu streams %zu events %zu\n", F.dev_allocs.size(), F.dev_allocs.begin()->second.first, F.dev_allocs.begin()->second.second, F.host_allocs.size(), F.streams.size(), F.events.size());
        CHECK(F.dev_allocs.empty() && F.host_allocs.empty() && F.streams.empty() && F.events.empty());
        for (auto &kv : F.used) CHECK(kv.second == 0);
    }
    CHECK(F.n_bad_free == 0 && F.n_bad_handle == 0);
    if (failures) { std::fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    std::printf("resources_mt: ok (%ld hipMalloc, %ld hipFree, %ld refused, %d devices)\n", (long)F.n_malloc, (long)F.n_free, (long)F.n_oom, G);
    return 0;
}
