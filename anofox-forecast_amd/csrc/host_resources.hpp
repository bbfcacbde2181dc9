// host_resources.hpp -- the device-keyed resources of the host layer, apart from everything that launches a kernel:
//   HipFail / HIPCHECK / LAUNCHCHECK, the environment knobs (Tunables, ProcessTunables), DeviceGuard,
//   the caching device allocator (DevCache: blocks keyed by (device, rounded size), LRU eviction under a per-device cap, deferred
//   evictions bounded by that cap), the pinned-host cache, the stream / event sets (one priority set per device), and the list of
//   devices the batch entry shards over.
// Included by host_api.hip INSIDE its anonymous namespace, after <hip/hip_runtime.h>.  It calls nothing of the HIP API but the ~20
// functions a test double can provide: tests/c_abi/fake_hip.h + tests/c_abi/resources_mt.cpp compile THIS file against a stubbed
// runtime with several "devices" and run it under ThreadSanitizer (the per-device caches, DeviceGuard and
// the pinned blocks keyed by device had only ever run on one device) -- test infrastructure, never the product.
#pragma once

constexpr int N_AUX_STREAMS = 32;   // one stream per candidate ETS spec (the hardware multiplexes them onto GPU_MAX_HW_QUEUES queues)

struct HipFail { std::string msg; bool oom = false; };      // oom: the device or pinned-host allocator refused (reported as ALLOCATION_ERROR)
void report_hip_failure(AnofoxError *out_error, const HipFail &f)
{
    if (f.oom) anofox::set_error(out_error, ALLOCATION_ERROR, "Allocation error: " + f.msg);      // error.rs:20-21, code 4
    else anofox::set_error(out_error, INTERNAL_ERROR, "Internal error: " + f.msg);
}
#define HIPCHECK(expr)                                                                             \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) throw HipFail{std::string(#expr) + ": " + hipGetErrorString(_e), _e == hipErrorOutOfMemory};    \
    } while (0)

// A launch that the runtime refuses (no code object for this GPU, LDS request over the limit, invalid grid) is only
// visible through hipGetLastError: without this check the outputs would keep whatever the buffers held.
#define LAUNCHCHECK(what)                                                                          \
    do {                                                                                           \
        hipError_t _e = hipGetLastError();                                                         \
        if (_e != hipSuccess) throw HipFail{std::string("kernel launch failed (") + (what) + "): " + hipGetErrorString(_e), false};    \
    } while (0)

// d_status of a usable series before any kernel has written it: a series a kernel never reached is reported as an
// INTERNAL_ERROR instead of a success with uninitialised forecasts
constexpr int32_t STATUS_NOT_COMPUTED = -1;

// Every environment variable the library reads, in ONE place (INTEGRATION.md "Environment" is the user-facing list).  The schedule
// knobs are read when a batch is created (tests and the sweep scripts under tools/ vary them between batches of one process);
// the process-wide ones (caches, devices, priority streams) once.  None of them changes a result: every schedule walks the same
// Nelder-Mead iterates (tests/test_gpu_parity.py::test_schedule_variants_are_bit_identical).
struct Tunables {
    std::vector<int> budgets{24, 24, 24, 24, 24, 24, 48, 48, 96, 96, 192, 1024};   // tune budgets: Nelder-Mead iterations per round (last = to completion)
    int seq_rounds = -1;        // tune seq_rounds: rounds run by the sequential driver (-1: decided from the live problems)
    int gather_cols = 0;        // tune gather_cols (tests): columns of the per-spec gather blocks (0: as many as fit 55 % of the device, at most ld)
    int gather = -1;            // tune gather: dense re-gather of the running problems between rounds (-1: on, block size by memory)
    int spec_below = 8192;      // tune spec_below[_md]: four lanes per problem once this few problems of a spec still run
    int spec_below_md = 8192;   //   (_MD: the damped multiplicative-trend specs, whose pass is ~10x longer)
    int spec2_below = 1024;     // tune spec2_below[_md]: one wave per problem, two iterations per pass, for the last problems
    int spec2_below_md = 2048;  //   (tools/spec2_sweep.sh: 0 / 256 / 1024 / 2048 / 4096 / 8192 -> 579 / 575 / 565 / 562 / 594 / 736 ms on the 30-spec M5 batch)
    int k4 = -1;                // tune k4: additive-class specs run one lane per problem with all four trial points of an iteration in ONE
    int k4_top = 1, k4_top_below = 20480;   // tune k4_top, k4_top_below: see host_api.hip run_fit (0 = like every other spec)
                                //   pass (ets_fit_kernel.hpp K4) wherever one or four LANES per problem would run.  -1 (default): when the run is
                                //   memory bound and fills the chip that way -- the general-class specs see under half of the series AND the
                                //   additive specs have >= 131,072 live problems together (two K4 waves per SIMD), or (round 5) the batch has
                                //   >= 524,288 live problems over all its specs; 0 never; 1 always.
                                //   Measured: intermittent M5 batch 86.3 -> 70.4 ms, 125k x 1,024 batch 181 -> 136-140 ms, 1M x 1,024 1,200 -> 853 ms;
                                //   beside the 19 general-class specs of the strictly positive batch it is time-neutral (fewer passes, the step is
                                //   bound by fp64 issue), and ONE additive spec on 30,490 series is slower with it (20.5 against 17.9 ms: 477 waves)
    std::string wave_trace;     // tune wave_trace=<file>: developer instrument -- one record per wave of the ETS round kernels (FitArgs::wave_trace), written by anofox_hip_batch_lane_stats
    int compact = -1;           // tune compact: compact storage of the streamed block (host_api.hip compact_storage_begin): -1 auto, 0 never, 1 float at most, 2 narrowest exact type whatever the batch size
    // batches that do NOT oversubscribe the chip one lane per problem (launch_fit_slots, round 6): speculation is dealt by expected work
    int fill_policy = 1;        // tune fill_policy: 0 = rounds 1-5 (every spec four lanes per problem from the first round), 1 = by chip fill
    int fill_target = 150;      // tune fill_target: lanes the first round may occupy, in percent of the resident lanes (2 waves x 1,024 SIMDs x 64)
    int fill_late_div = 2;      // tune fill_late_div: a spec that started one lane per problem switches to four lanes at 1 / this of its problems
    int fill_s2_div = 16;       // tune fill_s2_div: ... and any spec to one wave per problem at 1 / this of its problems (at most spec2_below)
    int top_boost_n = 0, top_boost_pct = 200;   // tune top_boost_n / top_boost_pct: the N heaviest chains switch to four lanes / one wave per problem at pct % of the class thresholds
    int prio_top = 0;           // tune prio_top: the N chains with the most expected work run their waves at raised issue priority (3, 2, 1, 1, ...: launch_fit_slots)
    int dm_head_rounds = 0;     // tune dm_head_rounds: rounds the damped multiplicative-trend chains run before the other specs' streams start (launch_fit_slots)
    bool merge_periods = true;  // tune merge_periods: auto-detected periods run as merged batches (0: one batch per period)
    int part_threads = 16;      // tune part_threads: host threads that run the small per-period parts of an auto-detected batch side by side
    int pack_threads = 0;       // ANOFOX_HIP_PACK_THREADS: host threads of the packer (0: all, at most 32)
    bool timing = false;        // ANOFOX_HIP_TIMING: phase times of the batch entry on stderr
    int arima_trace = 0;        // tune arima_trace: 1 = per-sweep queue lengths / per-wave refit timings on stderr, 2 = also passes per variant (atomics: slow)
    double arima_lookahead = 6.0;    // tune arima_lookahead, _lookahead_depth, _spec_factor: see arima.hip launch_arima
    int arima_lookahead_depth = 2;
    double arima_spec_factor = 4.0;
    double arima_shared_chunk_rounds = 1.0;   // tune arima_shared_chunk_rounds: see arima.hip launch_arima
    int arima_prep_lanes = 64;       // tune arima_prep_lanes: series per wave of the AutoARIMA prep kernel (1..64)
    int arima_queue_sort = 1;        // tune arima_queue_sort: see arima.hip ar_bucket
    int arima_refit_budget = 100;   // tune arima_refit_budget: evaluations per series before the exact-likelihood refit's speculative launch takes over (0: off)
    // ANOFOX_HIP_TUNE="key=value;key=value": the ONE developer / test variable behind every schedule knob above (keys: the field names;
    // budgets is a comma list) -- the sweep scripts under tools/ and the schedule-variant tests use it; a deployment never sets it.
    // The variables a deployment may set are ANOFOX_HIP_DEVICES, _COALESCE_US, _CACHE_GB, _PINNED_CACHE_GB, _PACK_THREADS and _TIMING.
    static std::map<std::string, std::string> tune_map()
    {
        std::map<std::string, std::string> kv;
        const char *e = std::getenv("ANOFOX_HIP_TUNE");
        if (!e) return kv;
        std::string all(e);
        for (size_t pos = 0; pos < all.size();) {
            size_t end = all.find(';', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            const size_t eq = item.find('=');
            if (eq != std::string::npos) {
                std::string k = item.substr(0, eq), v = item.substr(eq + 1);
                while (!k.empty() && k.front() == ' ') k.erase(k.begin());
                while (!k.empty() && k.back() == ' ') k.pop_back();
                kv[k] = v;
            }
            pos = end + 1;
        }
        return kv;
    }
    static Tunables from_env()
    {
        Tunables t;
        const std::map<std::string, std::string> kv = tune_map();
        auto geti = [&](const char *k, int &v) { auto it = kv.find(k); if (it != kv.end()) v = std::atoi(it->second.c_str()); };
        auto getd = [&](const char *k, double &v) { auto it = kv.find(k); if (it != kv.end()) v = std::atof(it->second.c_str()); };
        if (kv.count("budgets")) {
            std::vector<int> v;
            for (const char *q = kv.at("budgets").c_str(); *q;) {
                char *end = nullptr;
                long x = std::strtol(q, &end, 10);
                if (end == q) break;
                if (x > 0) v.push_back((int)x);
                q = (*end == ',') ? end + 1 : end;
            }
            if (!v.empty()) t.budgets = v;
        }
        if (kv.count("wave_trace")) t.wave_trace = kv.at("wave_trace");
        geti("compact", t.compact); geti("prio_top", t.prio_top); geti("top_boost_n", t.top_boost_n); geti("top_boost_pct", t.top_boost_pct);
        geti("fill_policy", t.fill_policy); geti("fill_target", t.fill_target); geti("fill_late_div", t.fill_late_div); geti("fill_s2_div", t.fill_s2_div);
        geti("seq_rounds", t.seq_rounds);
        geti("gather", t.gather);
        if (kv.count("spec_below")) t.spec_below = t.spec_below_md = std::atoi(kv.at("spec_below").c_str());
        geti("spec_below_md", t.spec_below_md);
        geti("gather_cols", t.gather_cols);
        geti("k4", t.k4); geti("dm_head_rounds", t.dm_head_rounds); geti("k4_top", t.k4_top); geti("k4_top_below", t.k4_top_below);
        if (kv.count("spec2_below")) t.spec2_below = t.spec2_below_md = std::atoi(kv.at("spec2_below").c_str());
        geti("spec2_below_md", t.spec2_below_md);
        if (kv.count("merge_periods")) t.merge_periods = std::atoi(kv.at("merge_periods").c_str()) != 0;
        geti("part_threads", t.part_threads);
        if (kv.count("arima_trace")) t.arima_trace = std::atoi(kv.at("arima_trace").c_str());
        getd("arima_lookahead", t.arima_lookahead);
        geti("arima_lookahead_depth", t.arima_lookahead_depth);
        getd("arima_spec_factor", t.arima_spec_factor);
        geti("arima_queue_sort", t.arima_queue_sort);
        geti("arima_prep_lanes", t.arima_prep_lanes);
        getd("arima_shared_chunk_rounds", t.arima_shared_chunk_rounds);
        geti("arima_refit_budget", t.arima_refit_budget);
        if (const char *e = std::getenv("ANOFOX_HIP_PACK_THREADS")) t.pack_threads = std::max(1, std::atoi(e));
        t.timing = std::getenv("ANOFOX_HIP_TIMING") != nullptr;
        return t;
    }
};
// process-wide settings, read once: ANOFOX_HIP_CACHE_GB (idle device blocks kept, default 1/4 of the device), ANOFOX_HIP_PINNED_CACHE_GB
// (idle pinned staging blocks, default 2), tune prio_streams (high-priority streams of the first stream set, default from
// GPU_MAX_HW_QUEUES), ANOFOX_HIP_DEVICES (devices the batch entry shards over, default: the caller's current device only)
struct ProcessTunables {
    double cache_gb = -1.0, pinned_cache_gb = 2.0;
    double coalesce_us = 1000.0;      // ANOFOX_HIP_COALESCE_US: how long the first of several concurrent anofox_ts_forecast calls waits for the others (0: never).
                                      // Measured with 8 C worker threads (tests/c_abi/concurrent.c, wall / calls): AutoETS 9.7 ms without, 4.0 ms at 200 us,
                                      // 2.3 ms at 1,000 us; SES 5.0 / 0.87 / 0.05 ms -- the leader leaves as soon as its peers have joined, so the
                                      // window is only ever waited out when a peer has stopped calling
    int prio_streams = -1;
    std::string devices;
    static const ProcessTunables &get()
    {
        static const ProcessTunables t = [] {
            ProcessTunables p;
            if (const char *e = std::getenv("ANOFOX_HIP_CACHE_GB")) p.cache_gb = std::atof(e);
            if (const char *e = std::getenv("ANOFOX_HIP_PINNED_CACHE_GB")) p.pinned_cache_gb = std::atof(e);
            { const auto kv = Tunables::tune_map(); auto it = kv.find("prio_streams"); if (it != kv.end()) p.prio_streams = std::atoi(it->second.c_str()); }
            if (const char *e = std::getenv("ANOFOX_HIP_COALESCE_US")) p.coalesce_us = std::atof(e);
            if (const char *e = std::getenv("ANOFOX_HIP_DEVICES")) p.devices = e;
            return p;
        }();
        return t;
    }
};

// RAII: make `dev` current for a scope (the device is a per-thread setting)
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// Device memory comes from a small caching allocator: a batch of the M5 shape is ~300 hipMalloc calls (25 spec chains x state,
// maps and two gather blocks) = 380-400 ms to create and 50-65 ms to destroy, as long as its fit on the intermittent batch four
// times over; a statement that forecasts chunk after chunk of the same shape pays that once.  Blocks are keyed by (device, size
// rounded to 512 B / 2 MiB) and handed back as they are -- nothing in the library relies on fresh memory being zero.  Per device
// at most ANOFOX_HIP_CACHE_GB (default: a quarter of the device -- an M5-shape AutoETS batch is ~14 GB) stays cached: a block
// handed back over the cap evicts the OLDEST idle blocks of its device first, so shapes that are no longer used age out instead
// of pinning the cache; an out-of-memory hipMalloc empties the cache and retries; anofox_hip_release_caches() (header block 2)
// gives everything back on request -- a co-resident allocator (torch's, another library's) cannot reach these blocks otherwise.
struct DevCache {
    struct Idle { int dev; size_t size; void *ptr; };
    std::mutex mu;
    std::map<uint64_t, Idle> by_age;                                   // idle blocks, oldest first
    std::multimap<std::pair<int, size_t>, uint64_t> by_key;            // (device, size) -> age
    std::unordered_map<void *, uint64_t> idle_ptr;                     // guards against a second free of a cached block
    std::unordered_map<void *, std::pair<int, size_t>> live;
    std::map<int, size_t> idle_bytes, cap;                             // per device
    uint64_t next_age = 0;
    int defer = 0;                                                     // > 0: evicted blocks wait in `deferred` (EvictionDeferral below)
    std::vector<void *> deferred;
    std::unordered_map<void *, size_t> dropped_size;                   // bytes of the blocks on their way out (evicted / not cacheable)
    size_t deferred_bytes = 0;                                         // ... parked in `deferred`: never more than a device's cache cap
};
DevCache &dev_cache() { static DevCache *c = new DevCache; return *c; }     // never destroyed: no HIP calls at process exit

// hipFree waits for the whole device.  A call that keeps several batches running from several host threads (auto-detected periods:
// the merged batches beside the per-period ones) would stall a finishing thread in its evictions for as long as the other threads'
// kernels run -- 0.5-1.1 s per destroy measured -- so inside such a call evicted blocks are parked and freed when the call ends
// (nothing is running then), or when an allocation needs the memory.  The counter is process wide, so with steadily overlapping
// callers it may never reach zero: parked memory is therefore BOUNDED by the cache cap (ANOFOX_HIP_CACHE_GB) -- a block that would
// take the parked set over it is freed on the spot together with everything parked (the price is one stalled destroy, not an
// unbounded hold on HBM that no co-resident allocator could reclaim).
void dev_cache_free_blocks(std::vector<void *> &drop)
{
    if (drop.empty()) return;
    DevCache &c = dev_cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        size_t bytes = 0, cap = 0;
        for (void *q : drop) { auto it = c.dropped_size.find(q); if (it != c.dropped_size.end()) bytes += it->second; }
        for (auto &kv : c.cap) cap = std::max(cap, kv.second);
        if (c.defer > 0 && c.deferred_bytes + bytes <= cap) {
            c.deferred.insert(c.deferred.end(), drop.begin(), drop.end());
            c.deferred_bytes += bytes;
            drop.clear();
            return;
        }
        if (c.defer > 0) { drop.insert(drop.end(), c.deferred.begin(), c.deferred.end()); c.deferred.clear(); c.deferred_bytes = 0; }
        for (void *q : drop) c.dropped_size.erase(q);
    }
    for (void *q : drop) (void)hipFree(q);
    drop.clear();
}
void dev_cache_flush_deferred()
{
    DevCache &c = dev_cache();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        drop.swap(c.deferred);
        c.deferred_bytes = 0;
        for (void *q : drop) c.dropped_size.erase(q);
    }
    for (void *q : drop) (void)hipFree(q);
}
struct EvictionDeferral {
    EvictionDeferral() { DevCache &c = dev_cache(); std::lock_guard<std::mutex> lock(c.mu); c.defer++; }
    ~EvictionDeferral()
    {
        DevCache &c = dev_cache();
        bool last;
        { std::lock_guard<std::mutex> lock(c.mu); last = --c.defer == 0; }
        if (last) dev_cache_flush_deferred();
    }
    EvictionDeferral(const EvictionDeferral &) = delete;
    EvictionDeferral &operator=(const EvictionDeferral &) = delete;
};

size_t dev_round(size_t bytes)
{
    const size_t g = bytes < (1u << 20) ? 512 : (2u << 20);
    return (std::max<size_t>(bytes, 1) + g - 1) / g * g;
}

// (lock held) take idle blocks of `dev` out of the cache, oldest first, until `need` more bytes fit under the cap (all of them
// when need == SIZE_MAX); the caller frees them outside the lock
void dev_cache_evict_locked(DevCache &c, int dev, size_t need, std::vector<void *> &drop)
{
    for (auto it = c.by_age.begin(); it != c.by_age.end();) {
        if (need != SIZE_MAX && c.idle_bytes[dev] + need <= c.cap[dev]) break;
        if (it->second.dev != dev) { ++it; continue; }
        const DevCache::Idle b = it->second;
        auto range = c.by_key.equal_range({b.dev, b.size});
        for (auto k = range.first; k != range.second; ++k) if (k->second == it->first) { c.by_key.erase(k); break; }
        c.idle_ptr.erase(b.ptr);
        c.idle_bytes[dev] -= b.size;
        drop.push_back(b.ptr);
        c.dropped_size[b.ptr] = b.size;
        it = c.by_age.erase(it);
    }
}

void *dev_alloc_bytes(size_t bytes)
{
    DevCache &c = dev_cache();
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    const size_t sz = dev_round(bytes);
    {
        std::lock_guard<std::mutex> lock(c.mu);
        if (!c.cap.count(dev)) {
            size_t free_b = 0, total_b = 0;
            // a quarter of the device: an eighth was tried (ADVICE round 2 asked for a smaller default) and the auto-detected M5 batch
            // -- six large batches alive at once, ~40 GB handed back within a second -- then spent 0.8-1.1 s per destroy in hipFree,
            // which waits for the whole device while the other host threads' batches are running (3.3 s per call)
            c.cap[dev] = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? total_b / 4 : 0;
            const double gb = ProcessTunables::get().cache_gb;
            if (gb >= 0.0) c.cap[dev] = (size_t)(gb * 1073741824.0);
        }
        auto it = c.by_key.find({dev, sz});
        if (it != c.by_key.end()) {
            const uint64_t age = it->second;
            void *p = c.by_age[age].ptr;
            c.by_age.erase(age);
            c.by_key.erase(it);
            c.idle_ptr.erase(p);
            c.idle_bytes[dev] -= sz;
            c.live[p] = {dev, sz};
            return p;
        }
    }
    void *p = nullptr;
    hipError_t err = hipMalloc(&p, sz);
    if (err == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(c.mu);
            dev_cache_evict_locked(c, dev, SIZE_MAX, drop);
            for (void *q : drop) c.dropped_size.erase(q);
        }
        for (void *q : drop) (void)hipFree(q);
        dev_cache_flush_deferred();
        err = hipMalloc(&p, sz);
    }
    HIPCHECK(err);
    std::lock_guard<std::mutex> lock(c.mu);
    c.live[p] = {dev, sz};
    return p;
}

// `quiesced`: the caller has already waited for every stream that may touch the block (batch destruction); otherwise the block's
// device is synchronised first, which is what hipFree does implicitly
void dev_free(void *p, bool quiesced = false)
{
    if (!p) return;
    DevCache &c = dev_cache();
    int dev = -1;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) dev = it->second.first;
        else if (c.idle_ptr.count(p)) return;          // already handed back: a second free must not reach hipFree while the block sits in the cache
    }
    if (dev < 0) { (void)hipFree(p); return; }         // not one of ours
    if (!quiesced) { DeviceGuard g(dev); (void)hipDeviceSynchronize(); }
    std::vector<void *> drop;
    bool cached = false;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.live.find(p);
        if (it == c.live.end()) return;
        const size_t sz = it->second.second;
        c.live.erase(it);
        if (sz <= c.cap[dev]) {
            dev_cache_evict_locked(c, dev, sz, drop);
            const uint64_t age = c.next_age++;
            c.by_age[age] = DevCache::Idle{dev, sz, p};
            c.by_key.insert({{dev, sz}, age});
            c.idle_ptr[p] = age;
            c.idle_bytes[dev] += sz;
            cached = true;
        } else c.dropped_size[p] = sz;
    }
    if (!cached) drop.push_back(p);
    dev_cache_free_blocks(drop);
}

void dev_cache_release_all()
{
    DevCache &c = dev_cache();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (auto &kv : c.by_age) drop.push_back(kv.second.ptr);
        c.by_age.clear(); c.by_key.clear(); c.idle_ptr.clear();
        for (auto &kv : c.idle_bytes) kv.second = 0;
    }
    for (void *q : drop) (void)hipFree(q);
    dev_cache_flush_deferred();
}

template <class T> T *dalloc(size_t n) { return (T *)dev_alloc_bytes(std::max<size_t>(n, 1) * sizeof(T)); }

// ... and the pinned staging blocks of the host packer (page-locking 467 MB is ~20 ms, unlocking it ~40 ms): the same scheme,
// sizes rounded to 2 MiB, at most ANOFOX_HIP_PINNED_CACHE_GB (default 2) kept, oldest evicted first
struct PinCache {
    std::mutex mu;
    std::map<uint64_t, std::pair<size_t, void *>> by_age;
    std::multimap<size_t, uint64_t> by_size;
    std::unordered_map<void *, size_t> live;
    size_t idle_bytes = 0;
    uint64_t next_age = 0;
};
PinCache &pin_cache() { static PinCache *c = new PinCache; return *c; }
void *pin_alloc_bytes(size_t bytes)
{
    const size_t sz = (std::max<size_t>(bytes, 1) + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    PinCache &c = pin_cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.by_size.find(sz);
        if (it != c.by_size.end()) {
            void *p = c.by_age[it->second].second;
            c.by_age.erase(it->second);
            c.by_size.erase(it);
            c.idle_bytes -= sz;
            c.live[p] = sz;
            return p;
        }
    }
    void *p = nullptr;
    HIPCHECK(hipHostMalloc(&p, sz, hipHostMallocDefault));
    std::lock_guard<std::mutex> lock(c.mu);
    c.live[p] = sz;
    return p;
}
void pin_free(void *p)
{
    if (!p) return;
    const size_t cap = (size_t)(std::max(0.0, ProcessTunables::get().pinned_cache_gb) * 1073741824.0);
    PinCache &c = pin_cache();
    std::vector<void *> drop;
    bool cached = false;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.live.find(p);
        if (it == c.live.end()) return;                 // not live: already handed back (or never ours)
        const size_t sz = it->second;
        c.live.erase(it);
        if (sz <= cap) {
            while (c.idle_bytes + sz > cap && !c.by_age.empty()) {
                auto old = c.by_age.begin();
                auto range = c.by_size.equal_range(old->second.first);
                for (auto k = range.first; k != range.second; ++k) if (k->second == old->first) { c.by_size.erase(k); break; }
                c.idle_bytes -= old->second.first;
                drop.push_back(old->second.second);
                c.by_age.erase(old);
            }
            const uint64_t age = c.next_age++;
            c.by_age[age] = {sz, p};
            c.by_size.insert({sz, age});
            c.idle_bytes += sz;
            cached = true;
        }
    }
    for (void *q : drop) (void)hipHostFree(q);
    if (!cached) (void)hipHostFree(p);
}
void pin_cache_release_all()
{
    PinCache &c = pin_cache();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (auto &kv : c.by_age) drop.push_back(kv.second.second);
        c.by_age.clear(); c.by_size.clear(); c.idle_bytes = 0;
    }
    for (void *q : drop) (void)hipHostFree(q);
}

// ... and the streams and events of a batch: 33 streams + 37 events are ~10 ms to create, and -- measured -- the streams a
// process creates FIRST get the better mapping onto the 16 hardware queues: the same 30-spec batch runs in 580 ms on the first
// batch of a process and in 690-700 ms on every batch created after that one was destroyed (tools/time_run_variants.py).  A
// batch borrows a set and hands it back (synchronised) when it is destroyed or parked in the single-series pool; idle sets are
// destroyed only by anofox_hip_release_caches().
struct StreamSet {
    int dev = 0;
    unsigned long id = 0;              // creation order
    bool prio = false;                 // holds the process's high-priority streams
    hipStream_t own = nullptr, aux[N_AUX_STREAMS] = {};
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_fit0 = nullptr, ev_fit1 = nullptr, ev_fork = nullptr, ev_join[N_AUX_STREAMS] = {};
};
struct StreamPool { std::mutex mu; std::vector<StreamSet *> idle; unsigned long created = 0; std::map<int, bool> prio_taken; };
StreamPool &stream_pool() { static StreamPool *p = new StreamPool; return *p; }
void stream_set_destroy(StreamSet *s)
{
    DeviceGuard g(s->dev);
    if (s->own) (void)hipStreamDestroy(s->own);
    for (auto &q : s->aux) if (q) (void)hipStreamDestroy(q);
    for (hipEvent_t e : {s->ev_start, s->ev_stop, s->ev_fit0, s->ev_fit1, s->ev_fork}) if (e) (void)hipEventDestroy(e);
    for (auto &e : s->ev_join) if (e) (void)hipEventDestroy(e);
    delete s;
}
// The candidate specs of a fit run on up to 25 streams side by side; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
// (default 4) and streams that share a queue serialise.  The variable is read when the runtime initialises, and the library does not
// touch the process environment (a setenv at load time races with getenv in a multi-threaded host such as DuckDB and changes every
// other HIP user of the process: INTEGRATION.md, Environment) -- the HOST exports it.  What the library does: the first stream set
// looks at what the environment says and, when the value is missing or below 16, says so ONCE on stderr -- only when the host has
// asked for the library's diagnostics (ANOFOX_HIP_TIMING; round 6: an embedding host such as DuckDB must not get library noise on its
// stderr at every process start).  Results are unaffected, the 25-spec AutoETS batch is 1.3-2x slower on 4 queues;
// INTEGRATION.md is the primary place that says so.
inline void warn_hw_queues_once()
{
    static std::once_flag once;
    std::call_once(once, [] {
        const char *q = std::getenv("GPU_MAX_HW_QUEUES");
        const int v = q ? std::atoi(q) : 0;
        if (v < 16 && std::getenv("ANOFOX_HIP_TIMING") != nullptr)
            std::fprintf(stderr, "[anofox-hip] warning: GPU_MAX_HW_QUEUES is %s (< 16): the candidate ETS specs run on concurrent HIP streams and will share "
                                 "hardware queues -- export GPU_MAX_HW_QUEUES=16 in the host's environment before its first HIP call (results are "
                                 "unaffected; the 25-spec AutoETS batch is about twice as slow)\n", q ? q : "unset");
    });
}
StreamSet *stream_set_take()
{
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    warn_hw_queues_once();
    StreamPool &p = stream_pool();
    // The first streams of a device's FIRST set carry the most expensive specs of a fit (launch_fit_slots orders the specs by
    // work) and get the highest priority: the command processor then dispatches their workgroups first whenever slots free up, the
    // cheap specs fill in behind -- longest chains first: 571 -> 536-545 ms on the 30-spec M5 batch, neutral elsewhere.  Every
    // priority level has its own hardware queues and the chip multiplexes well only up to ~23 of them in total (16 normal + 7
    // high: 541 ms, + 8: 747 ms; 13 + 10, 14 + 9, 15 + 8: 544 ms; 20 + 7: 785 ms), so the count follows GPU_MAX_HW_QUEUES (none
    // when the host has not set it: the runtime's default of 4 queues leaves no room) and later sets (concurrent batches of other
    // host threads) stay at normal priority.  ANOFOX_HIP_TUNE prio_streams=N overrides the count.  The priority set is RESERVED under
    // the lock before any stream exists, so two threads creating their first sets at once cannot both take it.
    bool want_prio = false;
    {
        std::lock_guard<std::mutex> lock(p.mu);
        // the OLDEST idle set of this device first: that is the one with the favourable queue mapping
        long best = -1;
        for (size_t i = 0; i < p.idle.size(); i++)
            if (p.idle[i]->dev == dev && (best < 0 || p.idle[i]->id < p.idle[(size_t)best]->id)) best = (long)i;
        if (best >= 0) { StreamSet *s = p.idle[(size_t)best]; p.idle.erase(p.idle.begin() + best); return s; }
        if (!p.prio_taken[dev]) { p.prio_taken[dev] = true; want_prio = true; }
    }
    int n_prio = 0;
    if (want_prio) {
        const char *q = std::getenv("GPU_MAX_HW_QUEUES");
        n_prio = q ? std::max(0, std::min(7, 23 - std::atoi(q))) : 0;
        const int forced = ProcessTunables::get().prio_streams;
        if (forced >= 0) n_prio = std::min(forced, N_AUX_STREAMS);
    }
    StreamSet *s = new StreamSet;
    try {
        s->dev = dev;
        s->prio = want_prio;
        HIPCHECK(hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking));
        int prio_least = 0, prio_greatest = 0;
        if (n_prio > 0) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        for (int i = 0; i < N_AUX_STREAMS; i++) {
            if (i < n_prio) HIPCHECK(hipStreamCreateWithPriority(&s->aux[i], hipStreamNonBlocking, prio_greatest));
            else HIPCHECK(hipStreamCreateWithFlags(&s->aux[i], hipStreamNonBlocking));
        }
        for (hipEvent_t *e : {&s->ev_start, &s->ev_stop, &s->ev_fit0, &s->ev_fit1}) HIPCHECK(hipEventCreate(e));
        HIPCHECK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        for (auto &e : s->ev_join) HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    } catch (...) {
        stream_set_destroy(s);
        if (want_prio) { std::lock_guard<std::mutex> lock(p.mu); p.prio_taken[dev] = false; }
        throw;
    }
    { std::lock_guard<std::mutex> lock(p.mu); s->id = p.created++; }
    return s;
}
void stream_set_give(StreamSet *s)
{
    if (!s) return;
    StreamPool &p = stream_pool();
    std::lock_guard<std::mutex> lock(p.mu);
    p.idle.push_back(s);
}
void stream_pool_release_all()
{
    StreamPool &p = stream_pool();
    std::vector<StreamSet *> drop;
    {
        std::lock_guard<std::mutex> lock(p.mu);
        drop.swap(p.idle);
        for (StreamSet *s : drop) if (s->prio) p.prio_taken[s->dev] = false;     // the next set created on that device takes the priority streams again
    }
    for (StreamSet *s : drop) stream_set_destroy(s);
}

// ---------------------------------------------------------------------------------------------
// shares of a host loop on threads: share(k), k = 0 .. n - 1, each exactly once.  Share 0 runs on the calling thread; a share whose
// thread cannot be created (std::system_error: EAGAIN under the process's or the cgroup's thread limit -- DuckDB's own workers call
// this library concurrently) runs on the calling thread too, and every thread that did start is joined on every way out: a
// joinable std::thread destroyed by an exception on its way through the caller is std::terminate, i.e. the host process gone.
// `share` does its own error reporting (it runs on other threads: nothing may leave it).
// ---------------------------------------------------------------------------------------------
#ifdef ANOFOX_TEST_HOOKS
inline std::atomic<int> &parallel_shares_refuse_after() { static std::atomic<int> v{-1}; return v; }     // tests/c_abi/resources_mt.cpp: threads granted before EAGAIN
#endif
template <class F> void parallel_shares(unsigned n, F &&share)
{
    if (n <= 1) { if (n == 1) share(0u); return; }
    std::vector<std::thread> pool;
    struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{pool};
    pool.reserve(n - 1);
    unsigned next = 1;
    for (; next < n; next++) {
        try {
#ifdef ANOFOX_TEST_HOOKS
            if (parallel_shares_refuse_after() >= 0 && (int)pool.size() >= parallel_shares_refuse_after())
                throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
#endif
            pool.emplace_back([&share, next] { share(next); });
        } catch (const std::system_error &) { break; }
    }
    share(0u);
    for (; next < n; next++) share(next);          // the shares nobody could be started for
}

// ---------------------------------------------------------------------------------------------
// the devices the batch entry shards over (anofox_hip_set_devices / ANOFOX_HIP_DEVICES)
// ---------------------------------------------------------------------------------------------
struct DeviceList { std::mutex mu; bool env_read = false; std::vector<int> devs; size_t min_series = 2048; };
inline DeviceList &device_list() { static DeviceList *d = new DeviceList; return *d; }

// "0,1,2,3" / "all" -> ordinals; false when an entry is not a visible device
inline bool parse_device_list(const std::string &spec, std::vector<int> &out)
{
    out.clear();
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return spec.empty();
    if (spec == "all" || spec == "ALL") { for (int i = 0; i < cnt; i++) out.push_back(i); return true; }
    for (const char *q = spec.c_str(); *q;) {
        while (*q == ',' || *q == ' ') q++;
        if (!*q) break;
        char *end = nullptr;
        const long v = std::strtol(q, &end, 10);
        if (end == q || v < 0 || v >= cnt) return false;
        out.push_back((int)v);
        q = end;
    }
    return true;
}

inline std::vector<int> devices_in_use(size_t *min_series)
{
    DeviceList &d = device_list();
    std::lock_guard<std::mutex> lock(d.mu);
    if (!d.env_read) {
        d.env_read = true;
        const std::string &spec = ProcessTunables::get().devices;
        std::vector<int> v;
        if (!spec.empty() && parse_device_list(spec, v)) d.devs = v;
        else if (!spec.empty()) std::fprintf(stderr, "[anofox-hip] ANOFOX_HIP_DEVICES=%s names a device that is not visible: ignored\n", spec.c_str());
    }
    if (min_series) *min_series = d.min_series;
    return d.devs;
}

