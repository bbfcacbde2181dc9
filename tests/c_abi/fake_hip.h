/* fake_hip.h -- TEST INFRASTRUCTURE: a host-only stand-in for the ~20 HIP runtime functions csrc/host_resources.hpp calls.
 *
 * Several "devices" (FAKE_HIP_DEVICES, default 4), each with its own memory capacity (FAKE_HIP_DEVICE_MB, default 64): the current
 * device is a per-thread setting as in HIP, hipMalloc charges the CURRENT device and fails with hipErrorOutOfMemory beyond its
 * capacity, every stream / event / allocation remembers the device it was created on, and a registry catches what a real run on one
 * GPU can never show: a block handed to a caller of another device, a double hipFree, a stream destroyed twice, leaks.  Nothing
 * of this is ever linked into the product (the product includes <hip/hip_runtime.h>); tests/c_abi/resources_mt.cpp is the only user.
 */
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorInvalidHandle = 400 };
struct FakeStream { int dev; int prio; bool alive; };
struct FakeEvent { int dev; bool alive; };
typedef FakeStream *hipStream_t;
typedef FakeEvent *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };

struct FakeHip {
    std::mutex mu;
    int n_devices = 4;
    size_t capacity = 64u << 20;
    std::map<void *, std::pair<int, size_t>> dev_allocs;       // pointer -> (device, bytes)
    std::map<void *, size_t> host_allocs;
    std::map<int, size_t> used;                                // bytes in use per device
    std::set<FakeStream *> streams;
    std::set<FakeEvent *> events;
    std::atomic<long> n_malloc{0}, n_free{0}, n_oom{0}, n_bad_free{0}, n_bad_handle{0}, n_sync{0}, n_host_malloc{0}, n_host_free{0};
    FakeHip()
    {
        if (const char *e = std::getenv("FAKE_HIP_DEVICES")) n_devices = std::atoi(e);
        if (const char *e = std::getenv("FAKE_HIP_DEVICE_MB")) capacity = (size_t)std::atoi(e) << 20;
    }
};
inline FakeHip &fake_hip() { static FakeHip *f = new FakeHip; return *f; }
inline int &fake_cur_dev() { static thread_local int d = 0; return d; }
inline hipError_t &fake_last_error() { static thread_local hipError_t e = hipSuccess; return e; }

inline const char *hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorInvalidHandle: return "invalid resource handle";
    default: return "invalid value";
    }
}
inline hipError_t hipGetLastError() { hipError_t e = fake_last_error(); fake_last_error() = hipSuccess; return e; }
inline hipError_t hipGetDeviceCount(int *n) { *n = fake_hip().n_devices; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = fake_cur_dev(); return hipSuccess; }
inline hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= fake_hip().n_devices) return fake_last_error() = hipErrorInvalidDevice;
    fake_cur_dev() = d;
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() { fake_hip().n_sync++; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b)
{
    FakeHip &f = fake_hip();
    std::lock_guard<std::mutex> lock(f.mu);
    *total_b = f.capacity;
    *free_b = f.capacity - f.used[fake_cur_dev()];
    return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t bytes)
{
    FakeHip &f = fake_hip();
    std::lock_guard<std::mutex> lock(f.mu);
    const int dev = fake_cur_dev();
    if (f.used[dev] + bytes > f.capacity) { f.n_oom++; *p = nullptr; return fake_last_error() = hipErrorOutOfMemory; }
    *p = std::malloc(bytes < 64 ? 64 : 64);          // a token: the test never touches "device" memory
    f.dev_allocs[*p] = {dev, bytes};
    f.used[dev] += bytes;
    f.n_malloc++;
    return hipSuccess;
}
inline hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    FakeHip &f = fake_hip();
    std::lock_guard<std::mutex> lock(f.mu);
    auto it = f.dev_allocs.find(p);
    if (it == f.dev_allocs.end()) { f.n_bad_free++; return fake_last_error() = hipErrorInvalidValue; }      // double free / not ours
    f.used[it->second.first] -= it->second.second;
    f.dev_allocs.erase(it);
    std::free(p);
    f.n_free++;
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned)
{
    FakeHip &f = fake_hip();
    *p = std::malloc(64);
    std::lock_guard<std::mutex> lock(f.mu);
    f.host_allocs[*p] = bytes;
    f.n_host_malloc++;
    return hipSuccess;
}
inline hipError_t hipHostFree(void *p)
{
    FakeHip &f = fake_hip();
    std::lock_guard<std::mutex> lock(f.mu);
    auto it = f.host_allocs.find(p);
    if (it == f.host_allocs.end()) { f.n_bad_free++; return fake_last_error() = hipErrorInvalidValue; }
    f.host_allocs.erase(it);
    std::free(p);
    f.n_host_free++;
    return hipSuccess;
}
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t fake_stream_create(hipStream_t *s, int prio)
{
    FakeHip &f = fake_hip();
    *s = new FakeStream{fake_cur_dev(), prio, true};
    std::lock_guard<std::mutex> lock(f.mu);
    f.streams.insert(*s);
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return fake_stream_create(s, 0); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int prio) { return fake_stream_create(s, prio); }
inline hipError_t hipStreamDestroy(hipStream_t s)
{
    FakeHip &f = fake_hip();
    std::lock_guard<std::mutex> lock(f.mu);
    if (!f.streams.erase(s)) { f.n_bad_handle++; return fake_last_error() = hipErrorInvalidHandle; }
    delete s;
    return hipSuccess;
}
inline hipError_t fake_event_create(hipEvent_t *e)
{
    FakeHip &f = fake_hip();
    *e = new FakeEvent{fake_cur_dev(), true};
    std::lock_guard<std::mutex> lock(f.mu);
    f.events.insert(*e);
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *e) { return fake_event_create(e); }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return fake_event_create(e); }
inline hipError_t hipEventDestroy(hipEvent_t e)
{
    FakeHip &f = fake_hip();
    std::lock_guard<std::mutex> lock(f.mu);
    if (!f.events.erase(e)) { f.n_bad_handle++; return fake_last_error() = hipErrorInvalidHandle; }
    delete e;
    return hipSuccess;
}
