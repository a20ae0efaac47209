// The host side of dlc_host_to_device / dlc_device_to_host / dlc_set_host_threads: the pinned staging ring's bookkeeping and
// the pool of copy threads.  A HEADER so that it builds twice: into libdlc_hip.so (host_staging.hip, after dlc_internal.h)
// and, with plain g++ -fsanitize=thread / address,undefined, against tests/host_sanitize/hip_stub.h -- a stand-in for the
// dozen HIP calls used here whose "DMA engine" is a thread per stream (tests/test_host_sanitize_cpu.py; `make host-sanitize`).
// Whoever includes it has declared: hipError_t / hipStream_t / hipEvent_t and the hip* functions below, dlc_ctx with the
// fields {device, staging, host_threads, host_lock}, dlc::fail, dlc::DeviceGuard, DLC_HIP_CHECK and the status codes.
//
// Threading contract (include/dlc.h): one staged transfer per context at a time; calls from several threads SERIALISE on the
// context's host_lock -- which also covers the first call's creation of the ring and dlc_set_host_threads' teardown of it
// (r04 created the ring outside any lock and kept the lock inside the object being torn down: two first callers, or a
// transfer racing a thread-count change, were undefined; found by reading, confirmed by the sanitizer build).
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#ifndef DLC_STAGE_BYTES
#define DLC_STAGE_BYTES (16ull << 20)          // one piece: 16 MiB (a few hundred microseconds of DMA)
#endif
#ifndef DLC_POOL_MIN_BYTES
#define DLC_POOL_MIN_BYTES (1u << 20)          // below this one thread copies: waking the pool costs more
#endif

namespace dlc_hs {

constexpr size_t STAGE_BYTES = DLC_STAGE_BYTES;
constexpr int STAGE_RING = 4;                    // pieces in flight per direction

// N worker threads that split one memcpy between them; run() returns when all slices are done.
class CopyPool {
public:
    explicit CopyPool(int n) : n_(n < 1 ? 1 : n) {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int threads() const { return n_; }
    void run(char* dst, const char* src, size_t bytes) {
        if (n_ == 1 || bytes < DLC_POOL_MIN_BYTES) { memcpy(dst, src, bytes); return; }
        {
            std::lock_guard<std::mutex> lk(m_);
            dst_ = dst; src_ = src; bytes_ = bytes;
            pending_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        slice(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void slice(int t) {
        const size_t per = ((bytes_ + n_ - 1) / n_ + 4095) & ~(size_t)4095;
        const size_t lo = per * t;
        if (lo >= bytes_) return;
        const size_t len = bytes_ - lo < per ? bytes_ - lo : per;
        memcpy(dst_ + lo, src_ + lo, len);
    }
    void loop(int t) {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            slice(t);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0;
    int pending_ = 0;
};

}  // namespace dlc_hs
using dlc_hs::CopyPool;
using dlc_hs::STAGE_BYTES;
using dlc_hs::STAGE_RING;

struct dlc_host_staging {
    char* up[STAGE_RING] = {};          // host -> device pieces
    char* down[STAGE_RING] = {};        // device -> host pieces
    hipEvent_t up_ev[STAGE_RING] = {};
    hipEvent_t down_ev[STAGE_RING] = {};
    bool up_busy[STAGE_RING] = {};
    CopyPool* pool = nullptr;
};

namespace dlc {

// (the caller holds the context's host lock)
static int staging_get(dlc_ctx* ctx, dlc_host_staging** out) {
    if (!ctx->staging) {
        dlc_host_staging* s = new (std::nothrow) dlc_host_staging;
        if (!s) return dlc::fail(ctx, DLC_ERR_HIP, "host staging: out of memory");
        for (int i = 0; i < STAGE_RING; ++i) {
            if (hipHostMalloc((void**)&s->up[i], STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void**)&s->down[i], STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&s->up_ev[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s->down_ev[i], hipEventDisableTiming) != hipSuccess) {
                dlc::staging_free(s);
                return dlc::fail(ctx, DLC_ERR_HIP, "host staging: cannot allocate %d pinned pieces of %zu bytes", 2 * STAGE_RING, STAGE_BYTES);
            }
        }
        int n = ctx->host_threads;
        if (n <= 0) {
            n = (int)std::thread::hardware_concurrency();
            n = n < 1 ? 1 : (n > 16 ? 16 : n);          // a one-GPU share of the host: 16 cores
        }
        s->pool = new (std::nothrow) CopyPool(n);
        if (!s->pool) { dlc::staging_free(s); return dlc::fail(ctx, DLC_ERR_HIP, "host staging: out of memory"); }
        ctx->staging = s;
    }
    *out = ctx->staging;
    return DLC_OK;
}

void staging_free(dlc_host_staging* s) {
    if (!s) return;
    delete s->pool;
    for (int i = 0; i < STAGE_RING; ++i) {
        if (s->up[i]) (void)hipHostFree(s->up[i]);
        if (s->down[i]) (void)hipHostFree(s->down[i]);
        if (s->up_ev[i]) (void)hipEventDestroy(s->up_ev[i]);
        if (s->down_ev[i]) (void)hipEventDestroy(s->down_ev[i]);
    }
    delete s;
}

}  // namespace dlc

extern "C" int dlc_set_host_threads(dlc_ctx* ctx, int threads) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (threads < 0 || threads > 256) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "set_host_threads: %d outside 0..256", threads);
    std::lock_guard<std::mutex> lk(*ctx->host_lock);   // (a transfer in another thread finishes first)
    if (ctx->staging) {                              // re-created with the new count on the next transfer
        dlc::DeviceGuard guard(ctx->device);
        (void)hipDeviceSynchronize();
        dlc::staging_free(ctx->staging);
        ctx->staging = nullptr;
    }
    ctx->host_threads = threads;
    return DLC_OK;
}

extern "C" int dlc_host_to_device(dlc_ctx* ctx, void* dst_device, const void* src_host, size_t bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!dst_device || !src_host) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "host_to_device: null pointer");
    if (bytes == 0) return DLC_OK;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    std::lock_guard<std::mutex> lk(*ctx->host_lock);
    dlc_host_staging* s;
    int rc = dlc::staging_get(ctx, &s);
    if (rc != DLC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    size_t off = 0;
    for (int k = 0; off < bytes; ++k, off += STAGE_BYTES) {
        const int b = k % STAGE_RING;
        const size_t len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
        if (s->up_busy[b]) DLC_HIP_CHECK(ctx, hipEventSynchronize(s->up_ev[b]));      // the DMA that last read this piece
        s->pool->run(s->up[b], (const char*)src_host + off, len);
        DLC_HIP_CHECK(ctx, hipMemcpyAsync((char*)dst_device + off, s->up[b], len, hipMemcpyHostToDevice, st));
        DLC_HIP_CHECK(ctx, hipEventRecord(s->up_ev[b], st));
        s->up_busy[b] = true;
    }
    return DLC_OK;          // src_host is consumed; the last pieces' DMAs are still in flight on `stream`
}

extern "C" int dlc_device_to_host(dlc_ctx* ctx, void* dst_host, const void* src_device, size_t bytes, void* stream) {
    if (!ctx) return DLC_ERR_BAD_ARG;
    if (!dst_host || !src_device) return dlc::fail(ctx, DLC_ERR_BAD_ARG, "device_to_host: null pointer");
    if (bytes == 0) return DLC_OK;
    dlc::DeviceGuard guard(ctx->device);
    if (!guard.ok) return dlc::fail(ctx, DLC_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    std::lock_guard<std::mutex> lk(*ctx->host_lock);
    dlc_host_staging* s;
    int rc = dlc::staging_get(ctx, &s);
    if (rc != DLC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const long long pieces = (long long)((bytes + STAGE_BYTES - 1) / STAGE_BYTES);
    // DMAs run STAGE_RING - 1 pieces ahead of the host copies out of the ring
    for (long long k = 0; k < pieces + STAGE_RING - 1; ++k) {
        if (k < pieces) {
            const int b = (int)(k % STAGE_RING);
            const size_t off = (size_t)k * STAGE_BYTES;
            const size_t len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
            DLC_HIP_CHECK(ctx, hipMemcpyAsync(s->down[b], (const char*)src_device + off, len, hipMemcpyDeviceToHost, st));
            DLC_HIP_CHECK(ctx, hipEventRecord(s->down_ev[b], st));
        }
        const long long j = k - (STAGE_RING - 1);
        if (j >= 0) {
            const int b = (int)(j % STAGE_RING);
            const size_t off = (size_t)j * STAGE_BYTES;
            const size_t len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
            DLC_HIP_CHECK(ctx, hipEventSynchronize(s->down_ev[b]));
            s->pool->run((char*)dst_host + off, s->down[b], len);
        }
    }
    return DLC_OK;          // dst_host is complete
}
