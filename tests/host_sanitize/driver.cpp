// Sanitizer driver of the library's host-side threads (csrc/host_staging_impl.h: copy pool, pinned staging ring, the
// context's host lock) -- VERDICT r04 #8.  Two caller threads push 10 000 round trips (host -> "device" -> host) of random
// sizes -- from one byte to six ring pieces and a bit -- through ONE context, each on its own stream, and compare every
// byte; a third thread keeps changing dlc_set_host_threads, which tears the ring and the pool down and has them rebuilt by
// the next transfer.  Built by `make -C deeploopcloser_amd/csrc host-sanitize` with -fsanitize=thread and with
// -fsanitize=address,undefined (pieces of 4 KiB, pool from 256 bytes: small transfers walk the whole ring).
#include "hip_stub.h"

#include "host_staging_impl.h"

#include <atomic>

int main(int argc, char** argv) {
    const int per_thread = argc > 1 ? atoi(argv[1]) : 5000;
    dlc_ctx ctx;
    memset(&ctx, 0, sizeof(ctx));
    ctx.host_lock = new std::mutex;
    ctx.host_threads = 3;
    std::atomic<bool> stop{false};
    std::atomic<long long> bad{0}, done{0};
    const size_t max_bytes = 6 * (size_t)DLC_STAGE_BYTES + 777;
    auto caller = [&](int id) {
        std::mt19937_64 rng(1000 + id);
        std::vector<unsigned char> src(max_bytes), back(max_bytes);
        unsigned char* dev = (unsigned char*)malloc(max_bytes);
        hipStream_t st = (hipStream_t)(uintptr_t)(0x100 + id);
        for (int it = 0; it < per_thread; ++it) {
            const unsigned pick = (unsigned)(rng() % 8);
            size_t n = pick == 0 ? 1 + rng() % 16 : (pick < 4 ? 1 + rng() % (2 * (size_t)DLC_STAGE_BYTES) : 1 + rng() % max_bytes);
            if (pick == 7) n = (1 + rng() % 6) * (size_t)DLC_STAGE_BYTES;                 // whole pieces exactly
            for (size_t i = 0; i < n; i += 1 + (n >> 6)) src[i] = (unsigned char)rng();
            src[0] = (unsigned char)it; src[n - 1] = (unsigned char)(it >> 8);
            if (dlc_host_to_device(&ctx, dev, src.data(), n, st) != DLC_OK) { ++bad; break; }
            src[n / 2] ^= 0xff;                                                            // consumed: the caller may overwrite at once
            if (dlc_device_to_host(&ctx, back.data(), dev, n, st) != DLC_OK) { ++bad; break; }
            src[n / 2] ^= 0xff;
            if (memcmp(src.data(), back.data(), n) != 0) ++bad;
            ++done;
        }
        hipDeviceSynchronize();
        free(dev);
    };
    std::thread t0(caller, 0), t1(caller, 1);
    std::thread knob([&] {
        int k = 0;
        while (!stop.load()) {
            static const int counts[] = {1, 4, 2, 0, 3, 7};
            if (dlc_set_host_threads(&ctx, counts[k++ % 6]) != DLC_OK) ++bad;
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
    });
    t0.join(); t1.join();
    stop = true;
    knob.join();
    if (dlc_set_host_threads(&ctx, 300) != DLC_ERR_BAD_ARG) ++bad;                        // (argument check, and the error string)
    hipDeviceSynchronize();
    if (ctx.staging) dlc::staging_free(ctx.staging);
    delete ctx.host_lock;
    hipstub::shutdown();
    printf("host_sanitize: %lld round trips, %lld bad\n", done.load(), bad.load());
    return bad.load() == 0 && done.load() == 2LL * per_thread ? 0 : 1;
}
