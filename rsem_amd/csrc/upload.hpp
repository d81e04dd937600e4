// upload.hpp -- host -> device copies of the large immutable inputs (CSR, reads) through pinned staging buffers.
//
// The callers' arrays are ordinary pageable memory or read-only file mappings (host/rsb.hpp).  hipMemcpy from such memory
// stages through one internal buffer on one thread; here two pinned buffers alternate, several host threads fill one
// while the DMA engine drains the other, so the copy runs at about the slower of (parallel memcpy, PCIe) instead of a
// single core's memcpy rate.  When staged_h2d returns the source may be released; the device side is ordered on `st`.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "common.hpp"

namespace rsem {

struct Stager {
    static constexpr size_t kChunk = (size_t)128 << 20;
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    int device = -1;
    ~Stager() { release(); }
    void release() {
        for (int i = 0; i < 2; i++) {
            if (ev[i]) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
            if (buf[i]) { (void)hipHostFree(buf[i]); buf[i] = nullptr; }
            used[i] = false;
        }
        device = -1;
    }
    int prepare() {
        int dev = 0;
        RSEM_HIP_TRY(hipGetDevice(&dev));
        if (device == dev && buf[0]) return RSEM_OK;
        release();
        for (int i = 0; i < 2; i++) {
            RSEM_HIP_TRY(hipHostMalloc(&buf[i], kChunk, hipHostMallocDefault));
            RSEM_HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        device = dev;
        return RSEM_OK;
    }
};

inline Stager& thread_stager() {
    static thread_local Stager s;
    return s;
}

// (RSEM_HIP_TIMING=2: bytes and seconds of the staged copies, summed; rsem-run-em prints them with its marks)
struct StagedStats { std::atomic<uint64_t> bytes{0}, ns{0}, ns_fill{0}, ns_wait{0}; };
inline StagedStats& staged_stats() { static StagedStats s; return s; }

inline int staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return RSEM_OK;
    const auto t_all = std::chrono::steady_clock::now();
    struct Acc { std::chrono::steady_clock::time_point t0; size_t b; ~Acc() { staged_stats().bytes += b; staged_stats().ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } acc{t_all, bytes};
    if (bytes < ((size_t)16 << 20)) {  // small: the runtime's own path, completed before returning (the source may go away)
        RSEM_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        RSEM_HIP_TRY(hipStreamSynchronize(st));
        return RSEM_OK;
    }
    Stager& S = thread_stager();
    int rc = S.prepare();
    if (rc != RSEM_OK) return rc;
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthr = (int)std::min<unsigned>(16, std::max<unsigned>(1, hw / 4));
    size_t done = 0;
    int k = 0;
    while (done < bytes) {
        const size_t n = std::min(Stager::kChunk, bytes - done);
        const int b = k & 1;
        const auto t_w = std::chrono::steady_clock::now();
        if (S.used[b]) RSEM_HIP_TRY(hipEventSynchronize(S.ev[b]));  // the DMA that last read this buffer is finished
        const auto t_f = std::chrono::steady_clock::now();
        staged_stats().ns_wait += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_f - t_w).count();
        {
            std::vector<std::thread> th;
            const size_t per = (n + nthr - 1) / nthr;
            for (int t = 0; t < nthr; t++) {
                const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
                if (hi > lo) th.emplace_back([=, &S]() { memcpy((char*)S.buf[b] + lo, (const char*)src + done + lo, hi - lo); });
            }
            for (auto& x : th) x.join();
        }
        staged_stats().ns_fill += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_f).count();
        RSEM_HIP_TRY(hipMemcpyAsync((char*)dst + done, S.buf[b], n, hipMemcpyHostToDevice, st));
        RSEM_HIP_TRY(hipEventRecord(S.ev[b], st));
        S.used[b] = true;
        done += n;
        ++k;
    }
    return RSEM_OK;
}

}  // namespace rsem
