// thp_probe.cpp -- measurement: are the parsers' big arrays (files.hpp, Arr::alloc: 2 MB-aligned, MADV_HUGEPAGE) backed by huge pages on this host, and what
// does giving them back cost?  g++ -O2 -pthread tools/thp_probe.cpp -o /tmp/thp_probe && /tmp/thp_probe [GB=8]
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
static long smaps(const char* key) {
    std::ifstream f("/proc/self/smaps_rollup");
    std::string l;
    while (std::getline(f, l)) if (l.rfind(key, 0) == 0) return atol(l.c_str() + strlen(key));
    return -1;
}
int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? (size_t)atol(argv[1]) : 8, bytes = gb << 30, huge = (size_t)2 << 20;
    for (int advise = 1; advise >= 0; advise--) {
        void* q = nullptr;
        if (posix_memalign(&q, huge, bytes) != 0) return 1;
        if (advise) madvise(q, bytes, MADV_HUGEPAGE);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < 16; t++) th.emplace_back([=]() { for (size_t o = bytes / 16 * t; o < bytes / 16 * (t + 1); o += 4096) ((char*)q)[o] = 1; });
        for (auto& x : th) x.join();
        const double touch = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const long anon_huge = smaps("AnonHugePages:"), rss = smaps("Rss:");
        t0 = std::chrono::steady_clock::now();
        free(q);
        const double fr = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%zu GB %s MADV_HUGEPAGE: first touch on 16 threads %.3f s, Rss %ld kB of which AnonHugePages %ld kB, free() %.3f s\n", gb, advise ? "with" : "without", touch, rss, anon_huge, fr);
    }
    std::ifstream e("/sys/kernel/mm/transparent_hugepage/enabled"), d("/sys/kernel/mm/transparent_hugepage/defrag");
    std::string a, b; std::getline(e, a); std::getline(d, b);
    printf("transparent_hugepage enabled: %s | defrag: %s\n", a.c_str(), b.c_str());
}
