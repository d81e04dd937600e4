// inflate_fast_check.cpp -- TEST INFRASTRUCTURE for rsem_amd/csrc/host/inflate_fast.hpp: the decoder against zlib.
//   inflate_fast_check file <path> [block_bytes]   the file in blocks, each deflated by zlib at levels 0 .. 9 x strategies (default, filtered,
//                                                  Huffman only, RLE, fixed codes) and by deflate_fast.hpp, each stream inflated here and
//                                                  compared; then the rates of this decoder and zlib's on the level-6 streams
//   inflate_fast_check fuzz <seed> <blocks>        generated blocks; every stream also truncated and with bits flipped: the decoder must
//                                                  say false or return exactly the input (it may not read or write out of bounds: run under ASan)
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../rsem_amd/csrc/host/deflate_fast.hpp"
#include "../rsem_amd/csrc/host/inflate_fast.hpp"

static std::vector<uint8_t> zdeflate(const uint8_t* p, size_t n, int level, int strategy) {
    std::vector<uint8_t> o(n + n / 8 + 1024);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    zs.next_in = (Bytef*)p; zs.avail_in = (uInt)n;
    zs.next_out = o.data(); zs.avail_out = (uInt)o.size();
    deflate(&zs, Z_FINISH);
    o.resize(zs.total_out);
    deflateEnd(&zs);
    return o;
}

static bool check(rsemh::FastInflate& fi, const std::vector<uint8_t>& z, const uint8_t* p, size_t n, const char* what) {
    // the output sits in the middle of a guarded buffer: not a byte beyond n may change
    std::vector<uint8_t> back(n + 64, 0xA5);
    const bool ok = fi.inflate(z.data(), z.size(), back.data() + 32, n);
    for (size_t i = 0; i < 32; i++)
        if (back[i] != 0xA5 || back[32 + n + i] != 0xA5) { fprintf(stderr, "%s: wrote outside its output\n", what); return false; }
    if (!ok) { fprintf(stderr, "%s: said false on a good stream of %zu bytes for %zu\n", what, z.size(), n); return false; }
    if (memcmp(back.data() + 32, p, n) != 0) { fprintf(stderr, "%s: wrong bytes\n", what); return false; }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: inflate_fast_check file <path> [block] | fuzz <seed> <blocks>\n"); return 2; }
    std::unique_ptr<rsemh::FastInflate> fi(new rsemh::FastInflate());
    std::unique_ptr<rsemh::FastDeflate> fd(new rsemh::FastDeflate());
    static const int strategies[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
    if (std::string(argv[1]) == "file") {
        FILE* f = fopen(argv[2], "rb");
        if (!f) { perror(argv[2]); return 2; }
        std::vector<uint8_t> d;
        uint8_t buf[1 << 16];
        size_t r;
        while ((r = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + r);
        fclose(f);
        const size_t blk = argc > 3 ? (size_t)atoll(argv[3]) : 0xff00;
        size_t streams = 0;
        std::vector<std::vector<uint8_t>> l6;
        const bool big = d.size() > (32u << 20);
        for (size_t o = 0; o < d.size(); o += blk) {
            const size_t n = std::min(blk, d.size() - o);
            const uint8_t* p = d.data() + o;
            for (int lv = 0; lv <= 9; lv++) {
                if (big && lv != 1 && lv != 6) continue;
                for (int st = 0; st < 5; st++) {
                    if (big && st != 0) continue;
                    std::vector<uint8_t> z = zdeflate(p, n, lv, strategies[st]);
                    char what[64];
                    snprintf(what, sizeof(what), "offset %zu level %d strategy %d", o, lv, st);
                    if (!check(*fi, z, p, n, what)) { printf("FAILED\n"); return 1; }
                    ++streams;
                    if (lv == 6 && st == 0) l6.push_back(std::move(z));
                }
            }
            if (n <= rsemh::FastDeflate::kMaxIn) {
                std::vector<uint8_t> z(rsemh::FastDeflate::kMaxOut + 16);
                z.resize(fd->compress(p, n, z.data()));
                if (!check(*fi, z, p, n, "deflate_fast's stream")) { printf("FAILED\n"); return 1; }
                ++streams;
            }
        }
        // rates on the level-6 streams
        std::vector<uint8_t> back(blk + 64);
        const int reps = d.size() < (64u << 20) ? 5 : 1;
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps; rep++) {
            size_t k = 0;
            for (size_t o = 0; o < d.size(); o += blk, k++) fi->inflate(l6[k].data(), l6[k].size(), back.data(), std::min(blk, d.size() - o));
        }
        const double s_fast = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps; rep++) {
            size_t k = 0;
            for (size_t o = 0; o < d.size(); o += blk, k++) {
                z_stream zs;
                memset(&zs, 0, sizeof(zs));
                inflateInit2(&zs, -15);
                zs.next_in = l6[k].data(); zs.avail_in = (uInt)l6[k].size();
                zs.next_out = back.data(); zs.avail_out = (uInt)back.size();
                inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
            }
        }
        const double s_z = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("ok %zu streams of %zu bytes of input; level-6 streams: %.0f MB/s, zlib's inflate %.0f MB/s\n", streams, d.size(), d.size() / 1e6 / s_fast, d.size() / 1e6 / s_z);
        return 0;
    }
    const uint64_t seed = strtoull(argv[2], nullptr, 10);
    const long blocks = argc > 3 ? atol(argv[3]) : 1000;
    std::mt19937_64 rng(seed);
    std::vector<uint8_t> d(70000);
    long rejected = 0, survived = 0;
    for (long b = 0; b < blocks; b++) {
        const int kind = (int)(rng() % 7);
        size_t n;
        switch (rng() % 5) {
            case 0: n = rng() % 64; break;
            case 1: n = 0xff00 - rng() % 8; break;
            default: n = rng() % (0xff00 + 1);
        }
        if (kind == 0) for (size_t i = 0; i < n; i++) d[i] = (uint8_t)rng();
        else if (kind == 1) memset(d.data(), (int)(rng() & 0xff), n);
        else if (kind == 2) for (size_t i = 0; i < n; i++) d[i] = (uint8_t)("ACGT"[rng() & 3]);
        else if (kind == 3) {
            size_t i = 0;
            while (i < n) {
                uint8_t rec[400];
                const size_t rl = 150 + rng() % 200;
                for (size_t k = 0; k < rl; k++) rec[k] = (uint8_t)(rng() % (k < 40 ? 256 : 41));
                const int copies = 1 + (int)(rng() % 16);
                for (int c = 0; c < copies && i < n; c++) {
                    for (int m = 0; m < 6; m++) rec[rng() % 36] = (uint8_t)rng();
                    for (size_t k = 0; k < rl && i < n; k++) d[i++] = rec[k];
                }
            }
        } else if (kind == 4) { const size_t per = 1 + rng() % 300; for (size_t i = 0; i < n; i++) d[i] = (uint8_t)((i % per) * 7 + (i / per)); }
        else if (kind == 5) for (size_t i = 0; i < n; i++) d[i] = (uint8_t)((rng() % 100 < 97) ? 0 : rng());
        else { size_t i = 0; while (i < n) { const size_t run = 1 + rng() % 600; const uint8_t v = (uint8_t)rng(); for (size_t k = 0; k < run && i < n; k++) d[i++] = v; } }
        const int lv = (int)(rng() % 10), st = (int)(rng() % 5);
        std::vector<uint8_t> z = (rng() % 4 == 0) ? std::vector<uint8_t>() : zdeflate(d.data(), n, lv, strategies[st]);
        if (z.empty()) { z.resize(rsemh::FastDeflate::kMaxOut + 16); z.resize(fd->compress(d.data(), n, z.data())); }
        if (!check(*fi, z, d.data(), n, "fuzz")) { printf("FAILED: seed %llu block %ld kind %d length %zu level %d strategy %d\n", (unsigned long long)seed, b, kind, n, lv, st); return 1; }
        // damaged streams: false, or (a flipped bit may not matter) exactly the input -- never a crash, never a byte outside
        for (int t = 0; t < 6; t++) {
            std::vector<uint8_t> bad = z;
            if (t < 2 && !bad.empty()) bad.resize(rng() % bad.size());
            else if (!bad.empty()) for (int k = 0; k < 1 + t; k++) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() & 7));
            std::vector<uint8_t> back(n + 64, 0xA5);
            const bool ok = fi->inflate(bad.data(), bad.size(), back.data() + 32, n);
            for (size_t i = 0; i < 32; i++)
                if (back[i] != 0xA5 || back[32 + n + i] != 0xA5) { printf("FAILED: a damaged stream made it write outside its output (seed %llu block %ld)\n", (unsigned long long)seed, b); return 1; }
            if (ok) ++survived; else ++rejected;
        }
    }
    printf("ok %ld blocks; damaged streams: %ld rejected, %ld inflated to the right length (the caller's CRC decides)\n", blocks, rejected, survived);
    return 0;
}
