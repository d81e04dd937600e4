// deflate_fast_check.cpp -- TEST INFRASTRUCTURE for rsem_amd/csrc/host/deflate_fast.hpp: every block the encoder writes is inflated
// with zlib and compared with its input.
//   deflate_fast_check file <path> [block_bytes]   the file in blocks (default 65280): bytes in / out, MB/s, and zlib level 6 beside it
//   deflate_fast_check fuzz <seed> <blocks>        generated blocks of many kinds and lengths
// Prints "ok <blocks> <bytes_in> <bytes_out> ..." or the first mismatch; exit code 0 / 1.
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <random>
#include <vector>

#include "../rsem_amd/csrc/host/deflate_fast.hpp"

static bool roundtrip(rsemh::FastDeflate& fd, const uint8_t* p, size_t n, size_t& out_bytes) {
    static std::vector<uint8_t> out(rsemh::FastDeflate::kMaxOut + 64), back(rsemh::FastDeflate::kMaxIn + 64);
    const size_t k = fd.compress(p, n, out.data());
    out_bytes = k;
    if (k > rsemh::FastDeflate::kMaxOut) { fprintf(stderr, "output of %zu bytes for %zu\n", k, n); return false; }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = out.data(); zs.avail_in = (uInt)k;
    zs.next_out = back.data(); zs.avail_out = (uInt)back.size();
    const int rc = inflate(&zs, Z_FINISH);
    const size_t got = zs.total_out, used = zs.total_in;
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) { fprintf(stderr, "inflate says %d (%s) for a block of %zu bytes\n", rc, zs.msg ? zs.msg : "-", n); return false; }
    if (got != n || used != k || memcmp(back.data(), p, n) != 0) { fprintf(stderr, "mismatch: %zu bytes in, %zu back, %zu of %zu consumed\n", n, got, used, k); return false; }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: deflate_fast_check file <path> [block] | fuzz <seed> <blocks>\n"); return 2; }
    std::unique_ptr<rsemh::FastDeflate> fd(new rsemh::FastDeflate());
    if (std::string(argv[1]) == "file") {
        FILE* f = fopen(argv[2], "rb");
        if (!f) { perror(argv[2]); return 2; }
        std::vector<uint8_t> d;
        uint8_t buf[1 << 16];
        size_t r;
        while ((r = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + r);
        fclose(f);
        const size_t blk = argc > 3 ? (size_t)atoll(argv[3]) : rsemh::FastDeflate::kMaxIn;
        size_t in = 0, out = 0, nb = 0;
        for (size_t o = 0; o < d.size(); o += blk) {
            const size_t n = std::min(blk, d.size() - o);
            size_t k;
            if (!roundtrip(*fd, d.data() + o, n, k)) { printf("FAILED at offset %zu\n", o); return 1; }
            in += n; out += k; ++nb;
        }
        // rates (compress only), this encoder and zlib level 6
        std::vector<uint8_t> o2(rsemh::FastDeflate::kMaxOut + 64);
        const int reps = d.size() < (64u << 20) ? 5 : 1;
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps; rep++)
            for (size_t o = 0; o < d.size(); o += blk) fd->compress(d.data() + o, std::min(blk, d.size() - o), o2.data());
        const double s_fast = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        size_t zout = 0;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        t0 = std::chrono::steady_clock::now();
        for (size_t o = 0; o < d.size(); o += blk) {
            deflateReset(&zs);
            zs.next_in = d.data() + o; zs.avail_in = (uInt)std::min(blk, d.size() - o);
            zs.next_out = o2.data(); zs.avail_out = (uInt)o2.size();
            deflate(&zs, Z_FINISH);
            zout += zs.total_out;
        }
        const double s_z = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        deflateEnd(&zs);
        printf("ok %zu blocks, %zu bytes in, %zu out (%.3f), %.0f MB/s; zlib level 6: %zu out (%.3f), %.0f MB/s\n", nb, in, out, (double)out / in,
               in / 1e6 / s_fast, zout, (double)zout / in, in / 1e6 / s_z);
        return 0;
    }
    const uint64_t seed = strtoull(argv[2], nullptr, 10);
    const long blocks = argc > 3 ? atol(argv[3]) : 1000;
    std::mt19937_64 rng(seed);
    std::vector<uint8_t> d(rsemh::FastDeflate::kMaxIn + 16);
    size_t in = 0, out = 0;
    for (long b = 0; b < blocks; b++) {
        const int kind = (int)(rng() % 9);
        size_t n;
        switch (rng() % 6) {
            case 0: n = rng() % 64; break;
            case 1: n = rsemh::FastDeflate::kMaxIn - rng() % 8; break;
            case 2: n = 250 + rng() % 20; break;
            default: n = rng() % (rsemh::FastDeflate::kMaxIn + 1);
        }
        if (kind == 0) for (size_t i = 0; i < n; i++) d[i] = (uint8_t)rng();                       // noise
        else if (kind == 1) memset(d.data(), (int)(rng() & 0xff), n);                               // one byte
        else if (kind == 2) for (size_t i = 0; i < n; i++) d[i] = (uint8_t)("ACGT"[rng() & 3]);     // four letters
        else if (kind == 3) {                                                                       // records: a read repeated with small changes
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
        } else if (kind == 4) { const size_t per = 1 + rng() % 300; for (size_t i = 0; i < n; i++) d[i] = (uint8_t)((i % per) * 7 + (i / per)); }  // near-periodic
        else if (kind == 5) for (size_t i = 0; i < n; i++) d[i] = (uint8_t)((rng() % 100 < 97) ? 0 : rng());  // sparse
        else if (kind == 6) { const size_t per = 1 + rng() % 40000; for (size_t i = 0; i < n; i++) d[i] = i < per ? (uint8_t)rng() : d[i - per]; }  // far repeats (beyond 32768 too)
        else if (kind == 7) for (size_t i = 0; i < n; i++) d[i] = (uint8_t)(rng() % (2 + rng() % 3));  // tiny alphabet
        else { size_t i = 0; while (i < n) { const size_t run = 1 + rng() % 600; const uint8_t v = (uint8_t)rng(); for (size_t k = 0; k < run && i < n; k++) d[i++] = v; } }  // runs
        size_t k;
        if (!roundtrip(*fd, d.data(), n, k)) { printf("FAILED: seed %llu block %ld kind %d length %zu\n", (unsigned long long)seed, b, kind, n); return 1; }
        in += n; out += k;
    }
    printf("ok %ld blocks, %zu bytes in, %zu out\n", blocks, in, out);
    return 0;
}
