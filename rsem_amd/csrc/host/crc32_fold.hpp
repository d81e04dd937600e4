// crc32_fold.hpp -- CRC-32 (the gzip / BGZF one: polynomial 0x04c11db7, reflected) by carry-less multiplication: 64 bytes are
// folded per step with PCLMULQDQ (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel
// 2009: fold constants x^(512+32), x^(512-32), x^(128+32), x^(128-32), x^64 mod P, Barrett reduction at the end), the head and tail
// that do not fill 16 bytes by zlib's table routine.  zlib 1.2.11's crc32() runs at 1.3 GB/s; a BGZF writer checksums every byte it
// compresses, and with this repository's encoder (deflate_fast.hpp) at 0.6 GB/s that was a third of a block's time
// (profiles/r06p_*).  Falls back to zlib's crc32 where the CPU lacks the instruction.  tests/test_deflate_fast_cpu.py holds it to
// zlib's on every length 0 .. 300 and on random buffers.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace rsemh {

#if defined(__x86_64__)
// buf: len >= 64, len % 16 == 0; crc: the running register (already inverted the way zlib keeps it internally)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold_blocks(const unsigned char* buf, size_t len, uint32_t crc) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64;
    len -= 64;
    while (len >= 64) {  // four lanes of 16 bytes, each folded 64 bytes ahead
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);  // the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {  // what is left, 16 bytes at a time
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);  // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);  // Barrett: 64 -> 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
inline bool crc32_fold_usable() {
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return ok;
}
#endif

// crc32 of p[0 .. n), continuing `crc` (as zlib's crc32(crc, p, n); start with crc32(0, NULL, 0) = 0)
inline uint32_t crc32_fast(uint32_t crc, const unsigned char* p, size_t n) {
#if defined(__x86_64__)
    if (n >= 64 + 16 && crc32_fold_usable()) {
        const size_t body = (n & ~(size_t)15);
        crc = ~crc32_fold_blocks(p, body, ~crc);
        p += body;
        n -= body;
    }
#endif
    return n ? (uint32_t)crc32(crc, p, (uInt)n) : crc;
}

}  // namespace rsemh
