// Host-only: the known-answer vectors of Random123 (kat_vectors) through rsem_amd/csrc/rng.hpp's own Philox functions.
#include <cstdio>
#include "../rsem_amd/csrc/rng.hpp"
int main() {
    int bad = 0;
    struct K4 { uint32_t c[4], k[2], o[4]; };
    const K4 k4[] = {{{0, 0, 0, 0}, {0, 0}, {0x6627e8d5u, 0xe169c58du, 0xbc57ac4cu, 0x9b00dbd8u}},
                     {{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu}, {0x408f276du, 0x41c83b0eu, 0xa20bc7c6u, 0x6d5451fdu}},
                     {{0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u}, {0xa4093822u, 0x299f31d0u}, {0xd16cfe09u, 0x94fdccebu, 0x5001e420u, 0x24126ea1u}}};
    for (const K4& v : k4) {
        rsem::Philox ph{v.k[0], v.k[1]};
        uint32_t o[4];
        ph.gen(v.c[0], v.c[1], v.c[2], v.c[3], o);
        for (int i = 0; i < 4; i++) if (o[i] != v.o[i]) { printf("philox4x32-10 mismatch %08x != %08x\n", o[i], v.o[i]); bad++; }
    }
    struct K2 { uint32_t c[2], k, o[2]; };
    const K2 k2[] = {{{0, 0}, 0, {0xff1dae59u, 0x6cd10df2u}}, {{0xffffffffu, 0xffffffffu}, 0xffffffffu, {0x2c3f628bu, 0xab4fd7adu}},
                     {{0x243f6a88u, 0x85a308d3u}, 0x13198a2eu, {0xdd7ce038u, 0xf62a4c12u}}};
    for (const K2& v : k2) {
        uint32_t o[2];
        rsem::philox2x32_10(v.k, v.c[0], v.c[1], o);
        for (int i = 0; i < 2; i++) if (o[i] != v.o[i]) { printf("philox2x32-10 mismatch %08x != %08x\n", o[i], v.o[i]); bad++; }
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
