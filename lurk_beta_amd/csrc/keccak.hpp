// keccak.hpp - Keccak-f[1600] on the host and the two sponges the library uses: SHAKE256 (keygen.hip: the XOF behind from_label) and
// the pre-standard Keccak-256 (transcript.hip: arecibo's Keccak256Transcript).  Host code only.
#pragma once
#include <cstdint>
#include <cstring>

namespace lurk {

inline void keccak_f(uint64_t* s) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; r++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
        }
        uint64_t t = s[1];
        for (int i = 0; i < 24; i++) {
            int j = PIL[i];
            uint64_t b = s[j];
            s[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = s[j + i];
            for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        s[0] ^= RC[r];
    }
}

// Keccak-256 as the `sha3` crate's Keccak256 (what arecibo's transcript is built on): rate 136, domain byte 0x01 (not SHA3's 0x06);
// incremental and copyable (the transcript clones a running hasher)
struct Keccak256 {
    static constexpr size_t RATE = 136;
    uint64_t st[25];
    size_t pos = 0;
    Keccak256() { memset(st, 0, sizeof(st)); }
    void update(const void* data, size_t len) {
        uint8_t* sb = reinterpret_cast<uint8_t*>(st);  // little-endian host
        const uint8_t* in = (const uint8_t*)data;
        for (size_t i = 0; i < len; i++) {
            sb[pos++] ^= in[i];
            if (pos == RATE) { keccak_f(st); pos = 0; }
        }
    }
    void finalize(uint8_t out[32]) const {  // leaves the running state untouched
        Keccak256 c = *this;
        uint8_t* sb = reinterpret_cast<uint8_t*>(c.st);
        sb[c.pos] ^= 0x01;
        sb[RATE - 1] ^= 0x80;
        keccak_f(c.st);
        memcpy(out, sb, 32);
    }
};

}  // namespace lurk
