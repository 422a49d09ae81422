// SHA-1 ids on the host for the records the consumer builds itself (splice-side merges, records K3b did not hash): the message
// `format!("{:?}{}{}", &seq, transcript_id, offset)` (reference: src/microphasing.rs:667-675, src/common.rs:387-395) is laid out and padded
// in one stack buffer and hashed with the SHA extensions of the host CPU where it has them (a portable block function otherwise).
// Same bytes as haplotype_id of util.hpp, which the CPU oracle keeps using.
#pragma once
#include <cpuid.h>
#include <immintrin.h>

#include <charconv>
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

namespace mp {

namespace hostsha {

inline uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

inline void blocks_portable(uint32_t h[5], const uint8_t* p, size_t nblocks) {
    for (; nblocks; nblocks--, p += 64) {
        uint32_t w[80];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t(p[4 * i]) << 24) | (uint32_t(p[4 * i + 1]) << 16) | (uint32_t(p[4 * i + 2]) << 8) | p[4 * i + 3];
        for (int i = 16; i < 80; i++) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
#define MP_SHA_ROUND(f, k) { const uint32_t t = rol(a, 5) + (f) + e + (k) + w[i]; e = d; d = c; c = rol(b, 30); b = a; a = t; }
        int i = 0;
        for (; i < 20; i++) MP_SHA_ROUND((b & c) | (~b & d), 0x5A827999u)
        for (; i < 40; i++) MP_SHA_ROUND(b ^ c ^ d, 0x6ED9EBA1u)
        for (; i < 60; i++) MP_SHA_ROUND((b & c) | (b & d) | (c & d), 0x8F1BBCDCu)
        for (; i < 80; i++) MP_SHA_ROUND(b ^ c ^ d, 0xCA62C1D6u)
#undef MP_SHA_ROUND
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
    }
}

// Four rounds per sha1rnds4; message schedule by sha1msg1 / sha1msg2, E by sha1nexte (Intel SHA extensions).
__attribute__((target("sha,sse4.1,ssse3"))) inline void blocks_shani(uint32_t h[5], const uint8_t* p, size_t nblocks) {
    const __m128i MASK = _mm_set_epi64x(0x0001020304050607ll, 0x08090a0b0c0d0e0fll);
    __m128i ABCD = _mm_shuffle_epi32(_mm_loadu_si128(reinterpret_cast<const __m128i*>(h)), 0x1B);
    __m128i E0 = _mm_set_epi32(int(h[4]), 0, 0, 0), E1;
    for (; nblocks; nblocks--, p += 64) {
        const __m128i ABCD_SAVE = ABCD, E0_SAVE = E0;
        __m128i M0 = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p)), MASK);
        __m128i M1 = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16)), MASK);
        __m128i M2 = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 32)), MASK);
        __m128i M3 = _mm_shuffle_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 48)), MASK);
        // rounds 0-15
        E0 = _mm_add_epi32(E0, M0); E1 = ABCD; ABCD = _mm_sha1rnds4_epu32(ABCD, E0, 0);
        E1 = _mm_sha1nexte_epu32(E1, M1); E0 = ABCD; ABCD = _mm_sha1rnds4_epu32(ABCD, E1, 0); M0 = _mm_sha1msg1_epu32(M0, M1);
        E0 = _mm_sha1nexte_epu32(E0, M2); E1 = ABCD; ABCD = _mm_sha1rnds4_epu32(ABCD, E0, 0); M1 = _mm_sha1msg1_epu32(M1, M2); M0 = _mm_xor_si128(M0, M2);
        // one group of four rounds: E (current) takes the next message words A; B's schedule is finished with A, D's is started, C is mixed
#define MP_SHA_G(EC, EO, A, B, C, D, F) EC = _mm_sha1nexte_epu32(EC, A); EO = ABCD; B = _mm_sha1msg2_epu32(B, A); ABCD = _mm_sha1rnds4_epu32(ABCD, EC, F); D = _mm_sha1msg1_epu32(D, A); C = _mm_xor_si128(C, A);
        MP_SHA_G(E1, E0, M3, M0, M1, M2, 0)   // 12-15
        MP_SHA_G(E0, E1, M0, M1, M2, M3, 0)   // 16-19
        MP_SHA_G(E1, E0, M1, M2, M3, M0, 1)   // 20-23
        MP_SHA_G(E0, E1, M2, M3, M0, M1, 1)   // 24-27
        MP_SHA_G(E1, E0, M3, M0, M1, M2, 1)   // 28-31
        MP_SHA_G(E0, E1, M0, M1, M2, M3, 1)   // 32-35
        MP_SHA_G(E1, E0, M1, M2, M3, M0, 1)   // 36-39
        MP_SHA_G(E0, E1, M2, M3, M0, M1, 2)   // 40-43
        MP_SHA_G(E1, E0, M3, M0, M1, M2, 2)   // 44-47
        MP_SHA_G(E0, E1, M0, M1, M2, M3, 2)   // 48-51
        MP_SHA_G(E1, E0, M1, M2, M3, M0, 2)   // 52-55
        MP_SHA_G(E0, E1, M2, M3, M0, M1, 2)   // 56-59
        MP_SHA_G(E1, E0, M3, M0, M1, M2, 3)   // 60-63
        MP_SHA_G(E0, E1, M0, M1, M2, M3, 3)   // 64-67
        MP_SHA_G(E1, E0, M1, M2, M3, M0, 3)   // 68-71
        MP_SHA_G(E0, E1, M2, M3, M0, M1, 3)   // 72-75
        MP_SHA_G(E1, E0, M3, M0, M1, M2, 3)   // 76-79 (the schedule steps past the last word are unused)
#undef MP_SHA_G
        E0 = _mm_sha1nexte_epu32(E0, E0_SAVE);
        ABCD = _mm_add_epi32(ABCD, ABCD_SAVE);
    }
    _mm_storeu_si128(reinterpret_cast<__m128i*>(h), _mm_shuffle_epi32(ABCD, 0x1B));
    h[4] = uint32_t(_mm_extract_epi32(E0, 3));
}

inline bool cpu_has_sha() {
    static const bool has = [] {
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
        const bool sha = (b >> 29) & 1;
        if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
        return sha && ((c >> 19) & 1) && ((c >> 9) & 1);   // SHA + SSE4.1 + SSSE3
    }();
    return has;
}

}  // namespace hostsha

// The id of a haplotype window: first 15 hex characters of SHA-1(format!("{:?}{}{}", seq, transcript_id, offset)) + 'F' | 'R'
inline void haplotype_id_into(std::string& id, const uint8_t* seq, size_t n, std::string_view transcript_id, uint64_t offset, char strand_initial) {
    uint8_t stack[512];
    std::vector<uint8_t> heap;
    const size_t bound = 5 * n + transcript_id.size() + 24 + 2 + 72;   // text + padding
    uint8_t* m = stack;
    if (bound > sizeof stack) { heap.resize(bound + 64); m = heap.data(); }
    size_t len = 0;
    m[len++] = '[';
    for (size_t i = 0; i < n; i++) {   // `{:?}` of a Vec<u8>: "[65, 67, ...]"
        if (i) { m[len++] = ','; m[len++] = ' '; }
        const unsigned v = seq[i];
        if (v >= 100) m[len++] = uint8_t('0' + v / 100);
        if (v >= 10) m[len++] = uint8_t('0' + (v / 10) % 10);
        m[len++] = uint8_t('0' + v % 10);
    }
    m[len++] = ']';
    std::memcpy(m + len, transcript_id.data(), transcript_id.size());
    len += transcript_id.size();
    {
        char buf[24];
        const auto r = std::to_chars(buf, buf + sizeof buf, offset);
        std::memcpy(m + len, buf, size_t(r.ptr - buf));
        len += size_t(r.ptr - buf);
    }
    const uint64_t bits = uint64_t(len) * 8;
    m[len++] = 0x80;
    while (len % 64 != 56) m[len++] = 0;
    for (int i = 0; i < 8; i++) m[len++] = uint8_t(bits >> (56 - 8 * i));
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    if (hostsha::cpu_has_sha()) hostsha::blocks_shani(h, m, len / 64);
    else hostsha::blocks_portable(h, m, len / 64);
    char out[16];
    for (int i = 0; i < 15; i++) out[i] = "0123456789abcdef"[(h[i >> 3] >> (28 - 4 * (i & 7))) & 0xF];
    out[15] = strand_initial;
    id.assign(out, 16);
}

}  // namespace mp
