// lzf.hpp — LZF byte-stream coder for PCD "DATA binary_compressed" bodies (header-only, host).
//
// pcl::io::savePCDFileBinaryCompressed / loadPCDFile (what `rs-pcl` reads and writes through
// src/main.cpp:53,81,87 when a capture was stored compressed) keep the point data as one LZF
// stream over the fields laid out one after the other (all x, all y, all z, all rgb).  The
// stream format is Marc Lehmann's LZF:
//   ctrl < 32            : a literal run of ctrl + 1 bytes follows
//   ctrl >= 32           : a back reference; len = ctrl >> 5 (7: add the next byte),
//                          distance = ((ctrl & 31) << 8 | next byte) + 1, copy len + 2 bytes
// Any stream a conforming decoder accepts is valid; this encoder uses a small hash of
// 3-byte prefixes and greedy matching (written from the format, not from liblzf).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace rsreg {
namespace lzf {

// upper bound of the encoded size of n input bytes (worst case: literals only)
inline size_t max_encoded_size(size_t n) { return n + n / 32 + 2; }

// Encodes in[0..n) into out (capacity cap); returns the encoded size, 0 if cap is too small.
inline size_t encode(const uint8_t *in, size_t n, uint8_t *out, size_t cap)
{
    constexpr int kHashBits = 16;
    constexpr size_t kMaxDist = 1u << 13, kMaxLen = 264;   // 13-bit distance, len + 2 <= 7 + 255 + 2
    std::vector<uint32_t> head(1u << kHashBits, 0xffffffffu);
    size_t ip = 0, op = 0, lit = 0;   // lit: start of the pending literal run
    auto flush = [&](size_t end) -> bool {
        while (lit < end) {
            const size_t run = end - lit < 32 ? end - lit : 32;
            if (op + 1 + run > cap) return false;
            out[op++] = (uint8_t)(run - 1);
            std::memcpy(out + op, in + lit, run);
            op += run;
            lit += run;
        }
        return true;
    };
    while (ip + 2 < n) {
        const uint32_t v = (uint32_t)in[ip] | (uint32_t)in[ip + 1] << 8 | (uint32_t)in[ip + 2] << 16;
        const uint32_t h = (v * 2654435761u) >> (32 - kHashBits);
        const uint32_t cand = head[h];
        head[h] = (uint32_t)ip;
        size_t len = 0;
        if (cand != 0xffffffffu && ip - cand <= kMaxDist && in[cand] == in[ip] && in[cand + 1] == in[ip + 1] &&
            in[cand + 2] == in[ip + 2]) {
            const size_t lim = n - ip < kMaxLen ? n - ip : kMaxLen;
            len = 3;
            while (len < lim && in[cand + len] == in[ip + len]) ++len;
        }
        if (len < 3) {
            ++ip;
            continue;
        }
        if (!flush(ip)) return 0;
        const size_t dist = ip - cand - 1, l = len - 2;
        if (op + 3 > cap) return 0;
        if (l < 7) {
            out[op++] = (uint8_t)((l << 5) | (dist >> 8));
        } else {
            out[op++] = (uint8_t)((7u << 5) | (dist >> 8));
            out[op++] = (uint8_t)(l - 7);
        }
        out[op++] = (uint8_t)(dist & 0xff);
        ip += len;
        lit = ip;
    }
    if (!flush(n)) return 0;
    return op;
}

// Decodes in[0..n) into out (capacity cap); returns the decoded size, 0 on a malformed stream
// or if cap is too small.
inline size_t decode(const uint8_t *in, size_t n, uint8_t *out, size_t cap)
{
    size_t ip = 0, op = 0;
    while (ip < n) {
        const uint32_t ctrl = in[ip++];
        if (ctrl < 32) {
            const size_t run = ctrl + 1;
            if (ip + run > n || op + run > cap) return 0;
            std::memcpy(out + op, in + ip, run);
            ip += run;
            op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) {
                if (ip >= n) return 0;
                len += in[ip++];
            }
            if (ip >= n) return 0;
            const size_t dist = ((size_t)(ctrl & 31) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > cap) return 0;
            for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];   // may overlap: byte by byte
        }
    }
    return op;
}

}  // namespace lzf
}  // namespace rsreg
