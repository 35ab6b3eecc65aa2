// Text primitives of the record writers (transcoder.cpp): a growable byte buffer written through raw pointers, and
// the `%g` of a float the way htslib prints SAM `f` tags (kputd / ksprintf "%g"), without going through printf.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>

namespace phq {

struct ByteBuffer {
    std::unique_ptr<char[]> data;
    size_t size = 0;
    size_t capacity = 0;
    void clear() { size = 0; }
    bool empty() const { return size == 0; }
    // room for `need` more bytes; returns the write position
    char* reserve(size_t need) {
        if(size + need > capacity) {
            size_t grown = capacity < (1u << 16) ? (1u << 16) : capacity * 2;
            while(grown < size + need) { grown *= 2; }
            std::unique_ptr<char[]> larger(new char[grown]);
            if(size > 0) { std::memcpy(larger.get(), data.get(), size); }
            data.swap(larger);
            capacity = grown;
        }
        return data.get() + size;
    }
    void commit(char* end) { size = static_cast<size_t>(end - data.get()); }
};

inline char* put_bytes(char* to, const char* from, size_t n) { std::memcpy(to, from, n); return to + n; }
template <size_t N> inline char* put_literal(char* to, const char (&text)[N]) { std::memcpy(to, text, N - 1); return to + (N - 1); }
inline char* put_unsigned(char* to, uint32_t value) {
    char scratch[12];
    int n = 0;
    do { scratch[n++] = static_cast<char>('0' + value % 10); value /= 10; } while(value > 0);
    while(n > 0) { *to++ = scratch[--n]; }
    return to;
}

// printf("%g", (double)value) for a float: 6 significant digits, exact decimal expansion of the binary value rounded
// half to even, trailing zeros dropped, scientific when the decimal exponent is < -4 (or >= 6).  Exact integer
// arithmetic in 128 bits for 1e-20 <= value < 1 (every SAM XB / XM / XC value that is not astronomically small);
// anything else takes snprintf.  At most 16 bytes are written.
inline char* put_float_g(char* to, float value) {
    const double d = static_cast<double>(value);
    if(!(d >= 1e-20 && d < 1.0)) {
        return to + std::snprintf(to, 32, "%g", d);
    }
    // a normal float (1e-20 is far above the subnormals): 24 bit mantissa and binary exponent straight from its bits
    uint32_t bits;
    std::memcpy(&bits, &value, sizeof(bits));
    const int binary_exponent = static_cast<int>((bits >> 23) & 0xffu) - 126;   // d = fraction * 2^e, fraction in [0.5, 1)
    const uint64_t mantissa = (bits & 0x7fffffu) | 0x800000u;                   // fraction * 2^24
    const int shift = 24 - binary_exponent;                                   // d = mantissa / 2^shift, shift in [24, 91]
    static const unsigned __int128 power_of_ten[27] = {
        (unsigned __int128)1ull, (unsigned __int128)10ull, (unsigned __int128)100ull, (unsigned __int128)1000ull, (unsigned __int128)10000ull,
        (unsigned __int128)100000ull, (unsigned __int128)1000000ull, (unsigned __int128)10000000ull, (unsigned __int128)100000000ull,
        (unsigned __int128)1000000000ull, (unsigned __int128)10000000000ull, (unsigned __int128)100000000000ull,
        (unsigned __int128)1000000000000ull, (unsigned __int128)10000000000000ull, (unsigned __int128)100000000000000ull,
        (unsigned __int128)1000000000000000ull, (unsigned __int128)10000000000000000ull, (unsigned __int128)100000000000000000ull,
        (unsigned __int128)1000000000000000000ull, (unsigned __int128)10000000000000000000ull,
        (unsigned __int128)10000000000000000000ull * 10u, (unsigned __int128)10000000000000000000ull * 100u,
        (unsigned __int128)10000000000000000000ull * 1000u, (unsigned __int128)10000000000000000000ull * 10000u,
        (unsigned __int128)10000000000000000000ull * 100000u, (unsigned __int128)10000000000000000000ull * 1000000u,
        (unsigned __int128)10000000000000000000ull * 10000000u};
    // floor(log10(d)) from the binary exponent, 2^(e-1) <= d < 2^e: (e - 1) log10(2) rounded down is at most one too
    // small, the loop below settles it
    int exponent = ((binary_exponent - 1) * 78913) >> 18;
    uint64_t digits = 0;
    for(int attempt = 0; attempt < 3; ++attempt) {
        const int scale = 5 - exponent;                                       // 6 .. 25 for the accepted range (+-1)
        if(scale < 0 || scale > 26) { return to + std::snprintf(to, 32, "%g", d); }
        const unsigned __int128 numerator = static_cast<unsigned __int128>(mantissa) * power_of_ten[scale];   // < 2^24 * 2^87
        const unsigned __int128 whole = numerator >> shift;
        if(whole >= 1000000u) { ++exponent; continue; }
        if(whole < 100000u) { --exponent; continue; }
        const unsigned __int128 remainder = numerator & ((static_cast<unsigned __int128>(1) << shift) - 1);
        const unsigned __int128 half = static_cast<unsigned __int128>(1) << (shift - 1);
        digits = static_cast<uint64_t>(whole);
        if(remainder > half || (remainder == half && (digits & 1u))) { ++digits; }
        if(digits == 1000000u) { digits = 100000u; ++exponent; }
        break;
    }
    if(digits == 0) { return to + std::snprintf(to, 32, "%g", d); }
    char text[6];
    uint64_t rest = digits;
    for(int i = 5; i >= 0; --i) { text[i] = static_cast<char>('0' + rest % 10); rest /= 10; }
    int significant = 6;
    while(significant > 1 && text[significant - 1] == '0') { --significant; }
    if(exponent < -4) {
        *to++ = text[0];
        if(significant > 1) { *to++ = '.'; to = put_bytes(to, text + 1, static_cast<size_t>(significant - 1)); }
        *to++ = 'e'; *to++ = '-';
        const int magnitude = -exponent;
        if(magnitude < 10) { *to++ = '0'; }
        return put_unsigned(to, static_cast<uint32_t>(magnitude));
    }
    if(exponent >= 0) { return to + std::snprintf(to, 32, "%g", d); }       // not reached for value < 1 (kept for safety)
    *to++ = '0'; *to++ = '.';
    for(int i = 0; i < -exponent - 1; ++i) { *to++ = '0'; }
    return put_bytes(to, text, static_cast<size_t>(significant));
}

}  // namespace phq
