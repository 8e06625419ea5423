// host_math.h -- host-side number theory and small multi-word integers used to build the device
// tables (twiddles, Shoup/Barrett constants, base-conversion tables).  Product code: independent of
// oracle/.  Mirrors what the reference gets from GaloisFields.jl / Primes.jl / BigInt at ring
// construction time (src/pow2_cyc_rings.jl:27-44, src/crt.jl:282-295, src/crt.jl:98-112).
#pragma once
#include <stdint.h>
#include <vector>
#include "modarith.h"

namespace hostmath {

inline u64 mulmod_slow(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
inline u64 powmod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mulmod_slow(r, a, q);
        a = mulmod_slow(a, a, q);
    }
    return r;
}
inline bool is_prime(u64 n) {  // deterministic Miller-Rabin for 64-bit
    if (n < 2) return false;
    static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : bases) {
        if (n % p == 0) return n == p;
    }
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; s++; }
    for (u64 a : bases) {
        u64 x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s; i++) {
            x = mulmod_slow(x, x, n);
            if (x == n - 1) { comp = false; break; }
        }
        if (comp) return false;
    }
    return true;
}
inline u64 invmod_prime(u64 a, u64 q) { return powmod(a % q, q - 2, q); }

// numerically smallest element of exact order n (n = 2N a power of two) in 𝔽q -- the
// GaloisFields.minimal_primitive_root the reference uses for ψ (pow2_cyc_rings.jl:40).
inline u64 minimal_primitive_root(u64 q, u64 n) {
    u64 z = 0;
    for (u64 g = 2;; g++) {
        z = powmod(g, (q - 1) / n, q);
        if (powmod(z, n / 2, q) == q - 1) break;
    }
    const u64 z2 = mulmod_slow(z, z, q);
    u64 best = z, cur = z;
    for (u64 i = 1; i < n / 2; i++) {
        cur = mulmod_slow(cur, z2, q);
        if (cur < best) best = cur;
    }
    return best;
}

inline tw_t make_tw(u64 w, u64 q) { return tw_t{w, (u64)(((u128)w << 64) / q)}; }
inline int bitlen(u64 x) { return x ? 64 - __builtin_clzll(x) : 0; }
inline barrett_t make_barrett(u64 q) {
    barrett_t b;
    const int k = bitlen(q);
    b.q = q;
    b.sh = (u32)(k - 2);
    b.mu = (u64)((((u128)1) << (k + 62)) / q);
    return b;
}

// ---- tiny unsigned multi-word integers (little-endian words) ----
typedef std::vector<u64> bigint;
inline void big_trim(bigint& a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); }
inline bigint big_from(u64 x) { return bigint{x}; }
inline bigint big_mul_u64(const bigint& a, u64 k) {
    bigint r(a.size() + 1, 0);
    u64 c = 0;
    for (size_t i = 0; i < a.size(); i++) {
        u128 p = (u128)a[i] * k + c;
        r[i] = (u64)p;
        c = (u64)(p >> 64);
    }
    r[a.size()] = c;
    big_trim(r);
    return r;
}
inline u64 big_mod_u64(const bigint& a, u64 m) {
    u128 r = 0;
    for (size_t i = a.size(); i-- > 0;) r = ((r << 64) | a[i]) % m;
    return (u64)r;
}
inline bigint big_shr1(const bigint& a) {
    bigint r(a.size(), 0);
    for (size_t i = 0; i < a.size(); i++) r[i] = (a[i] >> 1) | (i + 1 < a.size() ? a[i + 1] << 63 : 0);
    big_trim(r);
    return r;
}
inline int big_cmp(const bigint& a, const bigint& b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = a.size(); i-- > 0;)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}

}  // namespace hostmath
