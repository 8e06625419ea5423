"""Prototype of the register-blocked in-LDS negacyclic NTT index logic + LDS bank-conflict
simulator (design aid for toyfhe.jl_amd/csrc/ntt_core.h; not shipped, not a test).

Forward: merged-twist Cooley-Tukey (natural in, in-place bit-reversed positions), stage s has
m=2^s groups, twiddle W[m+i] = psi^brv(m+i).  Passes of k stages are done on 2^k registers.
"""
import random, sys
sys.path.insert(0, '/root/repo')
from oracle import spec


def brv(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1); x >>= 1
    return r


def tables(n, q, psi):
    N = 1 << n
    W = [0] * N
    Winv = [0] * N
    pinv = pow(psi, -1, q)
    for k in range(1, N):
        e = brv(k, n)
        W[k] = pow(psi, e, q); Winv[k] = pow(pinv, e, q)
    return W, Winv


def fwd_passes(a, n, q, W, passes, T):
    """a natural order; returns natural-order â. Emulates thread/register structure."""
    N = 1 << n
    e = n - (T.bit_length() - 1)
    lds = list(a)
    s0 = 0
    for k in passes:
        sets = 1 << (e - k)
        lo_bits = n - s0 - k
        new = list(lds)
        for tid in range(T):
            for u in range(sets):
                c = u * T + tid
                lo = c & ((1 << lo_bits) - 1); hi = c >> lo_bits
                idx = [(hi << (n - s0)) + (r << lo_bits) + lo for r in range(1 << k)]
                v = [lds[j] for j in idx]
                for d in range(k):
                    s = s0 + d
                    half = 1 << (k - 1 - d)
                    for r0 in range(1 << k):
                        if r0 & half: continue
                        tw = W[(1 << s) + (hi << d) + (r0 >> (k - d))]
                        U = v[r0]; V = v[r0 + half] * tw % q
                        v[r0] = (U + V) % q; v[r0 + half] = (U - V) % q
                for j, x in zip(idx, v): new[j] = x
        lds = new
        s0 += k
    return [lds[brv(kk, n)] for kk in range(N)]


def inv_passes(ahat, n, q, Winv, passes, T):
    N = 1 << n
    e = n - (T.bit_length() - 1)
    lds = [ahat[brv(j, n)] for j in range(N)]
    s_end = n
    for k in reversed(passes):
        s0 = s_end - k
        sets = 1 << (e - k)
        lo_bits = n - s0 - k
        new = list(lds)
        for tid in range(T):
            for u in range(sets):
                c = u * T + tid
                lo = c & ((1 << lo_bits) - 1); hi = c >> lo_bits
                idx = [(hi << (n - s0)) + (r << lo_bits) + lo for r in range(1 << k)]
                v = [lds[j] for j in idx]
                for d in reversed(range(k)):
                    s = s0 + d
                    half = 1 << (k - 1 - d)
                    for r0 in range(1 << k):
                        if r0 & half: continue
                        tw = Winv[(1 << s) + (hi << d) + (r0 >> (k - d))]
                        U = v[r0]; V = v[r0 + half]
                        v[r0] = (U + V) % q; v[r0 + half] = (U - V) * tw % q
                for j, x in zip(idx, v): new[j] = x
        lds = new
        s_end = s0
    ninv = pow(N, -1, q)
    return [x * ninv % q for x in lds]


if __name__ == '__main__':
    for n, T, passes in [(6, 4, [4, 2]), (8, 16, [4, 4]), (10, 64, [4, 4, 2]), (14, 1024, [4, 4, 4, 2])]:
        N = 1 << n
        q = spec.prime_chain(2**50 + 1, 1, N)[0]
        psi = spec.minimal_primitive_root(q, 2 * N)
        W, Winv = tables(n, q, psi)
        rng = random.Random(n)
        a = [rng.randrange(q) for _ in range(N)]
        want = spec.nntt(a, q, psi)
        got = fwd_passes(a, n, q, W, passes, T)
        back = inv_passes(want, n, q, Winv, passes, T)
        print(n, got == want, back == a)
