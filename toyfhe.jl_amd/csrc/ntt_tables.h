// ntt_tables.h -- host construction of the per-limb NTT tables (pure host C++, shared by
// toyfhe_hip.hip and tests/emul/).  W[k] = psi^brv(k), Winv[k] = psi^-brv(k), each with its Shoup
// companion; psi is the ring's 2N-th root (src/pow2_cyc_rings.jl:27-44).
#pragma once
#include <vector>

#include "host_math.h"
#include "ntt_core.h"

// returns 0, or -1 if psi is not a primitive 2N-th root of unity mod q
inline int build_ntt_tables(int64_t N, u64 q, u64 psi, std::vector<twd_t>& W, std::vector<twd_t>& Wi, ntt_limb_t* L,
                            std::vector<ftwd_t>* Wd = nullptr, std::vector<ftwd_t>* Wid = nullptr) {
    using namespace hostmath;
    int logN = 0;
    while ((1ll << logN) < N) logN++;
    if (psi >= q || powmod(psi, 2 * (u64)N, q) != 1 || powmod(psi, (u64)N, q) != q - 1) return -1;
    const u64 pinv = invmod_prime(psi, q);
    std::vector<u64> pw((size_t)N), pwi((size_t)N);
    u64 a = 1, b = 1;
    for (int64_t i = 0; i < N; i++) {
        pw[i] = a;
        pwi[i] = b;
        a = mulmod_slow(a, psi, q);
        b = mulmod_slow(b, pinv, q);
    }
    W.resize((size_t)N);
    Wi.resize((size_t)N);
    for (int64_t k = 0; k < N; k++) {
        u32 r = 0;
        for (int bit = 0; bit < logN; bit++) r |= (u32)((k >> bit) & 1) << (logN - 1 - bit);
        const tw_t f = make_tw(pw[r], q), g = make_tw(pwi[r], q);
        W[k] = twd_t{f.w, f.wp};
        Wi[k] = twd_t{g.w, g.wp};
    }
    L->q = q;
    const u64 ninv = invmod_prime((u64)N % q, q);
    L->ninv = make_tw(ninv, q);
    L->w1inv_ninv = make_tw(mulmod_slow(N > 1 ? Wi[1].w : 1, ninv, q), q);
    L->br = make_barrett(q);
    L->W = W.data();
    L->Winv = Wi.data();
    // fp64 variant: every table value is an exact integer < 2^51 held in a double
    L->pd = (double)q;
    L->pinvd = 1.0 / (double)q;
    L->Wd = nullptr;
    L->Winvd = nullptr;
    if (q < TFHE_FP_QMAX && Wd && Wid) {
        Wd->resize((size_t)N);
        Wid->resize((size_t)N);
        for (int64_t k = 0; k < N; k++) {
            (*Wd)[k] = (double)W[k].w;
            (*Wid)[k] = (double)Wi[k].w;
        }
        L->ninv_d = ftw_t{(double)L->ninv.w};
        L->w1inv_ninv_d = ftw_t{(double)L->w1inv_ninv.w};
        L->Wd = Wd->data();
        L->Winvd = Wid->data();
    }
    return 0;
}
