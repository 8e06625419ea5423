// ntt_tables.h -- host construction of the per-limb NTT tables (pure host C++, shared by
// toyfhe_hip.hip and tests/emul/).  W[k] = psi^brv(k), Winv[k] = psi^-brv(k), each with its Shoup
// companion; psi is the ring's 2N-th root (src/pow2_cyc_rings.jl:27-44).
#pragma once
#include <vector>

#include "host_math.h"
#include "ntt_core.h"

// returns 0, or -1 if psi is not a primitive 2N-th root of unity mod q
// permutation of the boundary-pass stages (see ntt_limb_t::Wb): for stage s >= logN - Kb, d = s - (logN - Kb),
// position (2^s + (c0 << d) + g) holds entry (2^s + (brv_{logN-Kb}(c0) << d) + g)
template <class T>
inline void permute_boundary(const std::vector<T>& in, std::vector<T>& out, int logN, int Kb) {
    out = in;
    const int hb = logN - Kb;  // bits of the group index
    for (int d = 0; d < Kb; d++) {
        const int s = hb + d;
        for (u32 c0 = 0; c0 < (1u << hb); c0++) {
            u32 r = 0;
            for (int b = 0; b < hb; b++) r |= ((c0 >> b) & 1u) << (hb - 1 - b);
            for (u32 g = 0; g < (1u << d); g++) out[((size_t)1 << s) + ((size_t)c0 << d) + g] = in[((size_t)1 << s) + ((size_t)r << d) + g];
        }
    }
}

// the same for the 2^logb-point sub-blocks of a 2^(logb+x)-point transform: local stage hb + d of sub-block prefix pre (2^x <= pre <
// 2^(x+1)) reads entries (pre << (hb + d)) + (hi << d) + g; the permuted copy holds them at (pre << (hb + d)) + (c0 << d) + g with
// hi = brv_hb(c0).  Before r04 the boundary passes of every N > 2^14 kernel loaded their twiddles through the bit-reversed index:
// 64 different cache lines per wave-load (k_ntt_fwd_quad fetched 1.6 x its algorithmic bytes, the one-pass inverse with a permuted
// table 1.1 x).
template <class T>
inline void permute_boundary_sub(const std::vector<T>& in, std::vector<T>& out, int logb, int x, int Kb) {
    out = in;
    const int hb = logb - Kb;
    for (u32 pre = 1u << x; pre < (2u << x); pre++)
        for (int d = 0; d < Kb; d++) {
            const size_t base = (size_t)pre << (hb + d);
            for (u32 c0 = 0; c0 < (1u << hb); c0++) {
                u32 r = 0;
                for (int b = 0; b < hb; b++) r |= ((c0 >> b) & 1u) << (hb - 1 - b);
                for (u32 g = 0; g < (1u << d); g++) out[base + ((size_t)c0 << d) + g] = in[base + ((size_t)r << d) + g];
            }
        }
}

struct ntt_host_tabs_t {  // every table of one limb (host copies)
    std::vector<twd_t> W, Wi, Wb, Wib;
    std::vector<ftwd_t> Wd, Wid, Wdb, Widb;
    std::vector<ftwd_t> I2t0, I2wib;  // N = 2^16, fp64 policy: k_ntt_inv_quad2 (ntt_limb_t::i2_*)
};

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
    L->Wb = nullptr; L->Winvb = nullptr; L->Wdb = nullptr; L->Winvdb = nullptr;
    L->i2_t0 = nullptr; L->i2_winvb = nullptr; L->i2_iinv = 0.0; L->i2_g[0] = L->i2_g[1] = L->i2_g[2] = 0.0;
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

// all tables of a limb, including the boundary-permuted copies for the register-blocked geometry of this N
inline int build_ntt_tables_all(int64_t N, u64 q, u64 psi, ntt_host_tabs_t& T, ntt_limb_t* L) {
    int rc = build_ntt_tables(N, q, psi, T.W, T.Wi, L, &T.Wd, &T.Wid);
    if (rc) return rc;
    int logN = 0;
    while ((1ll << logN) < N) logN++;
    if (logN >= 10 && logN <= 14) {
        const int logt = logt_for(logN);
        // forward last pass and inverse first pass cover the same stages: K = boundary pass width
        int s0 = 0, Kb = 0;
        while (s0 < logN) { Kb = pass_k_fwd(logN, logt, s0); s0 += Kb; }
        if (Kb == pass_k_inv(logN, logt, logN)) {
            permute_boundary(T.W, T.Wb, logN, Kb);
            permute_boundary(T.Wi, T.Wib, logN, Kb);
            L->Wb = T.Wb.data(); L->Winvb = T.Wib.data();
            if (L->Wd) {
                permute_boundary(T.Wd, T.Wdb, logN, Kb);
                permute_boundary(T.Wid, T.Widb, logN, Kb);
                L->Wdb = T.Wdb.data(); L->Winvdb = T.Widb.data();
            }
        }
    }
    if (logN > 14 && logN <= 17) {
        // N > 2^14: 2^14-point sub-blocks (LOGB = 14 geometry) behind / in front of x = logN - 14 top stages
        constexpr int LB = 14;
        const int x = logN - LB, logt = logt_for(LB);
        int s0 = 0, Kb = 0;
        while (s0 < LB) { Kb = pass_k_fwd(LB, logt, s0); s0 += Kb; }
        if (Kb == pass_k_inv(LB, logt, LB)) {
            permute_boundary_sub(T.W, T.Wb, LB, x, Kb);
            permute_boundary_sub(T.Wi, T.Wib, LB, x, Kb);
            L->Wb = T.Wb.data(); L->Winvb = T.Wib.data();
            if (L->Wd) {
                permute_boundary_sub(T.Wd, T.Wdb, LB, x, Kb);
                permute_boundary_sub(T.Wid, T.Widb, LB, x, Kb);
                L->Wdb = T.Wdb.data(); L->Winvdb = T.Widb.data();
            }
        }
    }
    if (logN == 16 && L->Wd) {
        // k_ntt_inv_quad2 (kernels.h): the two top stages first, then four 2^14-point inverses over psi^4, whose tables are the
        // first quarter of this limb's.  LOGB = 14, LOGT = 9 geometry.
        using namespace hostmath;
        constexpr int LB = 14, LT_ = logt_for(LB);
        const u64 pinv = invmod_prime(psi, q);
        T.I2t0.resize((size_t)3 << LT_);
        for (int c = 1; c <= 3; c++) {
            const u64 step = powmod(pinv, (u64)2 * c, q);               // psi^{-2c}: from point t to t + 1
            u64 v = powmod(pinv, (u64)c, q);                             // psi^{-c (2t + 1)} at t = 0
            for (u32 t = 0; t < (1u << LT_); t++) {
                T.I2t0[((size_t)(c - 1) << LT_) + t] = (double)v;
                v = mulmod_slow(v, step, q);
            }
            L->i2_g[c - 1] = (double)powmod(pinv, (u64)c << (LT_ + 1), q);   // psi^{-c 2^(LOGT+1)}: from point k' to k' + 2^LOGT
        }
        L->i2_iinv = (double)powmod(pinv, (u64)N / 2, q);
        const int Kb = pass_k_inv(LB, LT_, LB);
        std::vector<ftwd_t> first(T.Wid.begin(), T.Wid.begin() + ((size_t)1 << LB));
        permute_boundary(first, T.I2wib, LB, Kb);
        L->i2_t0 = T.I2t0.data();
        L->i2_winvb = T.I2wib.data();
    }
    return 0;
}
