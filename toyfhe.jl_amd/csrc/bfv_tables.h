// bfv_tables.h -- host construction of the exact base-conversion tables (conv_core.h) and of the BFV
// expand/contract table (bfv_core.h).  Pure host C++ (no HIP): toyfhe_hip.hip uploads the vectors and
// patches the pointers; tests/emul/ runs the same tables against the same per-coefficient code on CPU.
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "bfv_core.h"
#include "host_math.h"

struct conv_host_t {
    conv_tab_t tab;  // C/M/Aw point into the vectors below
    std::vector<u64> C, M, Aw;
};

inline void build_conv_host(const std::vector<u64>& a, const std::vector<u64>& t, conv_host_t* H) {
    using namespace hostmath;
    conv_tab_t* T = &H->tab;
    memset(T, 0, sizeof *T);
    const int k = (int)a.size(), m = (int)t.size();
    T->k = k;
    T->m = m;
    bigint A = big_from(1);
    for (u64 x : a) A = big_mul_u64(A, x);
    const int nw = (int)A.size();
    T->nwords = nw;
    const bigint half = big_shr1(A);  // floor(A/2); A is odd
    int maxbits = 0;
    H->C.assign((size_t)k * m, 0);
    H->M.assign((size_t)k * nw, 0);
    H->Aw.assign(A.begin(), A.end());
    for (int j = 0; j < k; j++) {
        bigint Mj = big_from(1);
        for (int l = 0; l < k; l++)
            if (l != j) Mj = big_mul_u64(Mj, a[l]);
        for (size_t w = 0; w < Mj.size(); w++) H->M[(size_t)j * nw + w] = Mj[w];
        T->a[j] = a[j];
        T->inv[j] = make_tw(invmod_prime(big_mod_u64(Mj, a[j]), a[j]), a[j]);
        T->half[j] = big_mod_u64(half, a[j]);
        const int sh = __builtin_clzll(a[j]);
        T->sh[j] = (u32)sh;
        T->rho[j] = (u64)((~(u128)0) / ((u128)(a[j] << sh)) - (((u128)1) << 64));
        maxbits = std::max(maxbits, bitlen(a[j]));
        for (int i = 0; i < m; i++) H->C[(size_t)j * m + i] = big_mod_u64(Mj, t[i]);
    }
    for (int i = 0; i < m; i++) {
        T->t[i] = make_barrett(t[i]);
        T->Amod[i] = big_mod_u64(A, t[i]);
        T->halfT[i] = big_mod_u64(half, t[i]);
        T->copy_from[i] = -1;
        for (int j = 0; j < k; j++)
            if (a[j] == t[i]) T->copy_from[i] = j;
    }
    const int room = 62 - maxbits;  // products that fit the Barrett window: 2^(62 - bits(a))
    T->lazy = std::max(1, std::min(k, room >= 20 ? (1 << 20) : (1 << std::max(0, room))));
    T->C = H->C.data();
    T->M = H->M.data();
    T->Aw = H->Aw.data();
}

struct bfv_host_t {
    bfv_tab_t tab;
    conv_host_t E, C1, C2;
};

// qs = primes of ℛ, pb = primes of ℛbig (each in buffer limb order).  Returns 0, or a negative code with *err set:
// -7 unsupported basis relation, -1 bad argument.
inline int build_bfv_host(const std::vector<u64>& qs, const std::vector<u64>& pb, u64 t, bfv_host_t* H, std::string* err) {
    using namespace hostmath;
    const int ns = (int)qs.size(), nb = (int)pb.size();
    bfv_tab_t& B = H->tab;
    memset(&B, 0, sizeof B);
    B.ns = ns;
    B.nb = nb;
    std::vector<int> pos_s(ns, -1);
    int found = 0;
    for (int i = 0; i < ns; i++)
        for (int j = 0; j < nb; j++)
            if (pb[j] == qs[i]) { pos_s[i] = j; found++; }
    if (found == ns) B.mode = 0;
    else if (found == 0) B.mode = 1;
    else { *err = "ℛbig must contain all primes of ℛ or none of them"; return -7; }
    if (B.mode == 0 && nb == ns) { *err = "ℛbig equals ℛ: no room for the tensor product"; return -1; }
    bigint q = big_from(1);
    for (u64 x : qs) q = big_mul_u64(q, x);
    const bigint hq = big_shr1(q);  // (q-1)/2
    std::vector<u64> P;             // primes of ℛbig coprime to q
    for (int j = 0; j < nb; j++) {
        B.p[j] = pb[j];
        B.tmul[j] = make_tw(t % pb[j], pb[j]);
        B.h[j] = big_mod_u64(hq, pb[j]);
        const u64 qm = big_mod_u64(q, pb[j]);
        if (qm) {
            B.qinv[j] = make_tw(invmod_prime(qm, pb[j]), pb[j]);
            if (B.mode == 0) { B.pos_p[(int)P.size()] = j; P.push_back(pb[j]); }
        }
    }
    for (int i = 0; i < ns; i++) { B.qs[i] = qs[i]; B.hs[i] = big_mod_u64(hq, qs[i]); B.pos_s[i] = pos_s[i]; }
    B.np = (int)P.size();
    build_conv_host(qs, pb, &H->E);
    if (B.mode == 0) { build_conv_host(qs, P, &H->C1); build_conv_host(P, qs, &H->C2); }
    else { build_conv_host(qs, pb, &H->C1); build_conv_host(pb, qs, &H->C2); }
    B.E = H->E.tab;
    B.C1 = H->C1.tab;
    B.C2 = H->C2.tab;
    return 0;
}

// ---- folded tables of the register-resident fast path (bfv_fast.h), superset mode only ----
#include "bfv_fast.h"
struct bfv_fast_host_t {
    bfv_fast_tab_t tab;
    std::vector<u64> Mq, Aq, Mp, Ap;
};
// returns false when the configuration is outside the fast path (then the general kernels are used)
inline bool build_bfv_fast_host(const std::vector<u64>& qs, const std::vector<u64>& pb, u64 t, bfv_fast_host_t* H) {
    using namespace hostmath;
    const int ns = (int)qs.size(), nb = (int)pb.size();
    bfv_fast_tab_t& B = H->tab;
    memset(&B, 0, sizeof B);
    std::vector<int> pos_s(ns, -1), pos_p;
    for (int i = 0; i < ns; i++)
        for (int j = 0; j < nb; j++)
            if (pb[j] == qs[i]) pos_s[i] = j;
    for (int i = 0; i < ns; i++)
        if (pos_s[i] < 0) return false;
    std::vector<u64> P;
    for (int j = 0; j < nb; j++) {
        bool shared = false;
        for (int i = 0; i < ns; i++) shared |= (pos_s[i] == j);
        if (!shared) { pos_p.push_back(j); P.push_back(pb[j]); }
    }
    const int np = (int)P.size();
    if (np < 1 || ns > TFHE_FAST_MAX || np > TFHE_FAST_MAX) return false;
    B.ns = ns; B.np = np; B.nb = nb;
    bigint q = big_from(1), Pb = big_from(1);
    for (u64 x : qs) q = big_mul_u64(q, x);
    for (u64 x : P) Pb = big_mul_u64(Pb, x);
    const bigint hq = big_shr1(q), hP = big_shr1(Pb);  // (q-1)/2 = floor(q/2), floor(P/2)
    std::vector<bigint> Qi(ns), Pj(np);
    for (int i = 0; i < ns; i++) { Qi[i] = big_from(1); for (int l = 0; l < ns; l++) if (l != i) Qi[i] = big_mul_u64(Qi[i], qs[l]); }
    for (int j = 0; j < np; j++) { Pj[j] = big_from(1); for (int l = 0; l < np; l++) if (l != j) Pj[j] = big_mul_u64(Pj[j], P[l]); }
    int maxq = 0, maxp = 0;
    for (int i = 0; i < ns; i++) {
        const u64 qi = qs[i];
        B.pos_s[i] = pos_s[i];
        B.q[i] = qi;
        B.qb[i] = make_barrett(qi);
        const u64 inv = invmod_prime(big_mod_u64(Qi[i], qi), qi);
        B.e_inv[i] = make_tw(inv, qi);
        B.e_half[i] = big_mod_u64(hq, qi);
        const int sh = __builtin_clzll(qi);
        B.sh_q[i] = (u32)sh;
        B.rho_q[i] = (u64)((~(u128)0) / ((u128)(qi << sh)) - (((u128)1) << 64));
        B.c_a1[i] = make_tw(mulmod_slow(t % qi, inv, qi), qi);
        B.c_b1[i] = mulmod_slow(big_mod_u64(hq, qi), inv, qi);
        B.c_A2[i] = big_mod_u64(Pb, qi);
        B.c_halfT[i] = big_mod_u64(hP, qi);
        maxq = std::max(maxq, bitlen(qi));
    }
    for (int j = 0; j < np; j++) {
        const u64 pj = P[j];
        B.pos_p[j] = pos_p[j];
        B.pb[j] = make_barrett(pj);
        const u64 qinv = invmod_prime(big_mod_u64(q, pj), pj);
        const u64 invP = invmod_prime(big_mod_u64(Pj[j], pj), pj);
        const u64 f = mulmod_slow(qinv, invP, pj);
        B.e_A[j] = big_mod_u64(q, pj);
        B.e_halfT[j] = big_mod_u64(hq, pj);
        B.c_a2[j] = make_tw(mulmod_slow(t % pj, f, pj), pj);
        const u64 hterm = mulmod_slow(big_mod_u64(hq, pj), qinv, pj);
        B.c_b2[j] = mulmod_slow((hterm + big_mod_u64(hP, pj)) % pj, invP, pj);
        B.c_A1[j] = invP;
        const int sh = __builtin_clzll(pj);
        B.sh_p[j] = (u32)sh;
        B.rho_p[j] = (u64)((~(u128)0) / ((u128)(pj << sh)) - (((u128)1) << 64));
        maxp = std::max(maxp, bitlen(pj));
        for (int i = 0; i < ns; i++) {
            const u64 c = big_mod_u64(Qi[i], pj);
            B.e_C[i][j] = c;
            B.c_C1[i][j] = mulmod_slow(c, f, pj);
            B.c_C2[j][i] = big_mod_u64(Pj[j], qs[i]);
        }
    }
    auto lazy_of = [](int bits, int k) { const int room = 62 - bits; return std::max(1, std::min(k, room >= 20 ? (1 << 20) : (1 << std::max(0, room)))); };
    u64 maxmod = 0;
    for (u64 x : qs) maxmod = std::max(maxmod, x);
    for (u64 x : P) maxmod = std::max(maxmod, x);
    B.narrow = (maxmod < TFHE_FP_QMAX && ns + 2 <= 16 && np + 2 <= 16) ? 1 : 0;
    if (B.narrow) {
        for (int i = 0; i < ns; i++) {
            B.f_q[i] = (double)qs[i]; B.f_qinv[i] = 1.0 / (double)qs[i];
            B.f_ea[i] = (double)B.e_inv[i].w; B.f_eb[i] = (double)mulmod_slow(B.e_half[i], B.e_inv[i].w, qs[i]);
            B.f_ca[i] = (double)B.c_a1[i].w; B.f_cb[i] = (double)B.c_b1[i];
        }
        for (int j = 0; j < np; j++) { B.f_p[j] = (double)P[j]; B.f_pinv[j] = 1.0 / (double)P[j]; }
        // every constant operand of a product sum carries the factor R = 2^78 that acc52_redc divides out
        auto neg = [](u64 c, u64 m) { return c ? m - c : 0; };
        auto mont = [](u64 m) {
            mont26_t M;
            M.pl = (u32)(m & 0x3ffffffu);
            M.ph = (u32)(m >> 26);
            u32 inv = 1;  // m^-1 mod 2^26 by Newton iteration (m odd)
            for (int it = 0; it < 5; it++) inv *= 2u - (u32)m * inv;
            M.pp = (0u - inv) & 0x3ffffffu;
            M.pad_ = 0;
            return M;
        };
        auto fold = [](u64 c, u64 m) { return mulmod_slow(c, powmod(2, TFHE_MONT26_RBITS, m), m); };
        for (int j = 0; j < np; j++) {
            const u64 pj = P[j];
            B.mp[j] = mont(pj);
            B.n_eNegA[j] = pack26(fold(neg(B.e_A[j], pj), pj));
            B.n_eNegHalf[j] = fold(neg(B.e_halfT[j], pj), pj);
            B.n_cA2[j] = pack26(fold(B.c_a2[j].w, pj));
            B.n_cA1[j] = pack26(fold(B.c_A1[j], pj));
            B.n_cB2[j] = fold(B.c_b2[j], pj);
            for (int i = 0; i < ns; i++) {
                B.n_eC[i][j] = pack26(fold(B.e_C[i][j], pj));
                B.n_cNegC1[i][j] = pack26(fold(neg(B.c_C1[i][j], pj), pj));
                B.n_cC2[j][i] = pack26(fold(B.c_C2[j][i], qs[i]));
            }
        }
        for (int i = 0; i < ns; i++)
            for (int j = 0; j < np; j++) { B.t_eC[j][i] = B.n_eC[i][j]; B.t_cNegC1[j][i] = B.n_cNegC1[i][j]; B.t_cC2[i][j] = B.n_cC2[j][i]; }
        for (int i = 0; i < ns; i++) {
            B.mq[i] = mont(qs[i]);
            B.n_cNegA2[i] = pack26(fold(neg(B.c_A2[i], qs[i]), qs[i]));
            B.n_cNegHalf[i] = fold(neg(B.c_halfT[i], qs[i]), qs[i]);
        }
    }
    B.lazy_q = lazy_of(maxq, ns);
    B.lazy_p = lazy_of(maxp, np);
    B.nwq = (int)q.size();
    B.nwp = (int)Pb.size();
    H->Mq.assign((size_t)ns * B.nwq, 0);
    H->Mp.assign((size_t)np * B.nwp, 0);
    for (int i = 0; i < ns; i++) for (size_t w = 0; w < Qi[i].size(); w++) H->Mq[(size_t)i * B.nwq + w] = Qi[i][w];
    for (int j = 0; j < np; j++) for (size_t w = 0; w < Pj[j].size(); w++) H->Mp[(size_t)j * B.nwp + w] = Pj[j][w];
    H->Aq.assign(q.begin(), q.end());
    H->Ap.assign(Pb.begin(), Pb.end());
    B.Mq = H->Mq.data(); B.Aq = H->Aq.data(); B.Mp = H->Mp.data(); B.Ap = H->Ap.data();
    return true;
}
