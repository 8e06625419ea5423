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
