// bfv_core.h -- per-coefficient bodies of BFV mul_expand / mul_contract.
//
// Reference semantics (exact integers, src/bfv.jl):
//   mul_expand  (:34, switch :222-226, switchel :202-220): x in [0,q) -> centred lift -> mod each prime of ℛbig
//   mul_contract(:35-40): per coefficient of ℛbig
//        z = t*y mod Qbig                         (SignedMod * Integer, signedmod.jl:24-28)
//        v = centred(z)                           (signedmod.jl:12-19)
//        w = div(v, q, RoundNearestTiesAway)      (multround :172-174, div_hacks.jl:120-135)
//        result_i = w mod q_i                     (oftype + switch back, :183-189, :222-226)
// q and Qbig are odd, so v/q is never a tie and w = floor((v + h)/q) with h = (q-1)/2.
// With r = (v + h) mod q in [0,q):  w = (v + h - r)/q exactly, which is evaluated limb-wise on the
// primes coprime to q (division by q is exact there), then lifted back centred; |w| <= (Qbig/q - 1)/2.
//   superset mode (ℛbig = ℛ ∪ P):  v ≡ z (mod q_i) and (mod p_j), so r_i = z_i + h_i; two exact
//        conversions per coefficient: r: basis(q) -> P (plain), w: P -> basis(q) (centred).
//   disjoint mode (test/bfv_crt.jl:9-25): r needs v mod q_i first (centred lift from Qbig); three
//        conversions: z: big -> small (centred), r: small -> big (plain), w: big -> small (centred).
#pragma once
#include "conv_core.h"

struct bfv_tab_t {
    int ns, nb, np;   // limbs of ℛ, of ℛbig, and of P = ℛbig \ ℛ (superset mode)
    int mode;         // 0 superset, 1 disjoint
    conv_tab_t E;     // expand: basis(ℛ) -> all of ℛbig, centred
    conv_tab_t C1;    // superset: basis(ℛ) -> P, plain      | disjoint: basis(ℛ) -> ℛbig, plain
    conv_tab_t C2;    // superset: P -> basis(ℛ), centred     | disjoint: ℛbig -> basis(ℛ), centred
    // per ℛbig limb j
    u64 p[TFHE_MAX_LIMBS];
    tw_t tmul[TFHE_MAX_LIMBS];  // t mod p_j
    u64 h[TFHE_MAX_LIMBS];      // (q-1)/2 mod p_j
    tw_t qinv[TFHE_MAX_LIMBS];  // q^-1 mod p_j (only for limbs coprime to q)
    // per ℛ limb i
    u64 qs[TFHE_MAX_LIMBS];
    u64 hs[TFHE_MAX_LIMBS];     // (q-1)/2 mod q_i
    int pos_s[TFHE_MAX_LIMBS];  // superset: index in ℛbig of ℛ limb i
    int pos_p[TFHE_MAX_LIMBS];  // superset: index in ℛbig of P limb (C1 target order, C2 source order)
};

// src/dst: limb l of this coefficient is at [l * lstride].  xi: scratch of >= max(ns, nb) words, element
// l at xi[l * xs] (LDS column of the thread on the GPU).
TFHE_HD void bfv_expand_coeff(const bfv_tab_t& B, const u64* src, size_t lstride_s, u64* dst, size_t lstride_d, u64* xi,
                              int xs) {
    for (int i = 0; i < B.ns; i++) xi[(size_t)i * xs] = src[(size_t)i * lstride_s];
    const u32 alpha = conv_prepare(B.E, xi, xs, true);
    for (int j = 0; j < B.nb; j++) dst[(size_t)j * lstride_d] = conv_eval(B.E, xi, xs, j, alpha, true);
}

// zb: scratch of nb words, rb: scratch of ns words (same striding as xi).
TFHE_HD void bfv_contract_coeff(const bfv_tab_t& B, const u64* src, size_t lstride_s, u64* dst, size_t lstride_d, u64* xi,
                                u64* zb, u64* rb, int xs) {
    for (int j = 0; j < B.nb; j++) zb[(size_t)j * xs] = shoup_full(src[(size_t)j * lstride_s], B.tmul[j], B.p[j]);
    if (B.mode == 0) {
        for (int i = 0; i < B.ns; i++) {
            const int pj = B.pos_s[i];
            xi[(size_t)i * xs] = addmod(zb[(size_t)pj * xs], B.h[pj], B.p[pj]);
        }
        const u32 a1 = conv_prepare(B.C1, xi, xs, false);
        for (int jp = 0; jp < B.np; jp++) {
            const int pj = B.pos_p[jp];
            const u64 pm = B.p[pj];
            const u64 r = conv_eval(B.C1, xi, xs, jp, a1, false);
            const u64 num = submod(addmod(zb[(size_t)pj * xs], B.h[pj], pm), r, pm);
            zb[(size_t)pj * xs] = shoup_full(num, B.qinv[pj], pm);
        }
        for (int jp = 0; jp < B.np; jp++) xi[(size_t)jp * xs] = zb[(size_t)B.pos_p[jp] * xs];
        const u32 a2 = conv_prepare(B.C2, xi, xs, true);
        for (int i = 0; i < B.ns; i++) dst[(size_t)i * lstride_d] = conv_eval(B.C2, xi, xs, i, a2, true);
    } else {
        for (int j = 0; j < B.nb; j++) xi[(size_t)j * xs] = zb[(size_t)j * xs];
        const u32 a0 = conv_prepare(B.C2, xi, xs, true);
        for (int i = 0; i < B.ns; i++)
            rb[(size_t)i * xs] = addmod(conv_eval(B.C2, xi, xs, i, a0, true), B.hs[i], B.qs[i]);
        for (int i = 0; i < B.ns; i++) xi[(size_t)i * xs] = rb[(size_t)i * xs];
        const u32 a1 = conv_prepare(B.C1, xi, xs, false);
        for (int j = 0; j < B.nb; j++) {
            const u64 pm = B.p[j];
            const u64 r = conv_eval(B.C1, xi, xs, j, a1, false);
            const u64 num = submod(addmod(zb[(size_t)j * xs], B.h[j], pm), r, pm);
            zb[(size_t)j * xs] = shoup_full(num, B.qinv[j], pm);
        }
        for (int j = 0; j < B.nb; j++) xi[(size_t)j * xs] = zb[(size_t)j * xs];
        const u32 a2 = conv_prepare(B.C2, xi, xs, true);
        for (int i = 0; i < B.ns; i++) dst[(size_t)i * lstride_d] = conv_eval(B.C2, xi, xs, i, a2, true);
    }
}
