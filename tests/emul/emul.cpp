// tests/emul/emul.cpp -- CPU emulation of the HIP kernels' per-thread bodies.
//
// TEST INFRASTRUCTURE ONLY: never loaded by the product package (toyfhe.jl_amd/), which fails loudly
// without its HIP library.  There is no GPU in the build container, so this file runs the very same
// device headers (ntt_core.h, conv_core.h, bfv_core.h) on the host -- one loop iteration per thread
// id, one loop boundary per __syncthreads() -- to check index logic, lazy-range bounds and the exact
// base-conversion slow path against the oracle before a kernel ever reaches the MI355X.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define TFHE_EMUL_COUNT_SLOW 1
static long g_slow_hits = 0;
static double g_fp_max_ratio = 0;  // fp64arith.h TFHE_TRACK
#include "../../toyfhe.jl_amd/csrc/bfv_tables.h"
#include "../../toyfhe.jl_amd/csrc/ntt_tables.h"
#include "../../toyfhe.jl_amd/csrc/ckks_core.h"

namespace {


template <class A, int LOGB, int LOGT, int S0>
void fwd_sched(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 pre, int x, u32 sbrev) {
    constexpr int K = pass_k_fwd(LOGB, LOGT, S0);
    constexpr bool LAST = (S0 + K == LOGB);
    for (u32 tid = 0; tid < (1u << LOGT); tid++)
        ntt_fwd_pass<A, LOGB, LOGT, S0, K, S0 == 0, LAST>(lds, gsrc, gdst, C, tid, pre, x, sbrev);
    if constexpr (!LAST) fwd_sched<A, LOGB, LOGT, S0 + K>(lds, gsrc, gdst, C, pre, x, sbrev);
}
template <class A, int LOGB, int LOGT, int SEND, bool SCALE>
void inv_sched(u64* lds, const u64* gsrc, u64* gdst, const typename A::ctx& C, u32 pre, int x, u32 sbrev) {
    constexpr int K = pass_k_inv(LOGB, LOGT, SEND);
    constexpr int S0 = SEND - K;
    for (u32 tid = 0; tid < (1u << LOGT); tid++)
        ntt_inv_pass<A, LOGB, LOGT, S0, K, SEND == LOGB, S0 == 0, SCALE>(lds, gsrc, gdst, C, tid, pre, x, sbrev);
    if constexpr (S0 != 0) inv_sched<A, LOGB, LOGT, S0, SCALE>(lds, gsrc, gdst, C, pre, x, sbrev);
}

template <class A, int LOGB>
void block_fwd_a(const u64* src, u64* dst, const ntt_limb_t& L, int x) {
    std::vector<u64> lds(lds_words<LOGB, logt_for(LOGB)>());
    const typename A::ctx C = A::make(L);
    for (u32 sb = 0; sb < (1u << x); sb++)
        fwd_sched<A, LOGB, logt_for(LOGB), 0>(lds.data(), src + ((size_t)sb << LOGB), dst, C, (1u << x) + sb, x, brev_bits(sb, x));
}
template <class A, int LOGB>
void block_inv_a(const u64* src, u64* dst, const ntt_limb_t& L, int x) {
    std::vector<u64> lds(lds_words<LOGB, logt_for(LOGB)>());
    const typename A::ctx C = A::make(L);
    for (u32 sb = 0; sb < (1u << x); sb++) {
        if (x == 0) inv_sched<A, LOGB, logt_for(LOGB), LOGB, true>(lds.data(), src, dst, C, 1u, 0, 0u);
        else inv_sched<A, LOGB, logt_for(LOGB), LOGB, false>(lds.data(), src, dst + ((size_t)sb << LOGB), C, (1u << x) + sb, x, brev_bits(sb, x));
    }
}
// fp64 butterflies when the modulus qualifies and the caller did not force the u64 path (mirrors sel_fp in toyfhe_hip.hip)
static bool g_force_int = false;
static bool g_small = false;   // variant 3: the range plan of the size class below 2^42 (ArithFpS)
template <int LOGB>
void block_fwd(const u64* src, u64* dst, const ntt_limb_t& L, int x) {
    if (L.Wd && g_small) block_fwd_a<ArithFpS, LOGB>(src, dst, L, x);
    else if (L.Wd && !g_force_int) block_fwd_a<ArithFp, LOGB>(src, dst, L, x); else block_fwd_a<ArithInt, LOGB>(src, dst, L, x);
}
template <int LOGB>
void block_inv(const u64* src, u64* dst, const ntt_limb_t& L, int x) {
    if (L.Wd && g_small) block_inv_a<ArithFpS, LOGB>(src, dst, L, x);
    else if (L.Wd && !g_force_int) block_inv_a<ArithFp, LOGB>(src, dst, L, x); else block_inv_a<ArithInt, LOGB>(src, dst, L, x);
}

template <int X>
void top_fwd(const u64* src, u64* dst, const ntt_limb_t& L, int logn) {
    const u64 stride = (u64)1 << (logn - X);
    for (u64 col = 0; col < stride; col++) ntt_fwd_top<X>(src, dst, L.W, L.q, col, stride);
}
template <int X>
void top_inv(const u64* src, u64* dst, const ntt_limb_t& L, int logn) {
    const u64 stride = (u64)1 << (logn - X);
    for (u64 col = 0; col < stride; col++) ntt_inv_top<X>(src, dst, L, col, stride);
}

}  // namespace

extern "C" {

// one limb-polynomial transform; variant 0 = register-blocked path (logn >= 10; fp64 butterflies when the
// modulus qualifies), 1 = generic radix-2, 2 = register-blocked path with u64 butterflies forced, 3 = register-blocked path
// with the fp64 range plan of moduli below 2^42 (ArithFpS).
// returns 0, -1 on bad psi, -2 on unsupported size
int emul_ntt(int logn, uint64_t q, uint64_t psi, int inverse, int variant, const uint64_t* src, uint64_t* dst) {
    const int64_t N = 1ll << logn;
    if (!psi) psi = hostmath::minimal_primitive_root(q, 2 * (u64)N);
    ntt_host_tabs_t HT;
    ntt_limb_t L;
    if (build_ntt_tables_all(N, q, psi, HT, &L)) return -1;
    g_force_int = (variant == 2);
    g_small = (variant == 3);
    if (g_small && q >= TFHE_FPS_QMAX) return -2;
    if (variant == 1 || logn < 10) {
        if (logn > 14) return -2;
        std::vector<u64> lds((size_t)N);
        if (!inverse) {
            for (int64_t i = 0; i < N; i++) lds[i] = src[i];
            for (int s = 0; s < logn; s++)
                for (u32 b = 0; b < (u32)(N / 2); b++) ntt_generic_fwd_stage(lds.data(), L.W, L.q, logn, s, b);
            for (int64_t i = 0; i < N; i++) dst[i] = csub(csub(lds[brev_bits((u32)i, logn)], 2 * q), q);
        } else {
            for (int64_t i = 0; i < N; i++) lds[brev_bits((u32)i, logn)] = src[i];
            for (int s = logn - 1; s >= 0; s--)
                for (u32 b = 0; b < (u32)(N / 2); b++) ntt_generic_inv_stage(lds.data(), L, logn, s, b);
            for (int64_t i = 0; i < N; i++) dst[i] = csub(lds[i], q);
        }
        return 0;
    }
    std::vector<u64> in(src, src + N), tmp((size_t)N), out((size_t)N);
    if (logn <= 14) {
        switch (logn) {
#define C_(LB) case LB: if (inverse) block_inv<LB>(in.data(), out.data(), L, 0); else block_fwd<LB>(in.data(), out.data(), L, 0); break;
            C_(10) C_(11) C_(12) C_(13) C_(14)
#undef C_
        }
    } else {
        const int x = logn - 14;
        if (x > 3) return -2;
        if (!inverse) {
            if (x == 1) top_fwd<1>(in.data(), tmp.data(), L, logn); else if (x == 2) top_fwd<2>(in.data(), tmp.data(), L, logn); else top_fwd<3>(in.data(), tmp.data(), L, logn);
            block_fwd<14>(tmp.data(), out.data(), L, x);
        } else {
            block_inv<14>(in.data(), tmp.data(), L, x);
            if (x == 1) top_inv<1>(tmp.data(), out.data(), L, logn); else if (x == 2) top_inv<2>(tmp.data(), out.data(), L, logn); else top_inv<3>(tmp.data(), out.data(), L, logn);
        }
    }
    memcpy(dst, out.data(), N * 8);
    return 0;
}

// exact conversion of `count` coefficients: res [count][k] -> out [count][m]; returns slow-path hits
long emul_conv(const uint64_t* a, int k, const uint64_t* t, int m, int centred, const uint64_t* res, uint64_t* out, long count) {
    conv_host_t H;
    build_conv_host(std::vector<u64>(a, a + k), std::vector<u64>(t, t + m), &H);
    g_slow_hits = 0;
    std::vector<u64> xi(k);
    for (long c = 0; c < count; c++) {
        for (int j = 0; j < k; j++) xi[j] = res[c * k + j];
        const u32 alpha = conv_prepare(H.tab, xi.data(), 1, centred != 0);
        for (int i = 0; i < m; i++) out[c * m + i] = conv_eval(H.tab, xi.data(), 1, i, alpha, centred != 0);
    }
    return g_slow_hits;
}

// base-2^w digits of `count` coefficients: res [count][k] -> out [count][nwin] (conv_core.h window_digits_coeff)
void emul_window_digits(const uint64_t* a, int k, int wbits, int nwin, const uint64_t* res, uint64_t* out, long count) {
    conv_host_t H;
    build_conv_host(std::vector<u64>(a, a + k), std::vector<u64>(), &H);
    std::vector<u64> d((size_t)nwin * k);
    for (long c = 0; c < count; c++) {
        window_digits_coeff(k > 1 ? &H.tab : nullptr, res + c * k, 1, k, wbits, nwin, d.data(), (size_t)k, 1);
        for (int i = 0; i < nwin; i++) out[c * nwin + i] = d[(size_t)i * k + (k - 1)];  // every limb holds the same digit
    }
}

// BFV expand / contract on [count][limbs][N] buffers; returns 0 or the table-construction error code
int emul_bfv(const uint64_t* qs, int ns, const uint64_t* pb, int nb, uint64_t t, int contract, int64_t N,
             const uint64_t* src, uint64_t* dst, long count, long* slow_hits) {
    bfv_host_t* H = new bfv_host_t();
    std::string err;
    int rc = build_bfv_host(std::vector<u64>(qs, qs + ns), std::vector<u64>(pb, pb + nb), t, H, &err);
    if (rc) { delete H; return rc; }
    g_slow_hits = 0;
    std::vector<u64> xi(nb + ns), zb(nb), rb(ns);
    for (long p = 0; p < count; p++)
        for (int64_t k = 0; k < N; k++) {
            if (!contract) bfv_expand_coeff(H->tab, src + p * ns * N + k, N, dst + p * nb * N + k, N, xi.data(), 1);
            else bfv_contract_coeff(H->tab, src + p * nb * N + k, N, dst + p * ns * N + k, N, xi.data(), zb.data(), rb.data(), 1);
        }
    if (slow_hits) *slow_hits = g_slow_hits;
    delete H;
    return 0;
}

// fast (register-resident, folded) BFV expand / contract; returns -9 if (ns, np) has no instantiation
// contract == 2: the narrow contraction on tensor rows given as reduced doubles (bit patterns), k_bfv_core_fused<.., OUTD>'s form
int emul_bfv_fast(const uint64_t* qs, int ns, const uint64_t* pb, int nb, uint64_t t, int contract, int64_t N,
                  const uint64_t* src, uint64_t* dst, long count) {
    bfv_fast_host_t* H = new bfv_fast_host_t();
    if (!build_bfv_fast_host(std::vector<u64>(qs, qs + ns), std::vector<u64>(pb, pb + nb), t, H)) { delete H; return -7; }
    const int np = H->tab.np;
    int rc = 0;
    for (long p = 0; p < count && !rc; p++)
        for (int64_t k = 0; k < N; k++) {
            const u64* s = contract ? src + p * nb * N + k : src + p * ns * N + k;
            u64* d = contract ? dst + p * ns * N + k : dst + p * nb * N + k;
#define FAST_(S, P_)                                                                           \
    else if (ns == S && np == P_) {                                                            \
        if (H->tab.narrow) { u64 col[TFHE_FAST_MAX]; if (contract == 2) bfv_contract_narrow<S, P_, true>(H->tab, s, N, d, N, col, 1); else if (contract) bfv_contract_narrow<S, P_>(H->tab, s, N, d, N, col, 1); else bfv_expand_narrow<S, P_>(H->tab, s, N, d, N, col, 1); } \
        else { if (contract) bfv_contract_fast<S, P_, false>(H->tab, s, N, d, N); else bfv_expand_fast<S, P_, false>(H->tab, s, N, d, N); } \
    }
            if (false) {}
            FAST_(8, 9) FAST_(3, 4) FAST_(2, 3) FAST_(6, 7)
            else { rc = -9; break; }
#undef FAST_
        }
    delete H;
    return rc;
}

}  // extern "C"

// largest |value| / p seen by the fp64 modular products / reductions since the last call (range-budget check)
extern "C" double emul_fp_max_ratio_reset() {
    const double r = g_fp_max_ratio;
    g_fp_max_ratio = 0;
    return r;
}

// ckks_core.h: n = round(x * smant * 2^sexp) mod q for `count` doubles (encode tail), and the inverse conversion of a
// signed multi-word magnitude to a double (decode head)
extern "C" void emul_ckks_round(const double* x, long count, uint64_t smant, int sexp, uint64_t q, uint64_t* out) {
    const barrett_t bt = hostmath::make_barrett(q);
    for (long i = 0; i < count; i++) out[i] = ckks_residue(ckks_round_scaled(x[i], smant, sexp), bt);
}
extern "C" double emul_ckks_to_double(const uint64_t* w, int nwords, int neg, uint64_t smant, int sexp) {
    return ckks_words_to_double(w, nwords, neg != 0, smant, sexp);
}

// bfv_fast.h centred_double_bits: the centred residue as the exact double (the form c2 takes between the BFV contraction
// and the fused key switch)
extern "C" double emul_centred_double(uint64_t r, uint64_t q) {
    const uint64_t bits = centred_double_bits(r, q);
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

// ntt_core.h ArithFpMD::out_moddown: the ModulusRaised contraction + "+ c" in the fused key switch's final store, on one lazy
// inverse-transform value v (an exact integer double, |v| < 7.9 p), the special limb's coefficient tsp and the addend word cw
extern "C" uint64_t emul_out_moddown(double v, uint64_t tsp, uint64_t cw, uint64_t q, uint64_t pinv_modq) {
    ArithFp::ctx C{};
    C.p = (double)q;
    C.pinv = 1.0 / (double)q;
    C.q = q;
    C.md_pinv = (double)pinv_modq;
    return ArithFpMD::out_moddown(v, tsp, cw, C);
}
