/*
 * oracle/ref_cpu.c -- CPU restatement of the ToyFHE.jl power-of-two-cyclotomic RNS path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load the library built from this file; the product (toyfhe.jl_amd/) never does.
 *
 * It follows the reference's *algorithmic structure* (per-limb psi-twist + radix-2 Cooley-Tukey
 * NTT, limb-wise modular ops, exact big-integer CRT for every basis change and rounding), so it
 * doubles as the CPU baseline that bench.py times ("kind": "port").  It is checked bit-for-bit
 * against oracle/spec.py (pure-Python big-int, by definition) which in turn is pinned to the
 * reference's documented known-answer vectors; see oracle/spec.py for the pinning status
 * (beyond the doc vectors: PARITY UNPINNED -- the reference cannot run here).
 *
 * File:line citations are relative to /root/reference/src/.
 *
 * Data layout everywhere: u64 residues, [count][limbs][N], limb-major SoA exactly like the
 * StructArray field arrays of crt.jl:150-156.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------ */
/* scalar modular arithmetic (GaloisFields.PrimeField stand-in; call sites pow2_cyc_rings.jl:3,15) */
/* ------------------------------------------------------------------------------------------ */

static inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

static u64 powmod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = mulmod(r, a, q);
        a = mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}
static u64 invmod(u64 a, u64 q) { return powmod(a, q - 2, q); } /* q prime */

/* Shoup constant floor(w * 2^64 / q) and multiplication by a precomputed constant */
static inline u64 shoup_pre(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
static inline u64 shoup_mul(u64 x, u64 w, u64 wp, u64 q) {
    u64 h = (u64)(((u128)x * wp) >> 64);
    u64 r = x * w - h * q;
    return r >= q ? r - q : r;
}

/* GaloisFields.minimal_primitive_root(F, n): numerically smallest element of exact order n
 * (n a power of two). Known answer (97, 8) -> 33, docs/src/man/background/rlwe.md:186. */
static u64 minimal_primitive_root(u64 q, u64 n) {
    u64 g = 2, z;
    for (;; g++) {
        z = powmod(g, (q - 1) / n, q);
        if (powmod(z, n / 2, q) == q - 1) break;
    }
    u64 best = z, z2 = mulmod(z, z, q), cur = z;
    for (u64 i = 1; i < n / 2; i++) {
        cur = mulmod(cur, z2, q);
        if (cur < best) best = cur;
    }
    return best;
}

/* ------------------------------------------------------------------------------------------ */
/* fixed-capacity unsigned big integers (BigInt stand-in; call sites crt.jl:98-112,
 * signedmod.jl:12-19, bfv.jl:172-226)                                                         */
/* ------------------------------------------------------------------------------------------ */

#define BW 48 /* words: 3072 bits */
typedef struct {
    u64 w[BW];
    int n;
} bn;

static void bn_norm(bn *a) { while (a->n > 0 && a->w[a->n - 1] == 0) a->n--; }
static void bn_set_u64(bn *a, u64 x) { memset(a, 0, sizeof(*a)); a->w[0] = x; a->n = x ? 1 : 0; }
static int bn_cmp(const bn *a, const bn *b) {
    if (a->n != b->n) return a->n < b->n ? -1 : 1;
    for (int i = a->n - 1; i >= 0; i--)
        if (a->w[i] != b->w[i]) return a->w[i] < b->w[i] ? -1 : 1;
    return 0;
}
static void bn_add(bn *r, const bn *a, const bn *b) {
    int n = a->n > b->n ? a->n : b->n;
    u64 c = 0;
    bn t; memset(&t, 0, sizeof(t));
    for (int i = 0; i < n; i++) {
        u128 s = (u128)(i < a->n ? a->w[i] : 0) + (i < b->n ? b->w[i] : 0) + c;
        t.w[i] = (u64)s; c = (u64)(s >> 64);
    }
    t.n = n;
    if (c) { t.w[n] = c; t.n = n + 1; }
    *r = t;
}
static void bn_sub(bn *r, const bn *a, const bn *b) { /* a >= b */
    bn t; memset(&t, 0, sizeof(t));
    u64 br = 0;
    for (int i = 0; i < a->n; i++) {
        u64 bi = i < b->n ? b->w[i] : 0;
        u128 d = (u128)a->w[i] - bi - br;
        t.w[i] = (u64)d; br = (u64)(d >> 64) & 1;
    }
    t.n = a->n; bn_norm(&t);
    *r = t;
}
static void bn_mul_u64(bn *r, const bn *a, u64 k) {
    bn t; memset(&t, 0, sizeof(t));
    u64 c = 0;
    for (int i = 0; i < a->n; i++) {
        u128 p = (u128)a->w[i] * k + c;
        t.w[i] = (u64)p; c = (u64)(p >> 64);
    }
    t.n = a->n;
    if (c) { t.w[t.n++] = c; }
    bn_norm(&t);
    *r = t;
}
static void bn_addmul_u64(bn *acc, const bn *a, u64 k) { bn t; bn_mul_u64(&t, a, k); bn_add(acc, acc, &t); }
static u64 bn_mod_u64(const bn *a, u64 m) {
    u128 r = 0;
    for (int i = a->n - 1; i >= 0; i--) r = ((r << 64) | a->w[i]) % m;
    return (u64)r;
}
static void bn_shr1(bn *r, const bn *a) {
    bn t; memset(&t, 0, sizeof(t));
    for (int i = 0; i < a->n; i++) t.w[i] = (a->w[i] >> 1) | (i + 1 < a->n ? a->w[i + 1] << 63 : 0);
    t.n = a->n; bn_norm(&t);
    *r = t;
}
/* Knuth algorithm D, 64-bit digits.  q = floor(u / v), r = u - q v.  v != 0. */
static void bn_divrem(bn *qo, bn *ro, const bn *u, const bn *v) {
    bn q, r; memset(&q, 0, sizeof(q)); memset(&r, 0, sizeof(r));
    if (bn_cmp(u, v) < 0) { r = *u; *qo = q; *ro = r; return; }
    int n = v->n, m = u->n - v->n;
    if (n == 1) {
        u128 rem = 0;
        for (int i = u->n - 1; i >= 0; i--) {
            u128 cur = (rem << 64) | u->w[i];
            q.w[i] = (u64)(cur / v->w[0]); rem = cur % v->w[0];
        }
        q.n = u->n; bn_norm(&q);
        bn_set_u64(&r, (u64)rem);
        *qo = q; *ro = r; return;
    }
    int s = __builtin_clzll(v->w[n - 1]);
    u64 vn[BW], un[BW + 1];
    for (int i = n - 1; i > 0; i--) vn[i] = s ? (v->w[i] << s) | (v->w[i - 1] >> (64 - s)) : v->w[i];
    vn[0] = v->w[0] << s;
    un[u->n] = s ? u->w[u->n - 1] >> (64 - s) : 0;
    for (int i = u->n - 1; i > 0; i--) un[i] = s ? (u->w[i] << s) | (u->w[i - 1] >> (64 - s)) : u->w[i];
    un[0] = u->w[0] << s;
    for (int j = m; j >= 0; j--) {
        u128 num = ((u128)un[j + n] << 64) | un[j + n - 1];
        u128 qhat = num / vn[n - 1], rhat = num % vn[n - 1];
        while ((qhat >> 64) || (u128)(u64)qhat * vn[n - 2] > ((rhat << 64) | un[j + n - 2])) {
            qhat--; rhat += vn[n - 1];
            if (rhat >> 64) break;
        }
        /* multiply-subtract */
        u64 borrow = 0, carry = 0;
        for (int i = 0; i < n; i++) {
            u128 p = (u128)(u64)qhat * vn[i] + carry;
            carry = (u64)(p >> 64);
            u128 d = (u128)un[i + j] - (u64)p - borrow;
            un[i + j] = (u64)d; borrow = (u64)(d >> 64) & 1;
        }
        u128 d = (u128)un[j + n] - carry - borrow;
        un[j + n] = (u64)d;
        if ((u64)(d >> 64) & 1) { /* add back */
            qhat--;
            u64 c = 0;
            for (int i = 0; i < n; i++) {
                u128 t = (u128)un[i + j] + vn[i] + c;
                un[i + j] = (u64)t; c = (u64)(t >> 64);
            }
            un[j + n] += c;
        }
        q.w[j] = (u64)qhat;
    }
    q.n = m + 1; bn_norm(&q);
    for (int i = 0; i < n; i++) r.w[i] = s ? (un[i] >> s) | (un[i + 1] << (64 - s)) : un[i];
    r.n = n; bn_norm(&r);
    *qo = q; *ro = r;
}

/* test hook: divrem on raw words (tests/test_oracle_c.py checks it against Python ints) */
void ref_test_divrem(const u64 *u, int un, const u64 *v, int vn, u64 *q, int *qn, u64 *r, int *rn) {
    bn a, b, qq, rr; memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
    memcpy(a.w, u, un * 8); a.n = un; bn_norm(&a);
    memcpy(b.w, v, vn * 8); b.n = vn; bn_norm(&b);
    bn_divrem(&qq, &rr, &a, &b);
    memcpy(q, qq.w, qq.n * 8); *qn = qq.n;
    memcpy(r, rr.w, rr.n * 8); *rn = rr.n;
}

/* ------------------------------------------------------------------------------------------ */
/* ring context: NegacyclicRing{CRTEncoded{L,...},N} (pow2_cyc_rings.jl:27-65, crt.jl:282-295)  */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int N, logN, L;
    u64 *q, *psi;
    /* per limb tables, each N entries: psi^i, psi^-i * N^-1, and w^k / w^-k for k < N/2 (+Shoup) */
    u64 **tw, **twp, **itw, **itwp, **om, **omp, **iom, **iomp;
    int *brev;
} refctx;

static void *xmalloc(size_t n) { void *p = malloc(n); if (!p) { fprintf(stderr, "ref_cpu: OOM\n"); abort(); } return p; }

refctx *ref_ctx_create(int N, int L, const u64 *q, const u64 *psi) {
    if (N < 2 || (N & (N - 1))) return NULL;
    refctx *c = (refctx *)xmalloc(sizeof(refctx));
    c->N = N; c->L = L; c->logN = __builtin_ctz(N);
    c->q = (u64 *)xmalloc(8 * L); c->psi = (u64 *)xmalloc(8 * L);
    u64 ***tabs[8] = {&c->tw, &c->twp, &c->itw, &c->itwp, &c->om, &c->omp, &c->iom, &c->iomp};
    for (int t = 0; t < 8; t++) *tabs[t] = (u64 **)xmalloc(sizeof(u64 *) * L);
    c->brev = (int *)xmalloc(sizeof(int) * N);
    for (int i = 0; i < N; i++) {
        int r = 0;
        for (int b = 0; b < c->logN; b++) r |= ((i >> b) & 1) << (c->logN - 1 - b);
        c->brev[i] = r;
    }
    for (int l = 0; l < L; l++) {
        u64 ql = q[l];
        c->q[l] = ql;
        u64 p = (psi && psi[l]) ? psi[l] : minimal_primitive_root(ql, 2 * (u64)N);
        if (powmod(p, 2 * (u64)N, ql) != 1) { /* pow2_cyc_rings.jl:31 */ free(c); return NULL; }
        c->psi[l] = p;
        for (int t = 0; t < 8; t++) (*tabs[t])[l] = (u64 *)xmalloc(8 * N);
        u64 pinv = invmod(p, ql), ninv = invmod((u64)N % ql, ql);
        u64 w = mulmod(p, p, ql), winv = mulmod(pinv, pinv, ql);
        u64 a = 1, b = ninv, x = 1, y = 1;
        for (int i = 0; i < N; i++) {
            c->tw[l][i] = a; c->twp[l][i] = shoup_pre(a, ql);
            c->itw[l][i] = b; c->itwp[l][i] = shoup_pre(b, ql);
            a = mulmod(a, p, ql); b = mulmod(b, pinv, ql);
            if (i < N / 2 || N == 1) {
                c->om[l][i] = x; c->omp[l][i] = shoup_pre(x, ql);
                c->iom[l][i] = y; c->iomp[l][i] = shoup_pre(y, ql);
                x = mulmod(x, w, ql); y = mulmod(y, winv, ql);
            }
        }
    }
    return c;
}

void ref_ctx_destroy(refctx *c) {
    if (!c) return;
    u64 **tabs[8] = {c->tw, c->twp, c->itw, c->itwp, c->om, c->omp, c->iom, c->iomp};
    for (int t = 0; t < 8; t++) { for (int l = 0; l < c->L; l++) free(tabs[t][l]); free(tabs[t]); }
    free(c->q); free(c->psi); free(c->brev); free(c);
}
void ref_ctx_psi(const refctx *c, u64 *out) { memcpy(out, c->psi, 8 * c->L); }

/* cyclic radix-2 DIT on bit-reversed input (the CTPlan of pow2_cyc_rings.jl:301,315) */
static void cyc_ntt(u64 *x, int N, const u64 *om, const u64 *omp, u64 q) {
    for (int m = 2; m <= N; m <<= 1) {
        int h = m >> 1, step = N / m;
        for (int s = 0; s < N; s += m)
            for (int j = 0; j < h; j++) {
                u64 u = x[s + j];
                u64 v = shoup_mul(x[s + j + h], om[j * step], omp[j * step], q);
                x[s + j] = addmod(u, v, q);
                x[s + j + h] = submod(u, v, q);
            }
    }
}

/* nntt(c), pow2_cyc_rings.jl:295-303: powmul by psi^i (:298) then the forward plan (:301). */
static void limb_nntt(const refctx *c, int l, u64 *a) {
    int N = c->N; u64 q = c->q[l];
    u64 *t = (u64 *)xmalloc(8 * N);
    for (int i = 0; i < N; i++) t[c->brev[i]] = shoup_mul(a[i], c->tw[l][i], c->twp[l][i], q);
    cyc_ntt(t, N, c->om[l], c->omp[l], q);
    memcpy(a, t, 8 * N); free(t);
}
/* inntt(c~), pow2_cyc_rings.jl:308-318: inverse plan (:315) then * N^-1 * psi^-i (:316-317). */
static void limb_inntt(const refctx *c, int l, u64 *a) {
    int N = c->N; u64 q = c->q[l];
    u64 *t = (u64 *)xmalloc(8 * N);
    for (int i = 0; i < N; i++) t[c->brev[i]] = a[i];
    cyc_ntt(t, N, c->iom[l], c->iomp[l], q);
    for (int i = 0; i < N; i++) a[i] = shoup_mul(t[i], c->itw[l][i], c->itwp[l][i], q);
    free(t);
}

/* data: [count][nl][N]; limb j of the data uses context modulus idx[j] (crtselect, crt.jl:185-211).
 * Per-limb dispatch = crt.jl:247-267. */
void ref_nntt(const refctx *c, const int *idx, int nl, u64 *data, long count) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < count * nl; i++) limb_nntt(c, idx[i % nl], data + i * c->N);
}
void ref_inntt(const refctx *c, const int *idx, int nl, u64 *data, long count) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < count * nl; i++) limb_inntt(c, idx[i % nl], data + i * c->N);
}

/* limb-wise + - * and unary -, crt.jl:120-134; op: 0 add, 1 sub, 2 mul, 3 neg(a) */
void ref_pointwise(const refctx *c, const int *idx, int nl, int op, const u64 *a, const u64 *b,
                   u64 *dst, long count) {
    int N = c->N;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < count * nl; i++) {
        u64 q = c->q[idx[i % nl]];
        const u64 *pa = a + i * N, *pb = b ? b + i * N : NULL;
        u64 *pd = dst + i * N;
        for (int k = 0; k < N; k++)
            pd[k] = op == 0 ? addmod(pa[k], pb[k], q) : op == 1 ? submod(pa[k], pb[k], q)
                  : op == 2 ? mulmod(pa[k], pb[k], q) : negmod(pa[k], q);
    }
}

/* scalar_mul, pow2_cyc_rings.jl:177-180; scalar given as residues per selected limb */
void ref_scalar_mul(const refctx *c, const int *idx, int nl, const u64 *scal, const u64 *a, u64 *dst,
                    long count) {
    int N = c->N;
    for (long i = 0; i < count * nl; i++) {
        u64 q = c->q[idx[i % nl]], s = scal[i % nl] % q;
        for (int k = 0; k < N; k++) dst[i * N + k] = mulmod(a[i * N + k], s, q);
    }
}

/* apply_galois_element, pow2_cyc_rings.jl:321-329 (coefficient domain) */
void ref_galois(const refctx *c, const int *idx, int nl, u64 g, const u64 *src, u64 *dst, long count) {
    int N = c->N;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < count * nl; i++) {
        u64 q = c->q[idx[i % nl]];
        for (u64 k = 0; k < (u64)N; k++) {
            u64 prod = g * k, qq = prod / N, r = prod % N;
            dst[i * N + r] = (qq & 1) ? negmod(src[i * N + k], q) : src[i * N + k];
        }
    }
}

/* modswitch(::RingElement), crt.jl:226-228 + :215-220; src [count][nl][N] -> dst [count][nl-1][N],
 * coefficient domain; c_last is used as its unsigned representative (utils.jl:39). */
void ref_modswitch(const refctx *c, const int *idx, int nl, const u64 *src, u64 *dst, long count) {
    int N = c->N;
    u64 ql = c->q[idx[nl - 1]];
#pragma omp parallel for schedule(static)
    for (long i = 0; i < count; i++)
        for (int j = 0; j < nl - 1; j++) {
            u64 qj = c->q[idx[j]], inv = invmod(ql % qj, qj);
            const u64 *cj = src + (i * nl + j) * N, *cl = src + (i * nl + nl - 1) * N;
            u64 *d = dst + (i * (nl - 1) + j) * N;
            for (int k = 0; k < N; k++) d[k] = mulmod(inv, submod(cj[k], cl[k] % qj, qj), qj);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* exact basis changes through a big integer (crt.jl:91-112)                                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int n;          /* number of moduli */
    u64 q[64];
    bn Q, halfQ;    /* product and Q >> 1 */
    bn Qi[64];      /* Q / q_i */
    u64 Qi_inv[64]; /* (Q/q_i)^-1 mod q_i */
} basis;

static void basis_init(basis *b, const refctx *c, const int *idx, int n) {
    if (n > 40) { fprintf(stderr, "ref_cpu: basis of %d limbs exceeds the bignum capacity\n", n); abort(); }
    b->n = n;
    bn_set_u64(&b->Q, 1);
    for (int i = 0; i < n; i++) { b->q[i] = c->q[idx[i]]; bn_mul_u64(&b->Q, &b->Q, b->q[i]); }
    bn_shr1(&b->halfQ, &b->Q);
    for (int i = 0; i < n; i++) {
        bn_set_u64(&b->Qi[i], 1);
        for (int j = 0; j < n; j++) if (j != i) bn_mul_u64(&b->Qi[i], &b->Qi[i], b->q[j]);
        b->Qi_inv[i] = invmod(bn_mod_u64(&b->Qi[i], b->q[i]), b->q[i]);
    }
}
/* convert(Integer, ::CRTEncoded), crt.jl:105-112: the unique x in [0, Q) with the given residues */
static void basis_to_int(const basis *b, const u64 *res, bn *x) {
    if (b->n == 1) { bn_set_u64(x, res[0]); return; } /* crt.jl:106-107 */
    bn acc; bn_set_u64(&acc, 0);
    for (int i = 0; i < b->n; i++) bn_addmul_u64(&acc, &b->Qi[i], mulmod(res[i], b->Qi_inv[i], b->q[i]));
    bn qq, r; bn_divrem(&qq, &r, &acc, &b->Q);
    *x = r;
}

/* ------------------------------------------------------------------------------------------ */
/* BFV multiplication (rlwe_she.jl:247-262 + bfv.jl:34-40,172-226)                              */
/* ------------------------------------------------------------------------------------------ */

/* switchel(T, e), bfv.jl:202-220: x in [0,q) -> integer (sign, magnitude) to be reduced into T */
static void switchel_int(const bn *x, const basis *from, const basis *to, bn *mag, int *neg) {
    bn diff;
    int from_lt_to = bn_cmp(&from->Q, &to->Q) < 0;
    if (from_lt_to) bn_sub(&diff, &to->Q, &from->Q); else bn_sub(&diff, &from->Q, &to->Q);
    *neg = 0;
    if (bn_cmp(x, &from->halfQ) > 0) {
        if (from_lt_to) bn_add(mag, x, &diff);                       /* T(en + diff) */
        else if (bn_cmp(x, &diff) >= 0) bn_sub(mag, x, &diff);       /* T(en - diff) >= 0 */
        else { bn_sub(mag, &diff, x); *neg = 1; }                    /* T(en - diff) < 0 */
    } else *mag = *x;
}
static inline u64 reduce_signed(const bn *mag, int neg, u64 q) {
    u64 r = bn_mod_u64(mag, q);
    return neg ? negmod(r, q) : r;
}

/* switch(R, e), bfv.jl:222-226: src [ns][N] over `from` -> dst [nd][N] over `to`, coefficient domain */
static void switch_poly(const basis *from, const basis *to, int N, const u64 *src, u64 *dst) {
    u64 res[64];
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < from->n; i++) res[i] = src[(long)i * N + k];
        bn x, mag; int neg;
        basis_to_int(from, res, &x);
        switchel_int(&x, from, to, &mag, &neg);
        for (int j = 0; j < to->n; j++) dst[(long)j * N + k] = reduce_signed(&mag, neg, to->q[j]);
    }
}

/* multround(e, t, q) then switch back, bfv.jl:35-40,172-190: in place on src [nb][N] over `big`,
 * output dst [ns][N] over `small`.  Literal restatement:
 *   SignedMod(x) * t          -> (t x) mod Qbig                (signedmod.jl:24-28)
 *   div(., q, TiesAway)       -> round(centred / q)            (signedmod.jl:30-32, div_hacks.jl:120-135)
 *   oftype(e, .)              -> mod Qbig
 *   switch(R, .)              -> switchel into the small basis (bfv.jl:202-226)                 */
static void contract_poly(const basis *big, const basis *small, u64 t, int N, const u64 *src, u64 *dst) {
    u64 res[64];
    for (int k = 0; k < N; k++) {
        for (int i = 0; i < big->n; i++) res[i] = mulmod(src[(long)i * N + k], t % big->q[i], big->q[i]);
        bn z; basis_to_int(big, res, &z);
        int neg = bn_cmp(&z, &big->halfQ) > 0;
        bn mag; if (neg) bn_sub(&mag, &big->Q, &z); else mag = z;
        bn w, r, r2; bn_divrem(&w, &r, &mag, &small->Q);
        bn_add(&r2, &r, &r);
        if (bn_cmp(&r2, &small->Q) >= 0) { bn one; bn_set_u64(&one, 1); bn_add(&w, &w, &one); } /* ties away */
        /* back into ℛbig as an element of [0, Qbig) */
        bn en;
        if (neg && w.n) { bn wm, qq; bn_divrem(&qq, &wm, &w, &big->Q); if (wm.n) bn_sub(&en, &big->Q, &wm); else en = wm; }
        else { bn qq; bn_divrem(&qq, &en, &w, &big->Q); }
        bn mag2; int neg2;
        switchel_int(&en, big, small, &mag2, &neg2);
        for (int j = 0; j < small->n; j++) dst[(long)j * N + k] = reduce_signed(&mag2, neg2, small->q[j]);
    }
}

/* enc_mul for BFVParams. c1, c2: [batch][2][ns][N] over (cs, idx_s); out: [batch][3][ns][N];
 * all coefficient domain.  big ring = (cb, idx_b) with nb limbs. */
void ref_bfv_mul(const refctx *cs, const int *idx_s, int ns, const refctx *cb, const int *idx_b, int nb,
                 u64 t, const u64 *c1, const u64 *c2, u64 *out, long batch) {
    int N = cs->N;
    basis *small = (basis *)xmalloc(sizeof(basis)), *big = (basis *)xmalloc(sizeof(basis));
    basis_init(small, cs, idx_s, ns); basis_init(big, cb, idx_b, nb);
    long psz = (long)nb * N;
#pragma omp parallel for schedule(dynamic)
    for (long b = 0; b < batch; b++) {
        u64 *e = (u64 *)xmalloc(8 * psz * 4), *prod = (u64 *)xmalloc(8 * psz * 3);
        /* mul_expand, bfv.jl:34 */
        for (int p = 0; p < 2; p++) {
            switch_poly(small, big, N, c1 + ((b * 2 + p) * ns) * (long)N, e + p * psz);
            switch_poly(small, big, N, c2 + ((b * 2 + p) * ns) * (long)N, e + (2 + p) * psz);
        }
        for (int p = 0; p < 4; p++)
            for (int j = 0; j < nb; j++) limb_nntt(cb, idx_b[j], e + p * psz + (long)j * N);
        /* c[i+j-1] += c1[i]*c2[j], rlwe_she.jl:255-258 */
        for (int j = 0; j < nb; j++) {
            u64 q = cb->q[idx_b[j]];
            const u64 *a0 = e + 0 * psz + (long)j * N, *a1 = e + 1 * psz + (long)j * N;
            const u64 *b0 = e + 2 * psz + (long)j * N, *b1 = e + 3 * psz + (long)j * N;
            u64 *p0 = prod + 0 * psz + (long)j * N, *p1 = prod + 1 * psz + (long)j * N, *p2 = prod + 2 * psz + (long)j * N;
            for (int k = 0; k < N; k++) {
                p0[k] = mulmod(a0[k], b0[k], q);
                p1[k] = addmod(mulmod(a0[k], b1[k], q), mulmod(a1[k], b0[k], q), q);
                p2[k] = mulmod(a1[k], b1[k], q);
            }
        }
        for (int p = 0; p < 3; p++)
            for (int j = 0; j < nb; j++) limb_inntt(cb, idx_b[j], prod + p * psz + (long)j * N);
        /* mul_contract, bfv.jl:35-40 */
        for (int p = 0; p < 3; p++)
            contract_poly(big, small, t, N, prod + p * psz, out + ((b * 3 + p) * ns) * (long)N);
        free(e); free(prod);
    }
    free(small); free(big);
}

/* enc_mul without expand/contract (BGV / CKKS, rlwe_she.jl:39-40,247-262):
 * c1 [batch][n1][nl][N], c2 [batch][n2][nl][N] -> out [batch][n1+n2-1][nl][N], coefficient domain */
void ref_enc_mul(const refctx *c, const int *idx, int nl, const u64 *c1, int n1, const u64 *c2, int n2,
                 u64 *out, long batch) {
    int N = c->N, no = n1 + n2 - 1;
    long psz = (long)nl * N;
#pragma omp parallel for schedule(dynamic)
    for (long b = 0; b < batch; b++) {
        u64 *a = (u64 *)xmalloc(8 * psz * n1), *bb = (u64 *)xmalloc(8 * psz * n2);
        memcpy(a, c1 + b * n1 * psz, 8 * psz * n1); memcpy(bb, c2 + b * n2 * psz, 8 * psz * n2);
        for (long i = 0; i < (long)n1 * nl; i++) limb_nntt(c, idx[i % nl], a + i * N);
        for (long i = 0; i < (long)n2 * nl; i++) limb_nntt(c, idx[i % nl], bb + i * N);
        u64 *o = out + b * no * psz;
        memset(o, 0, 8 * psz * no);
        for (int i = 0; i < n1; i++) for (int j = 0; j < n2; j++) for (int l = 0; l < nl; l++) {
            u64 q = c->q[idx[l]];
            const u64 *x = a + i * psz + (long)l * N, *y = bb + j * psz + (long)l * N;
            u64 *z = o + (i + j) * psz + (long)l * N;
            for (int k = 0; k < N; k++) z[k] = addmod(z[k], mulmod(x[k], y[k], q), q);
        }
        for (long i = 0; i < (long)no * nl; i++) limb_inntt(c, idx[i % nl], o + i * N);
        free(a); free(bb);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* key switching (rlwe_she.jl:315-347, modulusraising.jl:35-49)                                 */
/* ------------------------------------------------------------------------------------------ */

/* kc: key ring context with Lk moduli (last = special prime when special != 0).
 * ct: [batch][np][l][N] (np = 2 or 3) over key limbs 0..l-1, coefficient domain.
 * evk: [ndig][2][Lk][N], component 0 = mask, 1 = masked (rlwe_she.jl:297), NTT domain
 *      (natural order), full key basis; digits 0..l-1 are consumed (rlwe_she.jl:340).
 * out: [batch][2][l][N] coefficient domain.
 * RNS-digit decomposition only (relin_window == 0, rlwe_she.jl:326-329). */
void ref_keyswitch(const refctx *kc, int l, int special, const u64 *evk, const u64 *ct, int np, u64 *out,
                   long batch) {
    int N = kc->N, Lk = kc->L;
    int nw = special ? l + 1 : l; /* limbs of the working ring */
    int widx[64];
    for (int j = 0; j < l; j++) widx[j] = j;
    if (special) widx[l] = Lk - 1; /* downswitch_keyelement, modulusraising.jl:43-49 */
    long wsz = (long)nw * N;
    u64 P = kc->q[Lk - 1];
#pragma omp parallel for schedule(dynamic)
    for (long b = 0; b < batch; b++) {
        const u64 *c = ct + b * np * (long)l * N;
        u64 *acc = (u64 *)xmalloc(8 * wsz * 2), *dig = (u64 *)xmalloc(8 * wsz);
        /* c1 = keyswitch_expand(c[1]); c2 = zero or keyswitch_expand(c[2]); rlwe_she.jl:323-324.
         * Accumulate in the NTT domain: acc[0] = c1, acc[1] = c2. */
        for (int s = 0; s < 2; s++) {
            u64 *a = acc + s * wsz;
            if (s == 1 && np == 2) { memset(a, 0, 8 * wsz); continue; }
            const u64 *src = c + (long)s * l * N;
            for (int j = 0; j < l; j++) {
                u64 q = kc->q[j], m = special ? P % q : 1; /* CRTExpand: * P, crt.jl:38-40 */
                for (int k = 0; k < N; k++) a[(long)j * N + k] = mulmod(src[(long)j * N + k], m, q);
            }
            if (special) memset(a + (long)l * N, 0, 8 * N); /* appended zero limb */
            for (int j = 0; j < nw; j++) limb_nntt(kc, widx[j], a + (long)j * N);
        }
        const u64 *cend = c + (long)(np - 1) * l * N;
        for (int i = 0; i < l; i++) {
            /* digit i: SignedMod of limb i lifted into every working limb, rlwe_she.jl:329 */
            u64 qi = kc->q[i], half = qi / 2;
            for (int j = 0; j < nw; j++) {
                u64 q = kc->q[widx[j]];
                for (int k = 0; k < N; k++) {
                    u64 x = cend[(long)i * N + k];
                    dig[(long)j * N + k] = x > half ? negmod((qi - x) % q, q) : x % q;
                }
                limb_nntt(kc, widx[j], dig + (long)j * N);
            }
            const u64 *mask = evk + ((long)i * 2 + 0) * Lk * N, *masked = evk + ((long)i * 2 + 1) * Lk * N;
            for (int j = 0; j < nw; j++) {
                u64 q = kc->q[widx[j]];
                const u64 *mk = mask + (long)widx[j] * N, *md = masked + (long)widx[j] * N, *d = dig + (long)j * N;
                u64 *a1 = acc + (long)j * N, *a2 = acc + wsz + (long)j * N;
                for (int k = 0; k < N; k++) {
                    a2[k] = addmod(a2[k], mulmod(mk[k], d[k], q), q); /* c2 += key.mask*ps[i],   :342 */
                    a1[k] = addmod(a1[k], mulmod(md[k], d[k], q), q); /* c1 += key.masked*ps[i], :343 */
                }
            }
        }
        for (int s = 0; s < 2; s++) {
            u64 *a = acc + s * wsz;
            for (int j = 0; j < nw; j++) limb_inntt(kc, widx[j], a + (long)j * N);
            u64 *o = out + (b * 2 + s) * (long)l * N;
            if (special) ref_modswitch(kc, widx, nw, a, o, 1); /* keyswitch_contract = modswitch, modulusraising.jl:42 */
            else memcpy(o, a, 8 * wsz);
        }
        free(acc); free(dig);
    }
}

/* exact centred lift of src [ns][N] over (c, idx_s) into (c2, idx_d) -- exposes switch() for tests */
void ref_switch(const refctx *c, const int *idx_s, int ns, const refctx *c2, const int *idx_d, int nd,
                const u64 *src, u64 *dst, long count) {
    basis *from = (basis *)xmalloc(sizeof(basis)), *to = (basis *)xmalloc(sizeof(basis));
    basis_init(from, c, idx_s, ns); basis_init(to, c2, idx_d, nd);
    int N = c->N;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < count; i++) switch_poly(from, to, N, src + i * ns * (long)N, dst + i * nd * (long)N);
    free(from); free(to);
}

/* multround + switch back for tests: src [count][nb][N] over big -> dst [count][ns][N] over small */
void ref_contract(const refctx *cb, const int *idx_b, int nb, const refctx *cs, const int *idx_s, int ns, u64 t,
                  const u64 *src, u64 *dst, long count) {
    basis *big = (basis *)xmalloc(sizeof(basis)), *small = (basis *)xmalloc(sizeof(basis));
    basis_init(big, cb, idx_b, nb); basis_init(small, cs, idx_s, ns);
    int N = cb->N;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < count; i++) contract_poly(big, small, t, N, src + i * nb * (long)N, dst + i * ns * (long)N);
    free(big); free(small);
}

int ref_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ref_set_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
