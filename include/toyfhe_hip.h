/*
 * toyfhe_hip.h -- C ABI of libtoyfhe_hip.so, the MI355X (gfx950) engine for the power-of-two
 * cyclotomic RNS polynomial path of JuliaCrypto/ToyFHE.jl.
 *
 * The reference has no FFI: its extension seam is Julia dispatch on the storage type of
 * RingElement{ℛ,Field,Storage} (src/pow2_cyc_rings.jl:93-96), exactly how src/crt.jl:247-275 plugs
 * the RNS NTT in.  Each entry point below names the reference method(s) a Julia shim binds it to
 * (see INTEGRATION.md for the ccall side).  All citations are relative to /root/reference/src/.
 *
 * Conventions
 *   - every function returns 0 (TFHE_OK) or a negative tfhe_status; tfhe_last_error() gives text.
 *     Nothing aborts or throws across the ABI.
 *   - residues are uint64_t in [0, q_l); q_l < 2^62, odd primes with 2N | q_l - 1.
 *   - polynomial data is device memory, layout [count][limbs][N] (limb-major SoA = the StructArray
 *     field arrays of src/crt.jl:150-156); a ciphertext batch is [batch][polys][limbs][N].
 *   - limb j of a buffer uses context modulus limb_idx[j] (crtselect, src/crt.jl:185-211);
 *     limb_idx == NULL means 0..limbs-1.  limb_idx is a HOST array.
 *   - domain (coefficient "primal" / NTT "dual", src/pow2_cyc_rings.jl:93-145) is tracked by the
 *     caller; NTT-domain data is in natural order: â[k] = a(ψ^(2k+1)) (src/pow2_cyc_rings.jl:278-294).
 *   - all work is enqueued on the context's stream (tfhe_ctx_set_stream); calls are asynchronous
 *     unless stated.  No host pointer is retained after a call returns.
 *   - threading: a tfhe_ctx (and a tfhe_bfv_plan) carries per-call scratch (transform workspaces, the prepared key rows
 *     of the fused key switch) that is recycled in stream order, so ONE host thread drives a given context at a time;
 *     different contexts / plans are independent and may be driven from different threads (no global mutable state
 *     besides the thread-local error text).  Share read-only inputs (evaluation keys, ciphertexts) freely.
 *     tfhe_free, tfhe_ctx_destroy, tfhe_bfv_plan_destroy and tfhe_comm_destroy may be called from ANY thread (a garbage
 *     collector's finalizer thread): tfhe_free takes the allocator's lock and parks the block behind events recorded on every
 *     live context's stream -- it never waits and never touches a context's scratch; the destroy calls synchronise the
 *     object's own stream first and must only not race with a call that is still USING that object.
 *   - a BFV plan over two contexts runs the extension-basis work on ℛbig's stream and the key switch on ℛ's stream and
 *     orders the two with events; it never re-points a context's stream.  Results are ordered on ℛ's stream
 *     (tfhe_ctx_sync(small) waits for them).
 */
#ifndef TOYFHE_HIP_H
#define TOYFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TFHE_OK = 0,
    TFHE_E_BADARG = -1,          /* @assert-class failures (pow2_cyc_rings.jl:31,61,116; rlwe_she.jl:318) */
    TFHE_E_DOMAIN = -2,
    TFHE_E_LEVEL_MISMATCH = -3,
    TFHE_E_PARAMS_MISMATCH = -4, /* UsageError (rlwe_she.jl:223-225,233-235,248-250) */
    TFHE_E_NOMEM = -5,
    TFHE_E_HIP = -6,
    TFHE_E_UNSUPPORTED = -7      /* error("... only implemented for ...") (crt.jl:270,274) */
} tfhe_status;

typedef struct tfhe_ctx tfhe_ctx;       /* a ring: NegacyclicRing{CRTEncoded{L,...},N} */
typedef struct tfhe_bfv_plan tfhe_bfv_plan; /* (ℛ, ℛbig, t) of a BFVParams */
typedef struct tfhe_comm tfhe_comm;     /* the ranks of a multi-GPU job (one process per GPU) */

const char *tfhe_last_error(void);      /* thread-local text of the last failure */
int tfhe_device_count(int *n);
int tfhe_set_device(int dev);

/* ---- ring context ---------------------------------------------------------------------------
 * NegacyclicRing{BaseRing,N}(ψ) (pow2_cyc_rings.jl:27-65) / NegacyclicRing(N, logqs) (crt.jl:282-295).
 * psi[l] == 0 (or psi == NULL) derives GaloisFields.minimal_primitive_root(𝔽q, 2N) (pow2_cyc_rings.jl:40,
 * crt.jl:142-144); a non-zero psi[l] must satisfy psi^(2N) == 1 (pow2_cyc_rings.jl:31) else TFHE_E_BADARG. */
int tfhe_ctx_create(int64_t N, int L, const uint64_t *q, const uint64_t *psi, tfhe_ctx **out);
int tfhe_ctx_destroy(tfhe_ctx *ctx);
int tfhe_ctx_psi(const tfhe_ctx *ctx, uint64_t *psi_out /* [L] */);
int tfhe_ctx_set_stream(tfhe_ctx *ctx, void *hip_stream /* hipStream_t, NULL = library-owned */);
int tfhe_ctx_sync(tfhe_ctx *ctx);
/* Device-side ordering between two contexts (each has its own stream): work submitted to `ctx` after this call starts after
 * everything submitted to `producer` before it.  No host wait.  (The reference is single-threaded and synchronous, so it has
 * no counterpart; the host mirrors use it where a ciphertext of one ring context meets a key of another, rlwe_she.jl:315.) */
int tfhe_ctx_wait_for(tfhe_ctx *ctx, tfhe_ctx *producer);
/* choose the NTT kernel family: 0 = auto (register-blocked LDS kernel; exact-integer fp64 butterflies
 * when every selected modulus is < 2^50 + 2^40, u64 Shoup butterflies otherwise), 1 = force the generic
 * radix-2 kernel, 2 = force the u64 register-blocked kernel, 3 = fp64 kernels one operation per launch: no fused
 * BFV core / key-switch kernels, no next-row overlap (LDS-DMA staging in the inverse, register prefetch in the
 * forward), no read-once digit lift -- 1, 2 and 3 are cross-check paths for tests */
int tfhe_ctx_set_ntt_variant(tfhe_ctx *ctx, int variant);

/* ---- device memory helpers (for callers without a GPU array package) -------------------------
 * tfhe_malloc / tfhe_free are a size-bucketed recycling allocator (csrc/dev_alloc.h): the reference allocates a fresh array
 * per ring operation (e.g. broadcast results, pow2_cyc_rings.jl:167,200-214), so the mirror frees and allocates thousands of
 * equally sized buffers per circuit.  tfhe_free never synchronises: the block is parked behind events recorded on every
 * live context's stream and is handed out again only once they have completed.  TFHE_ALLOC_CACHE=0 selects plain
 * hipMalloc / hipFree.  tfhe_alloc_trim returns the cached blocks to the driver (drains the device). */
int tfhe_malloc(size_t bytes, void **dptr);
int tfhe_free(void *dptr);
int tfhe_alloc_stats(uint64_t *live_bytes, uint64_t *cached_bytes, uint64_t *hip_mallocs, uint64_t *reuses);
int tfhe_alloc_trim(void);
int tfhe_memcpy_h2d(void *dst, const void *src, size_t bytes);  /* synchronous */
int tfhe_memcpy_d2h(void *dst, const void *src, size_t bytes);  /* synchronous */
int tfhe_memcpy_d2d(tfhe_ctx *ctx, void *dst, const void *src, size_t bytes); /* on the ctx stream */
int tfhe_memset(tfhe_ctx *ctx, void *dst, int byte, size_t bytes);
/* ciphertext staging: the reference keeps a ciphertext as a tuple of ring elements (CipherText.cs, rlwe_she.jl:131-136),
 * the fused entry points take [batch][polys][limbs][N].  Component p of a packed batch [count][polys][words]
 * <- / -> a batch of single polynomials [count][words] (words = limbs * N), one strided copy on the ctx stream. */
int tfhe_pack_poly(tfhe_ctx *ctx, uint64_t *packed, const uint64_t *src, int polys, int p, size_t words, int64_t count);
int tfhe_unpack_poly(tfhe_ctx *ctx, uint64_t *dst, const uint64_t *packed, int polys, int p, size_t words, int64_t count);
/* dst [count][words] <- src [words] repeated: one ring element (a plaintext operand, a key) against a batch of ciphertexts
 * (the `Ref(c)` / scalar-broadcast patterns of rlwe_she.jl:143 and ckksencoding.jl:99-124). */
int tfhe_broadcast_poly(tfhe_ctx *ctx, uint64_t *dst, const uint64_t *src, size_t words, int64_t count);

/* ---- K1/K2: nntt / inntt --------------------------------------------------------------------
 * NTT.nntt / NTT.inntt on RingCoeffs (pow2_cyc_rings.jl:295-318), per limb as crt.jl:247-267.
 * count = number of polynomials ([count][limbs][N]); src == dst allowed. */
int tfhe_nntt(tfhe_ctx *ctx, const uint64_t *src, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);
int tfhe_inntt(tfhe_ctx *ctx, const uint64_t *src, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);

/* ---- K3/K4: limb-wise arithmetic (crt.jl:120-134; pow2_cyc_rings.jl:167,177-219) -------------- */
int tfhe_add(tfhe_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);
int tfhe_sub(tfhe_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);
int tfhe_neg(tfhe_ctx *ctx, const uint64_t *a, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);
int tfhe_mul(tfhe_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);
/* dst = acc + a*b (the `c += x*y` pattern of rlwe_she.jl:257,342-343) */
int tfhe_mad(tfhe_ctx *ctx, const uint64_t *acc, const uint64_t *a, const uint64_t *b, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);
/* dst = (acc +) sum_{k < n_terms} a[k] .* b[k], every operand [count][limbs][N] (acc may be NULL; dst may alias acc): the
 * accumulation loop `result += rotated_k * diagonal_k` of the diagonal matrix-vector product (examples/encrypted_mnist/
 * infer.jl:140-149 -- there one ring multiplication and one ring addition per term, pow2_cyc_rings.jl:167,200-214) in one pass.
 * a, b: host arrays of n_terms device pointers.  Exact: the same canonical residues as tfhe_mad term by term. */
int tfhe_dot(tfhe_ctx *ctx, const uint64_t *acc, const uint64_t *const *a, const uint64_t *const *b, int n_terms, uint64_t *dst,
             int64_t count, int limbs, const int32_t *limb_idx);
/* dst = sum_{k < n_terms} scalars[k] * a[k], every a[k] and dst [count][limbs][N], either domain: the scalar-weighted sum of ring
 * elements (per term one scalar_mul, pow2_cyc_rings.jl:177-185, and one +, :200-214) -- e.g. a convolution with plaintext
 * scalar weights over encrypted inputs, examples/encrypted_mnist/infer.jl:127-129 (49 terms per channel and component) -- in
 * one pass.  scalars: HOST array [n_terms][limbs] of residues (scalars[k][j] < modulus of limb j); a: host array of n_terms
 * device pointers; n_terms <= 64.  Exact: the canonical residues of the term-by-term sum. */
int tfhe_lincomb(tfhe_ctx *ctx, const uint64_t *scalars, const uint64_t *const *a, int n_terms, uint64_t *dst, int64_t count, int limbs,
                 const int32_t *limb_idx);
/* n_out such sums of the SAME operands in one pass over them -- the output channels of a convolution layer, infer.jl:127-131 (each of
 * the 4 channels weighs the same 49 encrypted inputs): dst[o] = sum_k scalars[o][k] * a[k].  scalars: HOST array
 * [n_out][n_terms][limbs] of residues; dst: host array of n_out device pointers ([count][limbs][N] each, distinct from the
 * operands); n_terms <= 64.  Same words as n_out calls of tfhe_lincomb. */
int tfhe_lincomb_many(tfhe_ctx *ctx, const uint64_t *scalars, const uint64_t *const *a, int n_terms, uint64_t *const *dst, int n_out,
                      int64_t count, int limbs, const int32_t *limb_idx);
/* scalar_mul (pow2_cyc_rings.jl:177-185): scalar given as residues scal[j] mod q_{limb_idx[j]} (host array) */
int tfhe_scalar_mul(tfhe_ctx *ctx, const uint64_t *scal, const uint64_t *a, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);

/* ---- K5: tensor of two 2-element ciphertexts in the NTT domain (rlwe_she.jl:255-258) ----------
 * a, b: [batch][2][limbs][N]; out: [batch][3][limbs][N] = (a0 b0, a0 b1 + a1 b0, a1 b1). */
int tfhe_tensor(tfhe_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, int64_t batch, int limbs, const int32_t *limb_idx);

/* ---- K6/K7: level operations ------------------------------------------------------------------
 * modswitch(::RingElement) (crt.jl:226-228 with :215-220): coefficient domain,
 * src [count][limbs][N] -> dst [count][limbs-1][N]; c_last taken as its unsigned representative. */
int tfhe_rescale(tfhe_ctx *ctx, const uint64_t *src, uint64_t *dst, int64_t count, int limbs, const int32_t *limb_idx);
/* crtselect / drop_last / modswitch_drop (crt.jl:185-213,222-236): gather limbs `which` (positions in
 * the source buffer) of each polynomial: src [count][src_limbs][N] -> dst [count][n_which][N]. */
int tfhe_select_limbs(tfhe_ctx *ctx, const uint64_t *src, uint64_t *dst, int64_t count, int src_limbs, const int32_t *which, int n_which);

/* ---- K8: apply_galois_element (pow2_cyc_rings.jl:321-329), coefficient domain; src != dst. */
int tfhe_galois(tfhe_ctx *ctx, const uint64_t *src, uint64_t *dst, uint64_t galois_element, int64_t count, int limbs, const int32_t *limb_idx);

/* ---- K9-K11 (+K6): keyswitch (rlwe_she.jl:315-347) with RNS digits (relin_window == 0, :326-329) --
 * The KEY ring is the first key_limbs (= Lk) moduli of ctx (ctx may hold further moduli, e.g. a BFV
 * extension basis).  The ciphertext lives on key limbs 0..level-1.
 *   special != 0: ModulusRaised (modulusraising.jl:35-49): key modulus Lk-1 is the special prime;
 *                 expand = *P and append a zero limb, contract = modswitch by P; level <= Lk-1.
 *   evk: [n_digits][2][Lk][N], component 0 = mask, 1 = masked (rlwe_she.jl:297), NTT domain, full key
 *        basis; digits 0..level-1 are consumed (rlwe_she.jl:340), n_digits >= level.
 *   ct:  [batch][polys][level][N], polys in {2,3} (rlwe_she.jl:318), coefficient domain.
 *   out: [batch][2][level][N], coefficient domain. */
int tfhe_keyswitch(tfhe_ctx *ctx, int key_limbs, int level, int special, const uint64_t *evk, int n_digits, const uint64_t *ct, int polys, uint64_t *out, int64_t batch);
/* rotate(gk, c) = keyswitch(gk, apply_galois_element(c, g)) (rlwe_she.jl:355-359) */
int tfhe_rotate(tfhe_ctx *ctx, int key_limbs, int level, int special, const uint64_t *evk, int n_digits, uint64_t galois_element, const uint64_t *ct, uint64_t *out, int64_t batch);
/* the same with the key already prepared by tfhe_galois_key_prepare for this galois_element (below): a caller that rotates by one
 * Galois element again and again (infer.jl:140-149: 63 chained rotate(gk, rotated) per matrix product, ONE key) prepares the key
 * once instead of once per call.  Bit-identical to tfhe_rotate on the plain key. */
int tfhe_rotate_prepared(tfhe_ctx *ctx, int key_limbs, int level, int special, const uint64_t *evk_prepared, int n_digits, uint64_t galois_element,
                         const uint64_t *ct, uint64_t *out, int64_t batch);

/* Hoisted rotations: out[r] = rotate(gk_r, c) for r < n_rot from one digit decomposition of c (the RNS digits commute with
 * the automorphism; in the NTT domain it is an index permutation), bit-identical to n_rot calls of tfhe_rotate at
 * level * (level [+1]) forward transforms in total instead of per rotation -- the shape of a diagonal-method matrix product
 * (ckks_matmul.jl:33-41, infer.jl:140-149) when every rotation starts from the same ciphertext.
 *   evks: HOST array of n_rot device pointers (one Galois key each, layout as tfhe_keyswitch; prepared != 0: the outputs of
 *   tfhe_galois_key_prepare); galois: HOST array [n_rot];
 *   ct: [batch][2][level][N]; out: [n_rot][batch][2][level][N]. */
int tfhe_rotate_many(tfhe_ctx *ctx, int key_limbs, int level, int special, const uint64_t *const *evks, int n_digits, int prepared,
                     const uint64_t *galois_elements, int n_rot, const uint64_t *ct, uint64_t *out, int64_t batch);
/* Diagonal matrix-vector product in one call (examples/encrypted_mnist/infer.jl:140-149, test/ckks_matmul.jl:33-41):
 *     out = diag[0] .* c + sum_{r < n_rot} diag[r+1] .* rotate(gk_r, c)
 * with the rotations hoisted as in tfhe_rotate_many and every step run over all rotations at once; bit-identical to
 * tfhe_rotate_many -> tfhe_nntt -> tfhe_dot.  evks: HOST array of n_rot device pointers to PREPARED Galois keys
 * (tfhe_galois_key_prepare); galois_elements: HOST array [n_rot]; n_rot <= 64;
 *   diags: device [n_rot + 1][level][N], the plaintext diagonals in the NTT domain at the ciphertext's level, shared by the batch;
 *   ct: [batch][2][level][N] coefficient domain; out: [batch][2][level][N] NTT domain (the product's scale is the caller's business). */
int tfhe_matmul_diag(tfhe_ctx *ctx, int key_limbs, int level, int special, const uint64_t *const *evks, int n_digits,
                     const uint64_t *galois_elements, int n_rot, const uint64_t *diags, const uint64_t *ct, uint64_t *out, int64_t batch);
/* the one-time key preparation of the hoisted rotations: evk_out = the Galois key of x -> x^g (layout of tfhe_keyswitch,
 * n_digits components over key_limbs moduli) with every NTT-domain row permuted by g^-1, so that the key products run on the
 * transformed digits of the unrotated ciphertext.  prepared = 0 above takes plain keys and prepares them per call. */
int tfhe_galois_key_prepare(tfhe_ctx *ctx, int key_limbs, int n_digits, uint64_t galois_element, const uint64_t *evk, uint64_t *evk_out);

/* ---- K14: keyswitch with base-2^w digits (relin_window = w != 0, rlwe_she.jl:330-338; the default for
 * single-modulus rings, rlwe_she.jl:271) -------------------------------------------------------------
 * The ciphertext ring is limbs 0..level-1 of ctx, the key ring limbs 0..key_limbs-1 (as tfhe_keyswitch).  Digit i of a
 * coefficient is digit i of convert(Integer, x) in [0, Q_level) -- reconstructed exactly from the residues when
 * level > 1 -- embedded in every working limb; out_s = c_s + sum_i key_i,s * digit_i.
 *   special != 0: ModulusRaised (modulusraising.jl:35-49) -- limb key_limbs-1 is the special prime P, the working limbs
 *        are [0..level-1, key_limbs-1], c is raised to P c and the sums are contracted by floor(./P) (crt.jl:215-220).
 *   evk: [n_windows][2][key_limbs][N], component 0 = mask, 1 = masked, NTT domain; n_windows >= ndigits(Q_level, base = 2^w)
 *        = ceil(bitlength(Q_level) / w) (rlwe_she.jl:282,333; a key of a larger ring has more components, the first
 *        ndigits(Q_level) are used, rlwe_she.jl:340), else TFHE_E_PARAMS_MISMATCH.
 *   window_bits: 1..32 with 2^w below every modulus.   ct / out as tfhe_keyswitch. */
int tfhe_keyswitch_window(tfhe_ctx *ctx, int key_limbs, int level, int special, int window_bits, const uint64_t *evk, int n_windows,
                          const uint64_t *ct, int polys, uint64_t *out, int64_t batch);

/* ---- CKKS encode / decode (float; ckksencoding.jl:56-97, FixedRational ckks.jl:35-59) -- SURVEY §8(f) ----
 * The ring is limbs 0..level-1 of ctx.  scale = scale_mant * 2^scale_exp2 (2^40 = (1, 40); any positive scale to
 * 2^-63 relative).  slots: [batch][N/2] complex doubles (re, im interleaved), device memory.
 *   encode: slots -> ifft over the (Z/2N)^* orbit + psi-twist -> n_k = round(x_k * scale) (exact product, ties to even,
 *           ckks.jl:42) -> residues [batch][level][N], coefficient domain.
 *   decode: residues -> centred integer (exact CRT) -> Float64(n / scale) -> conj twist -> fft -> slots.
 * Float path: transform summation order differs from FFTW, so results agree with the reference to rounding
 * (encode: integers equal up to +-1 at rounding boundaries; decode: |err| <= 8 log2(N) eps max|slot|). */
int tfhe_ckks_encode(tfhe_ctx *ctx, int level, uint64_t scale_mant, int scale_exp2, const double *slots, uint64_t *out, int64_t batch);
int tfhe_ckks_decode(tfhe_ctx *ctx, int level, uint64_t scale_mant, int scale_exp2, const uint64_t *in, double *slots, int64_t batch);

/* ---- device-side samplers for RingSampler (poly.jl:7-23; RNS crt.jl:277-279) -- SURVEY §8(f) -------------------
 * The ring is limbs 0..level-1 of ctx; out: [count][level][N], coefficient domain.  The random stream is Philox4x32-10
 * keyed by `seed`, counter = (coefficient index, polynomial index first_poly + p < 2^32, attempt/limb, stream) -- defined in
 * csrc/sample_kernels.h and restated in oracle/spec.py (Julia's generator cannot be matched).
 *   uniform : independent exactly-uniform residues per limb (rejection sampling).
 *   gaussian: multiplier * round(N(0, sigma^2)) (Box-Muller, ties to even), the same integer in every limb
 *             (multiplier = 1 for BFV/CKKS noise and secrets, = t for BGV, bgv.jl:27-34). */
int tfhe_sample_uniform(tfhe_ctx *ctx, int level, uint64_t seed, uint32_t stream, uint64_t first_poly, uint64_t *out, int64_t count);
int tfhe_sample_gaussian(tfhe_ctx *ctx, int level, double sigma, uint64_t multiplier, uint64_t seed, uint32_t stream, uint64_t first_poly, uint64_t *out, int64_t count);

/* ---- K12/K13: BFV multiplication (rlwe_she.jl:247-262 with bfv.jl:34-40,172-226) ---------------
 * plan = (ℛ = small ctx limbs idx_s, ℛbig = big ctx limbs idx_b, t).  Supported basis relations:
 * ℛbig ⊇ ℛ as sets of primes, or disjoint (test/bfv_crt.jl); anything else TFHE_E_UNSUPPORTED.
 * c1, c2: [batch][2][ns][N]; out: [batch][3][ns][N]; coefficient domain. Exact (bit-identical to
 * the BigInt path). */
int tfhe_bfv_plan_create(tfhe_ctx *small, const int32_t *idx_s, int ns, tfhe_ctx *big, const int32_t *idx_b, int nb, uint64_t t, tfhe_bfv_plan **out);
int tfhe_bfv_plan_destroy(tfhe_bfv_plan *plan);
int tfhe_bfv_mul(tfhe_bfv_plan *plan, const uint64_t *c1, const uint64_t *c2, uint64_t *out, int64_t batch);
/* mul_expand (bfv.jl:34, switch :222-226) and mul_contract (bfv.jl:35-40) exposed on their own */
int tfhe_bfv_expand(tfhe_bfv_plan *plan, const uint64_t *src, uint64_t *dst, int64_t count);   /* [count][ns][N] -> [count][nb][N] */
int tfhe_bfv_contract(tfhe_bfv_plan *plan, const uint64_t *src, uint64_t *dst, int64_t count); /* [count][nb][N] -> [count][ns][N] */
/* c1*c2 followed by keyswitch(evk, ·) on the small ring (the BASELINE.json "ciphertext-mul +
 * relinearize" unit): out [batch][2][ns][N].  The key ring is ℛ itself (special == 0), which must be
 * limbs 0..ns-1 of the small ctx; evk: [n_digits][2][ns][N]. */
int tfhe_bfv_mul_relin(tfhe_bfv_plan *plan, const uint64_t *evk, int n_digits, const uint64_t *c1, const uint64_t *c2, uint64_t *out, int64_t batch);
/* expand/contract kernel family: 0 = auto (register-resident constant-folded kernels when ℛbig ⊇ ℛ and the
 * limb counts are instantiated, else the general kernels), 1 = force the general kernels (cross-check). */
int tfhe_bfv_plan_set_variant(tfhe_bfv_plan *plan, int variant);
/* ciphertexts processed per internal chunk (workspace = chunk * (7 nb + 3 ns) * N * 8 bytes); 0 = default (256) */
int tfhe_bfv_plan_set_chunk(tfhe_bfv_plan *plan, int chunk);

/* ---- multi-GPU: the final gather (the reference is single-process) ---------------------------------------
 * A batch of independent ciphertexts shards by ciphertext over one process per GPU (contexts and keys replicated, tens of
 * MiB); nothing is exchanged while computing.  The one optional collective is the gather of per-rank results: an all-gather
 * over RCCL (xGMI), enqueued on the context's stream.  RCCL is bound at run time (dlopen; TFHE_RCCL_LIB overrides the
 * library path), so single-GPU use has no RCCL dependency.
 *   tfhe_comm_id     : rank 0 creates the 128-byte rendezvous id; the host side (MPI.jl / Distributed / torch.distributed)
 *                      hands it to every rank.
 *   tfhe_comm_create : collective over the ranks (after tfhe_set_device on each).  The rendezvous has a deadline
 *                      (TFHE_COMM_TIMEOUT_S, default 180 s): a rank that never joins is a TFHE_E_HIP with the story in
 *                      tfhe_last_error, not a hung job.
 *   tfhe_gather      : dst [nranks][words_per_rank] <- every rank's src [words_per_rank]; equal shard sizes (pad the last). */
int tfhe_comm_id(void *id_out /* 128 bytes */);
int tfhe_comm_create(const void *id, int nranks, int rank, tfhe_comm **out);
int tfhe_comm_destroy(tfhe_comm *comm);
int tfhe_gather(tfhe_comm *comm, tfhe_ctx *ctx, const uint64_t *src, uint64_t *dst, size_t words_per_rank);

/* ---- measurement hooks (bench.py): HIP events on the ctx stream --------------------------------
 * While enabled, every NTT kernel launch is bracketed by a pair of events; read returns the number of
 * launches, the number of limb-polynomials transformed and the summed kernel time, then resets. */
int tfhe_prof_enable(tfhe_ctx *ctx, int on);
int tfhe_prof_read(tfhe_ctx *ctx, int64_t *launches, int64_t *limb_polys, double *total_ms);
int tfhe_event_create(void **ev);
int tfhe_event_destroy(void *ev);
int tfhe_event_record(tfhe_ctx *ctx, void *ev);
int tfhe_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */

#ifdef __cplusplus
}
#endif
#endif /* TOYFHE_HIP_H */
