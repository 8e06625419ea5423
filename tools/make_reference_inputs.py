#!/usr/bin/env python3
"""Seeded INPUTS for reference-generated fixtures (nothing the reference holds constrains the cross-limb steps).

The reference cannot run in the build image (pure Julia, no toolchain), so the fixture loop is split:
  1. this script (here)              -> tests/golden/ref_julia/<case>/{case.json, in_*.tfhe}      committed
  2. tools/gen_reference_fixtures.jl (a machine with Julia + the reference's Manifest): runs the UNMODIFIED reference CPU path
     on those inputs                 -> tests/golden/ref_julia/<case>/out.tfhe                    commit them
  3. tests/test_reference_fixtures.py: oracle (CPU) and HIP path (GPU) against out.tfhe, bit for bit; skipped, loudly, while
     no out.tfhe exists.
Files are the wire format of toyfhe.jl_amd/wire.py.  Cases follow the reference's own test parameter sets (cited in case.json).
usage: python tools/make_reference_inputs.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import spec                                   # noqa: E402  (prime chains and psi by the reference's rules)
from tests import helpers as H                            # noqa: E402
from toyfhe_jl_amd import wire                            # noqa: E402  (pure numpy)

OUT = os.path.join(ROOT, "tests", "golden", "ref_julia")


def psis(qs, N):
    return [spec.minimal_primitive_root(int(q), 2 * N) for q in qs]


def put(case, name, res, qs, N, **kw):
    os.makedirs(os.path.join(OUT, case), exist_ok=True)
    open(os.path.join(OUT, case, name), "wb").write(wire.dump(res, [int(q) for q in qs], psis(qs, N), **kw))


def meta(case, **kw):
    os.makedirs(os.path.join(OUT, case), exist_ok=True)
    json.dump(kw, open(os.path.join(OUT, case, "case.json"), "w"), indent=1)


def main():
    rng = np.random.default_rng(0xF1C5)
    # ---- ring product (pow2_cyc_rings.jl:147-173) at test/ckks_rotate.jl:9-16 -------------------------------------------------
    N = 16
    qs = H.chain(40, 2, N)
    put("ring_mul", "in_a.tfhe", H.rand_residues(rng, qs, (1, 1), N), qs, N)
    put("ring_mul", "in_b.tfhe", H.rand_residues(rng, qs, (1, 1), N), qs, N)
    meta("ring_mul", op="ring_mul", N=N, moduli=[int(q) for q in qs], psi=psis(qs, N), reference="src/pow2_cyc_rings.jl:147-173; ring of test/ckks_rotate.jl:9-16")
    # ---- apply_galois_element (pow2_cyc_rings.jl:321-329) ----------------------------------------------------------------------
    put("galois", "in_a.tfhe", H.rand_residues(rng, qs, (1, 1), N), qs, N)
    meta("galois", op="galois", N=N, moduli=[int(q) for q in qs], psi=psis(qs, N), galois_elements=[3, 5, 2 * N - 1], reference="src/pow2_cyc_rings.jl:321-329")
    # ---- BFV enc_mul: switch -> tensor -> multround -> switch (bfv.jl:34-40, 172-226) at test/bfv_crt.jl:9-20 ---------------------
    N = 2048
    ch = H.chain(50, 6, N)
    qs, qb = ch[:2], ch[2:]
    put("bfv_enc_mul", "in_c1.tfhe", H.rand_residues(rng, qs, (1, 2), N), qs, N)
    put("bfv_enc_mul", "in_c2.tfhe", H.rand_residues(rng, qs, (1, 2), N), qs, N)
    meta("bfv_enc_mul", op="bfv_enc_mul", N=N, moduli=[int(q) for q in qs], psi=psis(qs, N), big_moduli=[int(q) for q in qb], big_psi=psis(qb, N), t=53,
         reference="src/rlwe_she.jl:247-262 + src/bfv.jl:34-40,172-226; parameters of test/bfv_crt.jl:9-30")
    # ---- multround edge values: residues of x t^-1 for x at the rounding boundaries, through mul_contract alone ------------------
    big, small = spec.Ring(N, qb), spec.Ring(N, qs)
    y = H.rand_residues(rng, qb, (1, 1), N)
    tinv = pow(53, -1, big.Q)
    edges = [0, 1, big.Q - 1, big.Q // 2, big.Q // 2 + 1, small.Q // 2, small.Q // 2 + 1, small.Q, 5 * small.Q + small.Q // 2, 5 * small.Q + small.Q // 2 + 1]
    for k, x in enumerate(edges):
        y[0, 0, :, k] = [(x * tinv) % big.Q % int(p) for p in qb]
    put("bfv_contract", "in_e.tfhe", y, qb, N)
    meta("bfv_contract", op="bfv_contract", N=N, moduli=[int(q) for q in qs], psi=psis(qs, N), big_moduli=[int(q) for q in qb], big_psi=psis(qb, N), t=53,
         reference="mul_contract, src/bfv.jl:35-40 (multround :172-190 with div_hacks.jl:120-135 ties-away, switch :202-226)")
    # ---- RNS-digit key switch without special prime (rlwe_she.jl:315-347) on the bfv_crt ring, 3-element input ---------------------
    put("keyswitch_rns", "in_ct.tfhe", H.rand_residues(rng, qs, (1, 3), N), qs, N)
    put("keyswitch_rns", "in_evk.tfhe", H.uniform_evk(rng, qs, 2, N), qs, N, kind=wire.KIND_KEY, domain=1)
    meta("keyswitch_rns", op="keyswitch", special=0, N=N, moduli=[int(q) for q in qs], psi=psis(qs, N), reference="src/rlwe_she.jl:315-347 (RNS digits :326-329)")
    # ---- special-prime key switch (modulusraising.jl:20-49) at test/ckks_modraise.jl:10-20, and a rotation -------------------------
    N = 32
    qk = H.chain(40, 3, N)
    put("keyswitch_special", "in_ct.tfhe", H.rand_residues(rng, qk[:2], (1, 2), N), qk[:2], N)
    put("keyswitch_special", "in_evk.tfhe", H.uniform_evk(rng, qk, 3, N), qk, N, kind=wire.KIND_KEY, domain=1)
    meta("keyswitch_special", op="keyswitch", special=1, N=N, moduli=[int(q) for q in qk[:2]], psi=psis(qk[:2], N), key_moduli=[int(q) for q in qk], key_psi=psis(qk, N),
         reference="src/modulusraising.jl:20-49 over src/rlwe_she.jl:315-347; ring of test/ckks_modraise.jl:10-20")
    put("rotate_special", "in_ct.tfhe", H.rand_residues(rng, qk[:2], (1, 2), N), qk[:2], N)
    put("rotate_special", "in_evk.tfhe", H.uniform_evk(rng, qk, 3, N), qk, N, kind=wire.KIND_KEY, domain=1)
    meta("rotate_special", op="rotate", special=1, galois_element=pow(3, 2 * N - 1, 2 * N), N=N, moduli=[int(q) for q in qk[:2]], psi=psis(qk[:2], N),
         key_moduli=[int(q) for q in qk], key_psi=psis(qk, N), reference="rotate, src/rlwe_she.jl:355-359")
    # ---- rescale (crt.jl:215-236) ------------------------------------------------------------------------------------------------
    put("modswitch", "in_ct.tfhe", H.rand_residues(rng, qk, (1, 2), N), qk, N)
    meta("modswitch", op="modswitch", N=N, moduli=[int(q) for q in qk], psi=psis(qk, N), reference="modswitch, src/crt.jl:215-228 (floor semantics: unsigned representative of the dropped limb)")
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
