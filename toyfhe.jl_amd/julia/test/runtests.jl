# runtests.jl -- the reference's own tests, re-run on device storage NEXT TO the reference's CPU storage, residue for residue.
#
# STATUS: never executed (the build image has no Julia; SURVEY.md 8c).  This file is the one command a maintainer with Julia,
# the reference's Manifest and a gfx950 GPU runs to validate the shim (and thereby turn `parity` from "partial" to pinned):
#
#     cd <ToyFHE.jl checkout>                                   # the UNMODIFIED reference, Manifest instantiated
#     TOYFHE_HIP_LIB=<repo>/toyfhe.jl_amd/libtoyfhe_hip.so \
#         julia --project=. <repo>/toyfhe.jl_amd/julia/test/runtests.jl
#
# Every test takes the reference test's rings / parameters (cited), generates keys and ciphertexts with the REFERENCE code on
# the REFERENCE's storage, copies them to the device (ToyFHEHIP.todevice), runs the same operation through both storages --
# the reference's generic code dispatches on the storage type parameter (src/pow2_cyc_rings.jl:93-96) -- and compares the
# residues with ==.  Integer paths must agree exactly; CKKS decode within the tolerance stated in SURVEY.md 8(a18).
using Test, Random, OffsetArrays, StructArrays, Primes, GaloisFields
using ToyFHE
using ToyFHE: NTT, CRTEncoded, KeySwitchKey, KeyComponent, CipherText, GaloisKey, EvalMultKey, ModulusRaised
using ToyFHE.NTT: NegacyclicRing, RingElement, coeffs_primal, coeffs_dual

include(joinpath(@__DIR__, "..", "ToyFHEHIP.jl"))
using .ToyFHEHIP
const H = ToyFHEHIP

# ---- helpers ----------------------------------------------------------------------------------------------------------------
residues(re::RingElement) = [UInt64[convert(Integer, x) for x in col] for col in StructArrays.fieldarrays(coeffs_primal(re).parent)]
residues_dev(re::RingElement) = residues(H.tohost(re)[1])
dev(re::RingElement) = H.todevice(re)
dev(c::CipherText{Enc}) where {Enc} = CipherText{Enc}(c.params, map(dev, c.cs))
dev(kc::KeyComponent) = KeyComponent(dev(kc.mask), dev(kc.masked))
dev(ek::KeySwitchKey) = KeySwitchKey(ek.params, map(dev, ek.key))
dev(gk::GaloisKey) = GaloisKey(gk.galois_element, dev(gk.key))
dev(ek::EvalMultKey) = EvalMultKey(dev(ek.key))
samebits(host::RingElement, device::RingElement) = residues(host) == residues_dev(device)
samebits(host::CipherText, device::CipherText) = length(host.cs) == length(device.cs) && all(samebits(a, b) for (a, b) in zip(host.cs, device.cs))
samebits(host::Tuple, device::Tuple) = length(host) == length(device) && all(samebits(a, b) for (a, b) in zip(host, device))

function rns_ring(n, primes)
    CT = CRTEncoded{length(primes), Tuple{GaloisField.(primes)...}}
    NegacyclicRing{CT, n}(GaloisFields.minimal_primitive_root(CT, 2n))
end
function chain(start, k, n)
    ps = [nextprime(start, 1; interval=2n)]
    while length(ps) < k
        push!(ps, nextprime(ps[end] + 2n, 1; interval=2n))
    end
    ps
end

# ---- ring layer: nntt / inntt / * + - (src/pow2_cyc_rings.jl:147-224, 295-318; src/crt.jl:247-267) -----------------------------
@testset "ring layer (test/ckks_rotate.jl:9-16 ring; bfv_crt.jl:9-20 ring)" begin
    for (n, ps) in ((2^4, chain(Int128(2)^40 + 1, 2, 2^4)), (2048, chain(Int128(2)^50 + 1, 2, 2048)))
        ℛ = rns_ring(n, ps)
        rng = MersenneTwister(1)
        a = rand(rng, ToyFHE.RingSampler(ℛ, ToyFHE.DiscreteUniform(ToyFHE.NTT.coefftype(ℛ))))
        b = rand(rng, ToyFHE.RingSampler(ℛ, ToyFHE.DiscreteUniform(ToyFHE.NTT.coefftype(ℛ))))
        da, db = dev(a), dev(b)
        @test samebits(a * b, da * db)
        @test samebits(a + b, da + db)
        @test samebits(a - b, da - db)
        @test samebits(-a, -da)
        # the NTT domain is user visible (src/encoding.jl:35-43): natural order, psi = the ring's psi
        hd = [UInt64[convert(Integer, x) for x in col] for col in StructArrays.fieldarrays(coeffs_dual(a).parent)]
        dd = [UInt64[convert(Integer, x) for x in col] for col in StructArrays.fieldarrays(H.download(coeffs_dual(da).parent))]
        @test hd == dd
        for g in (3, 5, 2n - 1)
            @test samebits(NTT.apply_galois_element(a, g), NTT.apply_galois_element(da, g))     # pow2_cyc_rings.jl:321-329
        end
    end
end

# ---- scalar-weighted sums in one pass (tfhe_lincomb / tfhe_lincomb_many; examples/encrypted_mnist/infer.jl:127-131) -----------
@testset "lincomb: sum_k w_k .* a_k against scalar_mul and + (pow2_cyc_rings.jl:177-185, 200-214)" begin
    n = 2^6
    ℛ = rns_ring(n, chain(Int128(2)^40 + 1, 3, n))
    rng = MersenneTwister(3)
    as = [rand(rng, ToyFHE.RingSampler(ℛ, ToyFHE.DiscreteUniform(ToyFHE.NTT.coefftype(ℛ)))) for _ in 1:5]
    das = map(dev, as)
    ws = [3, 70000, 1, 2^33 + 5, 12]
    want = sum(w * a for (w, a) in zip(ws, as))
    got = H.lincomb(ws, [coeffs_primal(x) for x in das])
    @test residues(want) == [UInt64[convert(Integer, x) for x in col] for col in StructArrays.fieldarrays(H.download(got.parent))]
    rows = [ws, reverse(ws)]
    gots = H.lincomb_many(rows, [coeffs_primal(x) for x in das])
    for (row, g) in zip(rows, gots)
        w2 = sum(w * a for (w, a) in zip(row, as))
        @test residues(w2) == [UInt64[convert(Integer, x) for x in col] for col in StructArrays.fieldarrays(H.download(g.parent))]
    end
end

# ---- BFV over RNS with a disjoint extension basis (test/bfv_crt.jl:8-47) ------------------------------------------------------
@testset "bfv_crt: enc_mul, multround / switch (src/bfv.jl:34-40, 172-226)" begin
    n = 2048
    p = chain(Int128(2)^50 + 1, 6, n)
    ℛ, ℛbig = rns_ring(n, p[1:2]), rns_ring(n, p[3:6])
    ℛplain = plaintext_space(ℛ, 53)
    params = BFVParams(ℛ, ℛbig, ℛplain, 0, 3.2,
                       div(NTT.modulus(NTT.coefftype(ℛ)), NTT.modulus(NTT.coefftype(ℛplain))))
    kp = keygen(params)
    plain = zero(plaintext_space(params)); plain[0] = 6
    c = encrypt(kp, plain)
    dc = dev(c)
    y, dy = c * c, dc * dc
    @test samebits(y, dy)                                                # tfhe_bfv_mul against the BigInt path, bit for bit
    @test decrypt(kp, CipherText(c.params, map(x -> H.tohost(x)[1], dy.cs)))[0] == 0x24
    z, dz = y * c, dy * dc                                               # 3 x 2 components: the generic convolution (rlwe_she.jl:255-258)
    @test samebits(z, dz)
    ek = keygen(EvalMultKey, kp.priv)
    @test samebits(keyswitch(ek.key, y), keyswitch(dev(ek).key, dy))     # RNS digits (rlwe_she.jl:326-329, 340-344)
    @test samebits(c + c, dc + dc) && samebits(c - c, dc - dc)
end

# ---- digit-window key switch on a single-modulus ring (test/bfv_keyswitch.jl:5-27; rlwe_she.jl:330-338) -----------------------
@testset "bfv_keyswitch: base-2^w digits" begin
    n = 1024
    p = chain(Int128(2)^50 + 1, 4, n)
    ℛ, ℛbig = rns_ring(n, p[1:1]), rns_ring(n, p)
    ℛplain = plaintext_space(ℛ, 7)
    params = BFVParams(ℛ, ℛbig, ℛplain, 1, 3.2, div(NTT.modulus(NTT.coefftype(ℛ)), NTT.modulus(NTT.coefftype(ℛplain))))
    kp = keygen(params)
    ek = keygen(EvalMultKey, kp.priv)
    plain = zero(plaintext_space(params)); plain[0] = 2
    c = encrypt(kp, plain); sq = c * c
    @test samebits(keyswitch(ek.key, sq), keyswitch(dev(ek).key, dev(sq)))
end

# ---- BGV (test/bgv_triv.jl) ---------------------------------------------------------------------------------------------------
@testset "bgv_triv: tensor without scale-round (rlwe_she.jl:39-40, 247-262)" begin
    n = 1024
    ℛ = rns_ring(n, chain(Int128(2)^50 + 1, 2, n))
    params = BGVParams(ℛ, plaintext_space(ℛ, 17), 8 / sqrt(2pi))
    kp = keygen(params)
    plain = zero(plaintext_space(params)); plain[0] = 3
    c = encrypt(kp, plain)
    @test samebits(c * c, dev(c) * dev(c))
end

# ---- CKKS rescale (test/ckks_modswitch.jl; src/crt.jl:215-236, ckksencoding.jl:127-130) ----------------------------------------
@testset "ckks_modswitch: modswitch with the unsigned representative of the dropped limb" begin
    N = 2^5
    ℛ = rns_ring(N, chain(Int128(2)^40 + 1, 3, N))
    params = CKKSParams(ℛ, 0, 3.2)
    kp = keygen(params)
    Tscale = FixedRational{2^40}
    plain = CKKSEncoding{Tscale}(zero(ℛ)); plain .= OffsetArray(1:div(N, 2), 0:div(N, 2)-1)
    c = encrypt(kp, plain); dc = dev(c)
    @test samebits(modswitch(c * c), modswitch(dc * dc))
    for (a, b) in zip(c.cs, dc.cs)
        @test samebits(ToyFHE.modswitch(a), ToyFHE.modswitch(b))
        @test samebits(ToyFHE.modswitch_drop(a), ToyFHE.modswitch_drop(b))
    end
end

# ---- special-prime key switch (test/ckks_modraise.jl:10-30; src/modulusraising.jl:20-49) ---------------------------------------
@testset "ckks_modraise: ModulusRaised keyswitch, floor(./P)" begin
    N = 2^5
    ℛ = rns_ring(N, chain(Int128(2)^40 + 1, 3, N))
    params = ModulusRaised(CKKSParams(ℛ, 0, 3.2))
    kp = keygen(params)
    Tscale = FixedRational{2^40}
    plain = CKKSEncoding{Tscale}(zero(ℛ_cipher(params))); plain .= OffsetArray(1:div(N, 2), 0:div(N, 2)-1)
    c = encrypt(kp, plain); dc = dev(c)
    ek = ToyFHE.make_eval_key(Random.GLOBAL_RNG, kp.priv.secret => kp.priv)
    @test samebits(keyswitch(ek, c), keyswitch(dev(ek), dc))
    sq, dsq = c * c, dc * dc
    rk = keygen(EvalMultKey, kp.priv)
    @test samebits(keyswitch(rk.key, sq), keyswitch(dev(rk).key, dsq))
    @test samebits(modswitch(keyswitch(rk.key, sq)), modswitch(keyswitch(dev(rk).key, dsq)))
end

# ---- rotations (test/ckks_rotate.jl:9-45; rlwe_she.jl:355-359; pow2_cyc_rings.jl:321-329) --------------------------------------
@testset "ckks_rotate: galois + key switch, RNS and window digits" begin
    N = 2^4
    ℛ = rns_ring(N, chain(Int128(2)^40 + 1, 2, N))
    Tscale = FixedRational{2^60}
    for (params, Rc) in ((CKKSParams(ℛ, 1, 3.2), ℛ), (ModulusRaised(CKKSParams(rns_ring(N, chain(Int128(2)^40 + 1, 3, N)), 0, 3.2)), nothing))
        kp = keygen(params)
        R = Rc === nothing ? ℛ_cipher(params) : Rc
        plain = CKKSEncoding{Tscale}(zero(R)); plain .= OffsetArray(1:div(N, 2), 0:div(N, 2)-1)
        c = encrypt(kp, plain); dc = dev(c)
        for steps in (1, 3)
            gk = keygen(GaloisKey, kp.priv; steps=steps)
            @test samebits(ToyFHE.rotate(gk, c), ToyFHE.rotate(dev(gk), dc))
        end
    end
end

# ---- the diagonal matrix product (test/ckks_matmul.jl:33-41; examples/encrypted_mnist/infer.jl:140-149) ------------------------
@testset "ckks_matmul: rotate_many / matmul_diag against the reference's loop" begin
    N = 2^5
    ℛ = rns_ring(N, chain(Int128(2)^40 + 1, 4, N))
    params = ModulusRaised(CKKSParams(ℛ, 0, 3.2))
    kp = keygen(params)
    R = ℛ_cipher(params)
    Tscale = FixedRational{2^40}
    plain = CKKSEncoding{Tscale}(zero(R)); plain .= OffsetArray(collect(1.0:div(N, 2)), 0:div(N, 2)-1)
    c = encrypt(kp, plain); dc = dev(c)
    gks = [keygen(GaloisKey, kp.priv; steps=s) for s in 1:3]
    diags = [rand(div(N, 2)) for _ in 1:4]
    # the reference's loop: result += rotated_k .* diagonal_k (each rotation from c: one key per step)
    want = diags[1] .* c
    for (k, gk) in enumerate(gks)
        want = want + diags[k + 1] .* ToyFHE.rotate(gk, c)
    end
    dgks = map(dev, gks)
    rots = H.rotate_many(dgks, dc)
    for (gk, r) in zip(gks, rots)
        @test samebits(ToyFHE.rotate(gk, c), r)
    end
    enc(v) = (p = CKKSEncoding{Tscale}(zero(R)); p .= OffsetArray(v, 0:length(v)-1); dev(convert(NTT.RingElement, p)))
    got = H.matmul_diag(dgks, [enc(v) for v in diags], dc)
    @test typeof(got).parameters[1] == typeof(want).parameters[1]          # CKKSEncoding{Tscale^2}: the product's scale
    @test samebits(want, got)
end

# ---- CKKS encode / decode on the device (float: stated tolerance, SURVEY.md 8 a18) --------------------------------------------
@testset "ckks encode / decode (ckksencoding.jl:43-97)" begin
    N = 2^6
    ℛ = rns_ring(N, chain(Int128(2)^40 + 1, 2, N))
    denom = 2^40
    slots = ComplexF64[complex(k, -0.5k) for k in 1:div(N, 2)]
    plain = CKKSEncoding{FixedRational{denom}}(zero(ℛ)); plain .= OffsetArray(slots, 0:div(N, 2)-1)
    host = convert(NTT.RingElement, plain)
    devel = H.encode(typeof(dev(host)), slots, denom)
    a, b = residues(host), residues_dev(devel)
    q = [NTT.modulus(F) for F in fieldtypes(ToyFHE.moduli(NTT.coefftype(ℛ)))]
    @test all(all(min(mod(Int128(x) - Int128(y), qq), mod(Int128(y) - Int128(x), qq)) <= 1 for (x, y) in zip(ra, rb)) for (ra, rb, qq) in zip(a, b, q))
    back = H.decode(devel, denom)
    @test maximum(abs.(back .- slots)) <= 8 * log2(N) * eps(Float64) * maximum(abs.(slots)) + 2.0^-38
end
