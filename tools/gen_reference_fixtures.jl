# gen_reference_fixtures.jl -- run the UNMODIFIED reference CPU path on the committed seeded inputs and write its outputs.
#
# STATUS: never executed (no Julia in the build image).  On a machine with Julia and the reference's Manifest instantiated:
#
#     cd <ToyFHE.jl checkout>
#     julia --project=. <repo>/tools/gen_reference_fixtures.jl <repo>/tests/golden/ref_julia
#
# reads every <case>/case.json + in_*.tfhe (written by tools/make_reference_inputs.py, wire format of toyfhe.jl_amd/wire.py)
# and writes <case>/out.tfhe.  Commit the out.tfhe files: tests/test_reference_fixtures.py then holds the C / Python oracle
# (CPU suite) and the HIP path (GPU suite) to the reference's own bits for the cross-limb steps no reference-held datum
# constrains today -- multround / switch (src/bfv.jl:172-226), keyswitch (src/rlwe_she.jl:315-347, src/modulusraising.jl:35-49),
# modswitch (src/crt.jl:215-220), apply_galois_element (src/pow2_cyc_rings.jl:321-329).
#
# Nothing here re-implements arithmetic: residues go into the reference's own types (GaloisFields prime fields in a StructArray,
# src/crt.jl:150-156) and every operation is the reference's function.  No dependency beyond the reference's own (the case
# files are read with a few regular expressions instead of a JSON package).
using ToyFHE, GaloisFields, StructArrays, OffsetArrays, Random
using ToyFHE: NTT, CRTEncoded, KeySwitchKey, KeyComponent, CipherText, GaloisKey, ModulusRaised
using ToyFHE.NTT: NegacyclicRing, RingElement, coeffs_primal, coeffs_dual

# ---- wire format (toyfhe.jl_amd/wire.py) ------------------------------------------------------------------------------------
struct Blob
    kind::Int; logn::Int; L::Int; polys::Int; domain::Int; count::Int
    moduli::Vector{UInt64}; psi::Vector{UInt64}; res::Array{UInt64,4}          # res[n, l, poly, count] (column-major view of [count][polys][L][N])
end
function readblob(path)
    open(path) do io
        magic = read(io, 8); String(magic) == "TFHEWIRE" || error("bad magic in $path")
        ver, kind, logn, L, polys, domain = (read(io, UInt32) for _ in 1:6)
        count = read(io, UInt64); _mant = read(io, UInt64); _exp = read(io, Int32); _win = read(io, UInt32)
        ver == 1 || error("unsupported version")
        moduli = [read(io, UInt64) for _ in 1:L]; psi = [read(io, UInt64) for _ in 1:L]
        N = 1 << logn
        res = Array{UInt64,4}(undef, N, L, polys, count); read!(io, res)
        Blob(kind & 0xff, logn, L, polys, domain, count, moduli, psi, res)
    end
end
function writeblob(path, res::Array{UInt64,4}, moduli, psi; kind=1, domain=0)
    N, L, polys, count = size(res)
    open(path, "w") do io
        write(io, "TFHEWIRE"); write(io, UInt32(1)); write(io, UInt32(kind)); write(io, UInt32(trailing_zeros(N))); write(io, UInt32(L))
        write(io, UInt32(polys)); write(io, UInt32(domain)); write(io, UInt64(count)); write(io, UInt64(0)); write(io, Int32(0)); write(io, UInt32(0))
        write(io, UInt64.(moduli)); write(io, UInt64.(psi)); write(io, res)
    end
end

# ---- the reference's types from residues ----------------------------------------------------------------------------------------
function ring(N, moduli, psi)
    fields = Tuple(GaloisField(Int128(q)) for q in moduli)
    CT = CRTEncoded{length(moduli), Tuple{fields...}}
    NegacyclicRing{CT, N}(CT(Tuple(F(Int128(p)) for (F, p) in zip(fields, psi))))          # explicit ψ (pow2_cyc_rings.jl:27-37)
end
function element(ℛ, res::AbstractMatrix{UInt64}; dual=false)                 # res[n, l]
    CT = NTT.coefftype(ℛ); Fs = fieldtypes(ToyFHE.moduli(CT)); n = size(res, 1)
    sa = StructArray{CT}(Tuple(map(x -> F(Int128(x)), res[:, l]) for (l, F) in enumerate(Fs)))
    oa = OffsetArray(sa, 0:n-1)
    dual ? RingElement{ℛ}(nothing, oa) : RingElement{ℛ}(oa, nothing)
end
residues(re; dual=false) = hcat((UInt64[convert(Integer, x) for x in col] for col in StructArrays.fieldarrays((dual ? coeffs_dual(re) : coeffs_primal(re)).parent))...)
function stack(els; dual=false)                                                # -> res[n, l, poly, 1]
    m = [residues(e; dual=dual) for e in els]
    out = Array{UInt64,4}(undef, size(m[1], 1), size(m[1], 2), length(m), 1)
    for (k, x) in enumerate(m); out[:, :, k, 1] = x; end
    out
end
ciphertext(params, ℛ, b::Blob) = CipherText(params, Tuple(element(ℛ, b.res[:, :, p, 1]) for p in 1:b.polys))
function switchkey(params, ℛk, b::Blob)                                        # [digits][mask, masked][Lk][N], NTT domain
    b.domain == 1 || error("keys are stored in the NTT domain")
    KeySwitchKey(params, [KeyComponent(element(ℛk, b.res[:, :, 1, d]; dual=true), element(ℛk, b.res[:, :, 2, d]; dual=true)) for d in 1:b.count])
end

# ---- case files (flat JSON written by tools/make_reference_inputs.py) -----------------------------------------------------------
field(txt, key) = (m = match(Regex("\"$key\":\\s*(\"[^\"]*\"|\\[[^\\]]*\\]|-?\\d+)"), txt); m === nothing ? nothing : m.captures[1])
ints(s) = s === nothing ? nothing : [parse(UInt64, x.match) for x in eachmatch(r"\d+", s)]
int(s) = s === nothing ? nothing : parse(Int, s)
str(s) = strip(s, '"')

function run_case(dir)
    txt = read(joinpath(dir, "case.json"), String)
    op = str(field(txt, "op")); N = int(field(txt, "N")); q = ints(field(txt, "moduli")); ψ = ints(field(txt, "psi"))
    ℛ = ring(N, q, ψ)
    out = joinpath(dir, "out.tfhe")
    if op == "ring_mul"
        a, b = readblob(joinpath(dir, "in_a.tfhe")), readblob(joinpath(dir, "in_b.tfhe"))
        r = element(ℛ, a.res[:, :, 1, 1]) * element(ℛ, b.res[:, :, 1, 1])                                   # pow2_cyc_rings.jl:147-173
        writeblob(out, stack([r]), q, ψ)
    elseif op == "galois"
        a = readblob(joinpath(dir, "in_a.tfhe")); gs = ints(field(txt, "galois_elements"))
        writeblob(out, stack([NTT.apply_galois_element(element(ℛ, a.res[:, :, 1, 1]), Int(g)) for g in gs]), q, ψ)   # pow2_cyc_rings.jl:321-329
    elseif op in ("bfv_enc_mul", "bfv_contract")
        qb = ints(field(txt, "big_moduli")); ψb = ints(field(txt, "big_psi")); t = int(field(txt, "t"))
        ℛbig = ring(N, qb, ψb); ℛplain = plaintext_space(ℛ, t)
        params = BFVParams(ℛ, ℛbig, ℛplain, 0, 3.2, div(NTT.modulus(NTT.coefftype(ℛ)), NTT.modulus(NTT.coefftype(ℛplain))))
        if op == "bfv_enc_mul"
            c1 = ciphertext(params, ℛ, readblob(joinpath(dir, "in_c1.tfhe"))); c2 = ciphertext(params, ℛ, readblob(joinpath(dir, "in_c2.tfhe")))
            writeblob(out, stack(collect(ToyFHE.enc_mul(c1, c2))), q, ψ)                                   # rlwe_she.jl:247-262
        else
            e = element(ℛbig, readblob(joinpath(dir, "in_e.tfhe")).res[:, :, 1, 1])
            writeblob(out, stack(ToyFHE.mul_contract(params, [e])), q, ψ)                                   # bfv.jl:35-40
        end
    elseif op in ("keyswitch", "rotate")
        special = int(field(txt, "special")) == 1
        ctb, kb = readblob(joinpath(dir, "in_ct.tfhe")), readblob(joinpath(dir, "in_evk.tfhe"))
        if special
            ℛk = ring(N, ints(field(txt, "key_moduli")), ints(field(txt, "key_psi")))
            params = ModulusRaised(CKKSParams(ℛk, 0, 3.2))                                                  # modulusraising.jl:20
            ℛc = ℛ_cipher(params)
        else
            ℛk = ℛ; params = CKKSParams(ℛ, 0, 3.2); ℛc = ℛ
        end
        ek = switchkey(params, ℛk, kb); c = ciphertext(params, ℛc, ctb)
        r = op == "rotate" ? ToyFHE.rotate(GaloisKey(int(field(txt, "galois_element")), ek), c) : ToyFHE.keyswitch(ek, c)   # rlwe_she.jl:315-359
        writeblob(out, stack(collect(r.cs)), q, ψ)
    elseif op == "modswitch"
        b = readblob(joinpath(dir, "in_ct.tfhe"))
        rs = [ToyFHE.modswitch(element(ℛ, b.res[:, :, p, 1])) for p in 1:b.polys]                           # crt.jl:215-228
        writeblob(out, stack(rs), q[1:end-1], ψ[1:end-1])
    else
        error("unknown op $op in $dir")
    end
    println("wrote ", out)
end

root = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden", "ref_julia")
for d in sort(readdir(root))
    isfile(joinpath(root, d, "case.json")) && run_case(joinpath(root, d))
end
