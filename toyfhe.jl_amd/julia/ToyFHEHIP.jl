# ToyFHEHIP.jl -- the Julia side of the drop-in: device storage for ToyFHE's NegacyclicRing /
# RingElement / CipherText, bound to libtoyfhe_hip.so with ccall.
#
# STATUS: complete against include/toyfhe_hip.h and statically checked (tests/test_julia_shim_cpu.py parses every
# ccall below and compares symbol, arity and C types with the header and with the ctypes table of
# toyfhe.jl_amd/native.py; it also checks that every helper used here is defined here or imported).  It has NOT been
# executed: the build image has no Julia toolchain (SURVEY.md §8c).  The same call sequences are exercised from Python
# through ctypes (toyfhe.jl_amd/native.py, ring.py, she.py), which is the host mirror the tests run.  Validate on a
# machine with Julia >= 1.2 and the reference's Manifest before relying on it.
#
# How it plugs in: exactly like src/crt.jl:247-275 plugs the RNS NTT in -- by dispatch on the storage
# type parameter of RingElement{ℛ,Field,Storage} (src/pow2_cyc_rings.jl:93-96).  `HipVector` is that
# storage; the methods below override NTT.nntt / NTT.inntt, broadcast `+ - *`, `modswitch`, `modswitch_drop`,
# `apply_galois_element`, `keyswitch`, CKKS encode / decode and BFV `enc_mul` for it.  Everything else in ToyFHE
# (keygen, encrypt, decrypt, π, π⁻¹, CipherText + -) is generic code over those and runs unchanged.
module ToyFHEHIP

using ToyFHE
using ToyFHE: NTT, CRTEncoded, moduli, KeySwitchKey, CipherText, BFVParams, ModulusRaised, CKKSEncoding, FixedRational
using ToyFHE.NTT: NegacyclicRing, RingElement, RingCoeffs, degree, coeffs_primal, coeffs_dual
using OffsetArrays, StructArrays

const lib = get(ENV, "TOYFHE_HIP_LIB", "libtoyfhe_hip.so")

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:tfhe_last_error, lib), Cstring, ()))
    rc == -1 && throw(AssertionError(msg))              # pow2_cyc_rings.jl:31,61,116; rlwe_she.jl:318
    rc in (-3, -4) && throw(ToyFHE.UsageError(msg))     # rlwe_she.jl:223-225,233-235,248-250
    rc == -7 && error(msg)                              # crt.jl:270,274
    rc == -5 && throw(OutOfMemoryError())
    error("HIP: " * msg)
end

nlimbs(::Type{CRTEncoded{L,M}}) where {L,M} = L
limb_moduli(::Type{T}) where {T<:CRTEncoded} = UInt64[NTT.modulus(F) for F in fieldtypes(moduli(T))]

# ---- ring context: one per NegacyclicRing{CRTEncoded{L,...},N} value (pow2_cyc_rings.jl:27-37) ----------------------
mutable struct HipRing
    handle::Ptr{Cvoid}; N::Int; q::Vector{UInt64}
end
function make_ring(N::Integer, q::Vector{UInt64}, ψ::Vector{UInt64})
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:tfhe_ctx_create, lib), Cint, (Int64, Cint, Ptr{UInt64}, Ptr{UInt64}, Ptr{Ptr{Cvoid}}),
                N, length(q), q, ψ, h))
    r = HipRing(h[], N, q)
    finalizer(x -> ccall((:tfhe_ctx_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), r)
    r
end
const RINGS = Dict{Any,HipRing}()
function hipring(ℛ::NegacyclicRing{T,N}) where {T<:CRTEncoded,N}
    get!(RINGS, ℛ) do
        make_ring(N, limb_moduli(T), UInt64[convert(Integer, c) for c in ℛ.ψ.c])
    end
end
# limb-wise operations (+ - * neg, galois, rescale) depend on the moduli only, not on ψ: the broadcast hooks see the
# coefficient type T but not the ring value, so they use a context keyed on (T, N) whose ψ the library derives.
const MODRINGS = Dict{Any,HipRing}()
function modring(::Type{T}, N::Integer) where {T<:CRTEncoded}
    get!(MODRINGS, (T, N)) do
        make_ring(N, limb_moduli(T), zeros(UInt64, nlimbs(T)))
    end
end
sync(r::HipRing) = check(ccall((:tfhe_ctx_sync, lib), Cint, (Ptr{Cvoid},), r.handle))
# device-side ordering between two contexts (no host wait): r's later work runs after what `producer` has been given so far
wait_for(r::HipRing, producer::HipRing) = check(ccall((:tfhe_ctx_wait_for, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), r.handle, producer.handle))

# ---- device storage: [L][N] UInt64 residues, limb-major like StructArray field arrays (crt.jl:150-156) -------------
mutable struct HipVector{T} <: AbstractVector{T}
    ptr::Ptr{UInt64}; limbs::Int; n::Int
    function HipVector{T}(limbs::Integer, n::Integer) where T
        p = Ref{Ptr{Cvoid}}()
        check(ccall((:tfhe_malloc, lib), Cint, (Csize_t, Ptr{Ptr{Cvoid}}), 8 * limbs * n, p))
        v = new{T}(convert(Ptr{UInt64}, p[]), limbs, n)
        finalizer(x -> ccall((:tfhe_free, lib), Cint, (Ptr{Cvoid},), x.ptr), v)
        v
    end
end
Base.size(v::HipVector) = (v.n,)
Base.similar(::Type{HipVector{T}}, ::Type{T′}) where {T,T′} = HipVector{T′}      # crt.jl:196-197
words(v::HipVector) = v.limbs * v.n

function upload(sa::StructArray{T}) where {T<:CRTEncoded}
    cols = StructArrays.fieldarrays(sa); v = HipVector{T}(length(cols), length(sa))
    for (l, col) in enumerate(cols)
        host = UInt64[convert(Integer, x) for x in col]
        check(ccall((:tfhe_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.ptr + 8 * (l - 1) * v.n, host, 8 * v.n))
    end
    v
end
function upload(host::Vector{Float64})
    v = HipVector{Float64}(1, length(host))
    check(ccall((:tfhe_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.ptr, host, 8 * length(host)))
    v
end
function download(v::HipVector{T}) where {T<:CRTEncoded}
    host = Matrix{UInt64}(undef, v.n, v.limbs)
    check(ccall((:tfhe_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), host, v.ptr, 8 * length(host)))
    StructArray{T}(tuple((map(F, host[:, l]) for (l, F) in enumerate(fieldtypes(moduli(T))))...))
end
function download(v::HipVector{Float64})
    host = Vector{Float64}(undef, v.n)
    check(ccall((:tfhe_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), host, v.ptr, 8 * v.n))
    host
end
# element access goes through the host (getindex / setindex! on RingElement, pow2_cyc_rings.jl:140-145)
Base.getindex(v::HipVector{T}, i::Int) where {T<:CRTEncoded} = download(v)[i]
function Base.setindex!(v::HipVector{T}, x, i::Int) where {T<:CRTEncoded}
    host = download(v); host[i] = convert(T, x); new = upload(host)
    check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), modring(T, v.n).handle, v.ptr, new.ptr, 8 * words(v)))
    sync(modring(T, v.n)); x
end
function Base.zero(o::OffsetVector{T,HipVector{T}}) where {T<:CRTEncoded}
    v = HipVector{T}(o.parent.limbs, o.parent.n)
    check(ccall((:tfhe_memset, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Csize_t), modring(T, v.n).handle, v.ptr, 0, 8 * words(v)))
    OffsetArray(v, axes(o)...)
end
# move a host ring element to the device / back
todevice(re::RingElement{ℛ,T}) where {ℛ,T<:CRTEncoded} =
    RingElement{ℛ}(OffsetArray(upload(coeffs_primal(re).parent), 0:degree(ℛ)-1), nothing)
tohost(re::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T}} =
    RingElement{ℛ}(OffsetArray(download(coeffs_primal(re).parent), 0:degree(ℛ)-1), nothing)

# ---- K1/K2: the NTT hooks, same shape as crt.jl:247-267 -------------------------------------------------------------
function NTT.nntt(rcs::RingCoeffs{ℛ,T,OffsetVector{T,S}})::RingCoeffs{ℛ} where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = rcs.coeffs.parent; dst = HipVector{T}(src.limbs, src.n)
    check(ccall((:tfhe_nntt, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                hipring(ℛ).handle, src.ptr, dst.ptr, 1, src.limbs, C_NULL))
    RingCoeffs{ℛ}(OffsetArray(dst, axes(rcs.coeffs)...))
end
function NTT.inntt(rcs::RingCoeffs{ℛ,T,OffsetVector{T,S}})::RingCoeffs{ℛ} where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = rcs.coeffs.parent; dst = HipVector{T}(src.limbs, src.n)
    check(ccall((:tfhe_inntt, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                hipring(ℛ).handle, src.ptr, dst.ptr, 1, src.limbs, C_NULL))
    RingCoeffs{ℛ}(OffsetArray(dst, axes(rcs.coeffs)...))
end

# ---- K3/K4: limb-wise broadcast (pow2_cyc_rings.jl:167,178-179,188-189,200-214) -------------------------------------
const DevVec{T} = OffsetVector{T,HipVector{T}}
function Base.broadcasted(::typeof(+), a::DevVec{T}, b::DevVec{T}) where {T<:CRTEncoded}
    dst = HipVector{T}(a.parent.limbs, a.parent.n)
    check(ccall((:tfhe_add, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                modring(T, dst.n).handle, a.parent.ptr, b.parent.ptr, dst.ptr, 1, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
function Base.broadcasted(::typeof(-), a::DevVec{T}, b::DevVec{T}) where {T<:CRTEncoded}
    dst = HipVector{T}(a.parent.limbs, a.parent.n)
    check(ccall((:tfhe_sub, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                modring(T, dst.n).handle, a.parent.ptr, b.parent.ptr, dst.ptr, 1, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
function Base.broadcasted(::typeof(*), a::DevVec{T}, b::DevVec{T}) where {T<:CRTEncoded}
    dst = HipVector{T}(a.parent.limbs, a.parent.n)
    check(ccall((:tfhe_mul, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                modring(T, dst.n).handle, a.parent.ptr, b.parent.ptr, dst.ptr, 1, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
# sum_k as[k] .* bs[k] in one device pass (tfhe_dot): the accumulation of the diagonal matrix-vector product, infer.jl:140-149
function dot(as::Vector{<:DevVec{T}}, bs::Vector{<:DevVec{T}}) where {T<:CRTEncoded}
    @assert length(as) == length(bs) && !isempty(as)
    dst = HipVector{T}(as[1].parent.limbs, as[1].parent.n)
    ap = Ptr{UInt64}[a.parent.ptr for a in as]
    bp = Ptr{UInt64}[b.parent.ptr for b in bs]
    check(ccall((:tfhe_dot, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{Ptr{UInt64}}, Ptr{Ptr{UInt64}}, Cint, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                modring(T, dst.n).handle, C_NULL, ap, bp, length(as), dst.ptr, 1, dst.limbs, C_NULL))
    OffsetArray(dst, axes(as[1])...)
end
function Base.broadcasted(::typeof(-), a::DevVec{T}) where {T<:CRTEncoded}
    dst = HipVector{T}(a.parent.limbs, a.parent.n)
    check(ccall((:tfhe_neg, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                modring(T, dst.n).handle, a.parent.ptr, dst.ptr, 1, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
# scalar_mul (pow2_cyc_rings.jl:177-185): `scalar .* coeffs`
function Base.broadcasted(::typeof(*), s::Union{Integer,CRTEncoded}, a::DevVec{T}) where {T<:CRTEncoded}
    scal = UInt64[convert(Integer, c) for c in convert(T, s).c]
    dst = HipVector{T}(a.parent.limbs, a.parent.n)
    check(ccall((:tfhe_scalar_mul, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                modring(T, dst.n).handle, scal, a.parent.ptr, dst.ptr, 1, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
Base.broadcasted(::typeof(*), a::DevVec{T}, s::Union{Integer,CRTEncoded}) where {T<:CRTEncoded} = Base.broadcasted(*, s, a)

# ---- K6/K7: modswitch / modswitch_drop / crtselect (crt.jl:185-236) --------------------------------------------------
function ToyFHE.modswitch(re::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = coeffs_primal(re).parent; ℛ′ = ToyFHE.drop_last(ℛ); T′ = eltype(ℛ′)
    dst = HipVector{T′}(src.limbs - 1, src.n)
    check(ccall((:tfhe_rescale, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                hipring(ℛ).handle, src.ptr, dst.ptr, 1, src.limbs, C_NULL))
    RingElement{ℛ′}(OffsetArray(dst, 0:src.n-1), nothing)
end
function select_limbs(ℛ, src::HipVector, ::Type{T′}, which) where {T′}
    idx = Int32[w - 1 for w in which]; dst = HipVector{T′}(length(idx), src.n)
    check(ccall((:tfhe_select_limbs, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}, Cint),
                hipring(ℛ).handle, src.ptr, dst.ptr, 1, src.limbs, idx, length(idx)))
    dst
end
function ToyFHE.crtselect(x::RingElement{ℛ,T,S}, which) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    ℛ′ = ToyFHE.crtselect(ℛ, which); T′ = eltype(ℛ′)
    sel(o) = o === nothing ? nothing : OffsetArray(select_limbs(ℛ, o.parent, T′, which), axes(o)...)
    RingElement{ℛ′}(sel(x.primal), sel(x.dual))
end
function ToyFHE.modswitch_drop(re::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    ℛ′ = ToyFHE.drop_last(ℛ); src = coeffs_primal(re).parent
    RingElement{ℛ′}(OffsetArray(select_limbs(ℛ, src, eltype(ℛ′), 1:src.limbs-1), 0:src.n-1), nothing)
end

# ---- K8: apply_galois_element (pow2_cyc_rings.jl:321-329) -----------------------------------------------------------
function NTT.apply_galois_element(re::RingElement{ℛ,T,S}, g::Integer) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = coeffs_primal(re).parent; dst = HipVector{T}(src.limbs, src.n)
    check(ccall((:tfhe_galois, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, UInt64, Int64, Cint, Ptr{Int32}),
                hipring(ℛ).handle, src.ptr, dst.ptr, g, 1, src.limbs, C_NULL))
    RingElement{ℛ}(OffsetArray(dst, 0:src.n-1), nothing)
end

# ---- ciphertext staging: the C ABI takes [polys][limbs][N] contiguously ------------------------------------------------
# pack: the coefficient-domain (dual = true: NTT-domain) components of a ciphertext / key, back to back.
function pack(ctx::HipRing, parts::Vector{<:HipVector}, ::Type{T}) where {T}
    limbs, n = parts[1].limbs, parts[1].n
    out = HipVector{T}(limbs * length(parts), n)
    for (k, p) in enumerate(parts)
        check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                    ctx.handle, out.ptr + 8 * (k - 1) * limbs * n, p.ptr, 8 * limbs * n))
    end
    out
end
pack(c::CipherText{Enc,P,<:RingElement{ℛ,T}}) where {Enc,P,ℛ,T} =
    pack(hipring(ℛ), HipVector[coeffs_primal(x).parent for x in c.cs], T)
# evaluation key: [digit][mask, masked][Lk][N], NTT domain (rlwe_she.jl:297, 340-344), packed once per key
const PACKED_KEYS = IdDict{Any,HipVector}()
function pack(ek::KeySwitchKey)
    get!(PACKED_KEYS, ek) do
        ℛk = NTT.ring(ek.key[1].mask); parts = HipVector[]
        for kc in ek.key
            push!(parts, coeffs_dual(kc.mask).parent); push!(parts, coeffs_dual(kc.masked).parent)
        end
        pack(hipring(ℛk), parts, eltype(ℛk))
    end
end
# unpack: `polys` ring elements of ℛ from a packed [polys][limbs][N] buffer (coefficient domain)
function unpack(buf::HipVector, ℛ, polys::Integer)
    T = eltype(ℛ); limbs = nlimbs(T); n = degree(ℛ); ctx = hipring(ℛ)
    els = map(1:polys) do k
        v = HipVector{T}(limbs, n)
        check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                    ctx.handle, v.ptr, buf.ptr + 8 * (k - 1) * limbs * n, 8 * limbs * n))
        RingElement{ℛ}(OffsetArray(v, 0:n-1), nothing)
    end
    sync(ctx)                                        # `buf` may be finalised as soon as we return
    tuple(els...)
end

# ---- K9-K11 / K14: keyswitch (rlwe_she.jl:315-347) as one fused call ---------------------------------------------------
function ToyFHE.keyswitch(ek::KeySwitchKey, c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) in (2, 3)                                                    # rlwe_she.jl:318
    keyring = NTT.ring(ek.key[1].mask); Lk = nlimbs(eltype(keyring)); level = nlimbs(T)
    ct = pack(c); out = HipVector{T}(2 * level, degree(ℛ))
    w = ToyFHE.relin_window(ek.params)
    if w != 0                                                                         # K14, rlwe_she.jl:330-338
        ek.params isa ModulusRaised && error("ModulusRaised with a digit window is not on the device path")
        check(ccall((:tfhe_keyswitch_window, lib), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Int64),
                    hipring(ℛ).handle, level, w, pack(ek).ptr, length(ek.key), ct.ptr, length(c.cs), out.ptr, 1))
        return CipherText{Enc}(c.params, unpack(out, ℛ, 2))
    end
    check(ccall((:tfhe_keyswitch, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Int64),
                hipring(keyring).handle, Lk, level, ek.params isa ModulusRaised ? 1 : 0, pack(ek).ptr, length(ek.key),
                ct.ptr, length(c.cs), out.ptr, 1))
    CipherText{Enc}(c.params, unpack(out, ℛ, 2))
end
# rotate(gk, c) = keyswitch(gk, apply_galois_element(c, g)) (rlwe_she.jl:355-359), fused on the device
function ToyFHE.rotate(gk::ToyFHE.GaloisKey, c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) == 2
    ek = gk.key; ToyFHE.relin_window(ek.params) != 0 && return ToyFHE.keyswitch(ek, ToyFHE.NTT.apply_galois_element(c, gk.galois_element))
    keyring = NTT.ring(ek.key[1].mask); Lk = nlimbs(eltype(keyring)); level = nlimbs(T)
    ct = pack(c); out = HipVector{T}(2 * level, degree(ℛ))
    check(ccall((:tfhe_rotate, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt64}, Cint, UInt64, Ptr{UInt64}, Ptr{UInt64}, Int64),
                hipring(keyring).handle, Lk, level, ek.params isa ModulusRaised ? 1 : 0, pack(ek).ptr, length(ek.key),
                gk.galois_element, ct.ptr, out.ptr, 1))
    CipherText{Enc}(c.params, unpack(out, ℛ, 2))
end

# hoisted rotations: [rotate(gk, c) for gk in gks] from one digit decomposition of c (tfhe_rotate_many)
function rotate_many(gks::Vector{<:ToyFHE.GaloisKey}, c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) == 2
    ek1 = gks[1].key; keyring = NTT.ring(ek1.key[1].mask); Lk = nlimbs(eltype(keyring)); level = nlimbs(T)
    ct = pack(c); n = degree(ℛ); out = HipVector{T}(2 * level * length(gks), n)
    keys = Ptr{UInt64}[pack(gk.key).ptr for gk in gks]; gs = UInt64[gk.galois_element for gk in gks]
    check(ccall((:tfhe_rotate_many, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{UInt64}}, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Ptr{UInt64}, Int64),
                hipring(keyring).handle, Lk, level, ek1.params isa ModulusRaised ? 1 : 0, keys, length(ek1.key), 0, gs, length(gks),
                ct.ptr, out.ptr, 1))
    map(1:length(gks)) do r
        part = HipVector{T}(2 * level, n)
        check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                    hipring(ℛ).handle, part.ptr, out.ptr + 8 * (r - 1) * 2 * level * n, 8 * 2 * level * n))
        CipherText{Enc}(c.params, unpack(part, ℛ, 2))
    end
end

# ---- CKKS encode / decode (ckksencoding.jl:56-97) on the device --------------------------------------------------------
# denom = mant * 2^exp2 with a 64-bit mant (exact for 2^k and for integers below 2^64 times 2^k; to 2^-63 otherwise)
function scale_parts(denom)
    denom > 0 || throw(AssertionError("scale must be positive"))
    r = Rational{BigInt}(denom); num, den = numerator(r), denominator(r)
    if ispow2(den)
        exp2 = -trailing_zeros(den); tz = trailing_zeros(num); num >>= tz; exp2 += tz
        num < big(2)^64 && return UInt64(num), Cint(exp2)
    end
    e = (ndigits(num, base=2) - ndigits(den, base=2)) - 63
    mant = e >= 0 ? round(BigInt, r / big(2)^e) : round(BigInt, r * big(2)^(-e))
    if mant >= big(2)^64
        mant >>= 1; e += 1
    end
    UInt64(mant), Cint(e)
end
function Base.convert(::Type{<:RingElement{ℛ,T,S}}, s::CKKSEncoding{FixedRational{denom}}) where {ℛ,T<:CRTEncoded,S<:HipVector{T},denom}
    mant, exp2 = scale_parts(denom); slots = upload(collect(reinterpret(Float64, collect(s.data))))
    out = HipVector{T}(nlimbs(T), degree(ℛ))
    check(ccall((:tfhe_ckks_encode, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, Cint, Ptr{Float64}, Ptr{UInt64}, Int64),
                hipring(ℛ).handle, out.limbs, mant, exp2, slots.ptr, out.ptr, 1))
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end
function ToyFHE.CKKSEncoding{FixedRational{denom}}(plain::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T},denom}
    mant, exp2 = scale_parts(denom); src = coeffs_primal(plain).parent; slots = HipVector{Float64}(1, degree(ℛ))
    check(ccall((:tfhe_ckks_decode, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, Cint, Ptr{UInt64}, Ptr{Float64}, Int64),
                hipring(ℛ).handle, src.limbs, mant, exp2, src.ptr, slots.ptr, 1))
    data = collect(reinterpret(ComplexF64, download(slots)))
    CKKSEncoding{FixedRational{denom}}(typeof(plain), OffsetArray(data, 0:degree(ℛ)÷2-1))
end

# ---- K12/K13: BFV enc_mul (rlwe_she.jl:247-262 + bfv.jl:34-40) ---------------------------------------------------------
# plan(params) = (ℛ, ℛbig, t) with the exact-conversion tables on the device, cached per BFVParams.
mutable struct HipBfvPlan
    handle::Ptr{Cvoid}
end
const PLANS = IdDict{Any,HipBfvPlan}()
function plan(params::BFVParams)
    get!(PLANS, params) do
        small, big = hipring(params.ℛ), hipring(params.ℛbig)
        t = UInt64(NTT.modulus(eltype(ToyFHE.plaintext_space(params))))
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:tfhe_bfv_plan_create, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Int32}, Cint, Ptr{Cvoid}, Ptr{Int32}, Cint, UInt64, Ptr{Ptr{Cvoid}}),
                    small.handle, C_NULL, length(small.q), big.handle, C_NULL, length(big.q), t, h))
        p = HipBfvPlan(h[])
        finalizer(x -> ccall((:tfhe_bfv_plan_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), p)
        p
    end
end
function ToyFHE.enc_mul(c1::CipherText{E,BFVParams,<:RingElement{ℛ,T,<:HipVector}}, c2::CipherText{E,BFVParams}) where {E,ℛ,T}
    c1.params !== c2.params && throw(ToyFHE.UsageError("Attempting to multiply ciphertexts with differing parameters"))
    (length(c1.cs) == 2 && length(c2.cs) == 2) || error("BFV enc_mul on the device takes 2-element ciphertexts")
    a, b = pack(c1), pack(c2); out = HipVector{T}(3 * nlimbs(T), degree(ℛ))
    check(ccall((:tfhe_bfv_mul, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                plan(c1.params).handle, a.ptr, b.ptr, out.ptr, 1))
    unpack(out, ℛ, 3)
end
# c1*c2 followed by keyswitch(ek, .) in one call (the BASELINE.json unit) for RNS-gadget keys on ℛ itself
function mul_relin(ek::KeySwitchKey, c1::CipherText{E,BFVParams,<:RingElement{ℛ,T,<:HipVector}}, c2::CipherText{E,BFVParams}) where {E,ℛ,T}
    c1.params !== c2.params && throw(ToyFHE.UsageError("Attempting to multiply ciphertexts with differing parameters"))
    a, b = pack(c1), pack(c2); out = HipVector{T}(2 * nlimbs(T), degree(ℛ))
    check(ccall((:tfhe_bfv_mul_relin, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Cint, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                plan(c1.params).handle, pack(ek).ptr, length(ek.key), a.ptr, b.ptr, out.ptr, 1))
    CipherText{E}(c1.params, unpack(out, ℛ, 2))
end

# ---- tensor for schemes whose mul_expand / mul_contract are the identity (BGV, CKKS; rlwe_she.jl:39-40,255-258) -----------
function ToyFHE.enc_mul(c1::CipherText{E,P,<:RingElement{ℛ,T,<:HipVector}}, c2::CipherText{E,P}) where {E,P,ℛ,T}
    c1.params !== c2.params && throw(ToyFHE.UsageError("Attempting to multiply ciphertexts with differing parameters"))
    (length(c1.cs) == 2 && length(c2.cs) == 2) || return invoke(ToyFHE.enc_mul, Tuple{CipherText,CipherText}, c1, c2)
    ctx = hipring(ℛ)
    a = pack(ctx, HipVector[coeffs_dual(x).parent for x in c1.cs], T); b = pack(ctx, HipVector[coeffs_dual(x).parent for x in c2.cs], T)
    out = HipVector{T}(3 * nlimbs(T), degree(ℛ))
    check(ccall((:tfhe_tensor, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, a.ptr, b.ptr, out.ptr, 1, nlimbs(T), C_NULL))
    limbs, n = nlimbs(T), degree(ℛ)
    els = map(1:3) do k                                        # NTT-domain results: dual-only ring elements
        v = HipVector{T}(limbs, n)
        check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                    ctx.handle, v.ptr, out.ptr + 8 * (k - 1) * limbs * n, 8 * limbs * n))
        RingElement{ℛ}(nothing, OffsetArray(v, 0:n-1))
    end
    sync(ctx)
    tuple(els...)
end

# ---- device samplers for RingSampler (poly.jl:7-23, crt.jl:277-279) ------------------------------------------------------
mutable struct HipRng
    seed::UInt64; next_poly::UInt64
end
function sample_uniform(rng::HipRng, ℛ)
    T = eltype(ℛ); out = HipVector{T}(nlimbs(T), degree(ℛ))
    check(ccall((:tfhe_sample_uniform, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, UInt32, UInt64, Ptr{UInt64}, Int64),
                hipring(ℛ).handle, out.limbs, rng.seed, 0, rng.next_poly, out.ptr, 1))
    rng.next_poly += 1
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end
function sample_gaussian(rng::HipRng, ℛ, σ::Real, multiplier::Integer=1)
    T = eltype(ℛ); out = HipVector{T}(nlimbs(T), degree(ℛ))
    check(ccall((:tfhe_sample_gaussian, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble, UInt64, UInt64, UInt32, UInt64, Ptr{UInt64}, Int64),
                hipring(ℛ).handle, out.limbs, σ, multiplier, rng.seed, 1, rng.next_poly, out.ptr, 1))
    rng.next_poly += 1
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end

# ---- multi-GPU: one Julia process per GPU (Distributed / MPI.jl), batch sharded by ciphertext, final gather -------------
set_device(dev::Integer) = check(ccall((:tfhe_set_device, lib), Cint, (Cint,), dev))
function device_count()
    n = Ref{Cint}(0); check(ccall((:tfhe_device_count, lib), Cint, (Ptr{Cint},), n)); Int(n[])
end

# rank 0 makes the RCCL rendezvous id (tfhe_comm_id) and hands it to the other ranks through the host-side transport
# (MPI.bcast, Distributed.remotecall ...); every rank then joins.  gather!: all-gather of equally sized per-rank shards.
mutable struct HipComm
    handle::Ptr{Cvoid}; nranks::Int; rank::Int
end
function comm_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:tfhe_comm_id, lib), Cint, (Ptr{Cvoid},), id)); id
end
function HipComm(id::Vector{UInt8}, nranks::Integer, rank::Integer)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:tfhe_comm_create, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}), id, nranks, rank, h))
    c = HipComm(h[], nranks, rank)
    finalizer(x -> ccall((:tfhe_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), c)
    c
end
function gather!(comm::HipComm, ring::HipRing, dst::HipVector, src::HipVector)
    check(ccall((:tfhe_gather, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Csize_t),
                comm.handle, ring.handle, src.ptr, dst.ptr, words(src)))
    dst
end

end # module
