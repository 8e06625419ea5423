# ToyFHEHIP.jl -- the Julia side of the drop-in: device storage for ToyFHE's NegacyclicRing /
# RingElement / CipherText, bound to libtoyfhe_hip.so with ccall.
#
# STATUS: written against include/toyfhe_hip.h but NEVER EXECUTED -- the build image has no Julia
# toolchain (SURVEY.md §8c).  The same call sequences are exercised from Python through ctypes
# (toyfhe.jl_amd/native.py, ring.py, she.py), which is the host mirror the tests run.  Validate on a
# machine with Julia >= 1.2 and the reference's Manifest before relying on it.
#
# How it plugs in: exactly like src/crt.jl:247-275 plugs the RNS NTT in -- by dispatch on the storage
# type parameter of RingElement{ℛ,Field,Storage} (src/pow2_cyc_rings.jl:93-96).  `HipVector` is that
# storage; the methods below override NTT.nntt / NTT.inntt, broadcast `+ - .*`, `modswitch`,
# `apply_galois_element`, `keyswitch` and BFV `enc_mul` for it.
module ToyFHEHIP

using ToyFHE
using ToyFHE: NTT, CRTEncoded, moduli, KeySwitchKey, CipherText, BFVParams, ModulusRaised
using ToyFHE.NTT: NegacyclicRing, RingElement, RingCoeffs, degree
using OffsetArrays, StructArrays

const lib = get(ENV, "TOYFHE_HIP_LIB", "libtoyfhe_hip.so")

struct UsageError <: Exception; msg::String; end
function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:tfhe_last_error, lib), Cstring, ()))
    rc == -1 && throw(AssertionError(msg))              # pow2_cyc_rings.jl:31,61,116; rlwe_she.jl:318
    rc in (-3, -4) && throw(ToyFHE.UsageError(msg))     # rlwe_she.jl:223-225,233-235,248-250
    rc == -7 && error(msg)                              # crt.jl:270,274
    rc == -5 && throw(OutOfMemoryError())
    error("HIP: " * msg)
end

# ---- ring context: one per NegacyclicRing{CRTEncoded{L,...},N} value --------------------------------
mutable struct HipRing
    handle::Ptr{Cvoid}; N::Int; q::Vector{UInt64}
end
const RINGS = IdDict{Any,HipRing}()
function hipring(ℛ::NegacyclicRing{T,N}) where {T<:CRTEncoded,N}
    get!(RINGS, ℛ) do
        q = UInt64[ToyFHE.NTT.modulus(F) for F in fieldtypes(moduli(T))]
        ψ = UInt64[convert(Integer, c) for c in ℛ.ψ.c]            # pow2_cyc_rings.jl:27-37
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:tfhe_ctx_create, lib), Cint, (Int64, Cint, Ptr{UInt64}, Ptr{UInt64}, Ptr{Ptr{Cvoid}}),
                    N, length(q), q, ψ, h))
        r = HipRing(h[], N, q)
        finalizer(r -> ccall((:tfhe_ctx_destroy, lib), Cint, (Ptr{Cvoid},), r.handle), r)
    end
end

# ---- device storage: [L][N] UInt64 residues, limb-major like StructArray field arrays (crt.jl:150-156)
mutable struct HipVector{T} <: AbstractVector{T}
    ptr::Ptr{UInt64}; limbs::Int; n::Int
    function HipVector{T}(limbs, n) where T
        p = Ref{Ptr{Cvoid}}()
        check(ccall((:tfhe_malloc, lib), Cint, (Csize_t, Ptr{Ptr{Cvoid}}), 8limbs * n, p))
        v = new{T}(convert(Ptr{UInt64}, p[]), limbs, n)
        finalizer(v -> ccall((:tfhe_free, lib), Cint, (Ptr{Cvoid},), v.ptr), v)
    end
end
Base.size(v::HipVector) = (v.n,)
function upload(sa::StructArray{T}) where {T<:CRTEncoded}
    cols = StructArrays.fieldarrays(sa); v = HipVector{T}(length(cols), length(sa))
    for (l, col) in enumerate(cols)
        host = UInt64[convert(Integer, x) for x in col]
        check(ccall((:tfhe_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.ptr + 8(l - 1) * v.n, host, 8v.n))
    end
    v
end
function download(v::HipVector{T}) where {T<:CRTEncoded}
    host = Matrix{UInt64}(undef, v.n, v.limbs)
    check(ccall((:tfhe_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), host, v.ptr, 8length(host)))
    StructArray{T}(tuple((map(F, host[:, l]) for (l, F) in enumerate(fieldtypes(moduli(T))))...))
end

# ---- K1/K2: the NTT hooks, same shape as crt.jl:247-267 ---------------------------------------------
for (f, sym) in ((:nntt, :tfhe_nntt), (:inntt, :tfhe_inntt))
    @eval function NTT.$f(rcs::RingCoeffs{ℛ,T,OffsetVector{T,S}})::RingCoeffs{ℛ} where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
        src = rcs.coeffs.parent; dst = HipVector{T}(src.limbs, src.n)
        check(ccall(($(QuoteNode(sym)), lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                    hipring(ℛ).handle, src.ptr, dst.ptr, 1, src.limbs, C_NULL))
        RingCoeffs{ℛ}(OffsetArray(dst, axes(rcs.coeffs)...))
    end
end

# ---- K3/K4: limb-wise broadcast (pow2_cyc_rings.jl:167,178-179,188-189,200-214) ---------------------
for (op, sym) in ((:+, :tfhe_add), (:-, :tfhe_sub), (:*, :tfhe_mul))
    @eval function Base.broadcasted(::typeof($op), a::OffsetVector{T,HipVector{T}}, b::OffsetVector{T,HipVector{T}}) where {T}
        dst = HipVector{T}(a.parent.limbs, a.parent.n)
        check(ccall(($(QuoteNode(sym)), lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                    CURRENT_RING[].handle, a.parent.ptr, b.parent.ptr, dst.ptr, 1, dst.limbs, C_NULL))
        OffsetArray(dst, axes(a)...)
    end
end
const CURRENT_RING = Ref{HipRing}()   # set by the RingElement-level wrappers below (ring is a type parameter upstream)

# ---- K6: modswitch(::RingElement) (crt.jl:226-228) ---------------------------------------------------
function ToyFHE.modswitch(re::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = NTT.coeffs_primal(re).parent; ℛ′ = ToyFHE.drop_last(ℛ); T′ = eltype(ℛ′)
    dst = HipVector{T′}(src.limbs - 1, src.n)
    check(ccall((:tfhe_rescale, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                hipring(ℛ).handle, src.ptr, dst.ptr, 1, src.limbs, C_NULL))
    RingElement{ℛ′}(OffsetArray(dst, 0:src.n-1), nothing)
end

# ---- K8: apply_galois_element (pow2_cyc_rings.jl:321-329) -------------------------------------------
function NTT.apply_galois_element(re::RingElement{ℛ,T,S}, g::Integer) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = NTT.coeffs_primal(re).parent; dst = HipVector{T}(src.limbs, src.n)
    check(ccall((:tfhe_galois, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, UInt64, Int64, Cint, Ptr{Int32}),
                hipring(ℛ).handle, src.ptr, dst.ptr, g, 1, src.limbs, C_NULL))
    RingElement{ℛ}(OffsetArray(dst, 0:src.n-1), nothing)
end

# ---- K9-K11: keyswitch (rlwe_she.jl:315-347) as one fused call --------------------------------------
# `pack(ek)` lays ek.key out as [digit][mask, masked][Lk][N] in the NTT domain (coeffs_dual), once per key.
function ToyFHE.keyswitch(ek::KeySwitchKey, c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) in (2, 3)                                                    # rlwe_she.jl:318
    keyring = NTT.ring(ek.key[1].mask); Lk = length(moduli(keyring).parameters); level = length(moduli(ℛ).parameters)
    ct = pack(c); out = HipVector{T}(2level, degree(ℛ))
    w = ToyFHE.relin_window(ek.params)
    if w != 0                                                                         # K14, rlwe_she.jl:330-338
        ek.params isa ModulusRaised && return invoke(ToyFHE.keyswitch, Tuple{KeySwitchKey,CipherText}, ek, c)
        check(ccall((:tfhe_keyswitch_window, lib), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Int64),
                    hipring(ℛ).handle, level, w, pack(ek).ptr, length(ek.key), ct.ptr, length(c.cs), out.ptr, 1))
        return CipherText{Enc}(c.params, unpack(out, ℛ, 2))
    end
    check(ccall((:tfhe_keyswitch, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Int64),
                hipring(keyring).handle, Lk, level, ek.params isa ModulusRaised, pack(ek).ptr, length(ek.key),
                ct.ptr, length(c.cs), out.ptr, 1))
    CipherText{Enc}(c.params, unpack(out, ℛ, 2))
end

# ---- CKKS encode / decode (ckksencoding.jl:56-97) on the device ---------------------------------------
# ScaleT = FixedRational{denom}: denom = mant * 2^exp2 (scale_parts as in toyfhe.jl_amd/she.py).
function Base.convert(::Type{<:RingElement{ℛ,T,S}}, s::CKKSEncoding{FixedRational{denom}}) where {ℛ,T,S<:HipVector{T},denom}
    mant, exp2 = scale_parts(denom); slots = upload(reinterpret(Float64, collect(s.data)))
    out = HipVector{T}(length(moduli(ℛ).parameters), degree(ℛ))
    check(ccall((:tfhe_ckks_encode, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, Cint, Ptr{Float64}, Ptr{UInt64}, Int64),
                hipring(ℛ).handle, out.limbs, mant, exp2, slots.ptr, out.ptr, 1))
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end
function ToyFHE.CKKSEncoding{FixedRational{denom}}(plain::RingElement{ℛ,T,S}) where {ℛ,T,S<:HipVector{T},denom}
    mant, exp2 = scale_parts(denom); src = NTT.coeffs_primal(plain).parent; slots = HipVector{Float64}(1, degree(ℛ))
    check(ccall((:tfhe_ckks_decode, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, Cint, Ptr{UInt64}, Ptr{Float64}, Int64),
                hipring(ℛ).handle, src.limbs, mant, exp2, src.ptr, slots.ptr, 1))
    CKKSEncoding{FixedRational{denom}}(typeof(plain), OffsetArray(reinterpret(ComplexF64, download(slots)), 0:degree(ℛ)÷2-1))
end

# ---- K12/K13: BFV enc_mul (rlwe_she.jl:247-262 + bfv.jl:34-40) --------------------------------------
# plan(params) = tfhe_bfv_plan_create(hipring(ℛ), idx, hipring(ℛbig), idx, t), cached per BFVParams.
function ToyFHE.enc_mul(c1::CipherText{E,BFVParams,<:RingElement{ℛ,T,<:HipVector}}, c2::CipherText{E,BFVParams}) where {E,ℛ,T}
    c1.params !== c2.params && throw(ToyFHE.UsageError("Attempting to multiply ciphertexts with differing parameters"))
    a, b = pack(c1), pack(c2); out = HipVector{T}(3length(moduli(ℛ).parameters), degree(ℛ))
    check(ccall((:tfhe_bfv_mul, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                plan(c1.params), a.ptr, b.ptr, out.ptr, 1))
    unpack(out, ℛ, 3)
end

# pack / unpack / plan: contiguous [polys][limbs][N] staging with tfhe_memcpy_d2d; omitted details are
# the same as `_pack` / `_unpack` / `BFVParams.plan` in toyfhe.jl_amd/she.py.
function pack end; function unpack end; function plan end

end # module
