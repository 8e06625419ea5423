# ToyFHEHIP.jl -- the Julia side of the drop-in: device storage for ToyFHE's NegacyclicRing /
# RingElement / CipherText, bound to libtoyfhe_hip.so with ccall.
#
# STATUS: complete against include/toyfhe_hip.h and statically checked (tests/test_julia_shim_cpu.py parses every
# ccall below and compares symbol, arity and C types with the header and with the ctypes table of
# toyfhe.jl_amd/native.py; it checks that every helper used here is defined here or imported, and that no call passes a
# literal 1 for a `count` / `batch` parameter).  It has NOT been executed: the build image has no Julia toolchain
# (SURVEY.md §8c).  The same call sequences are exercised from Python through ctypes (toyfhe.jl_amd/native.py, ring.py,
# she.py), which is the host mirror the tests run.  Validate on a machine with Julia >= 1.2 and the reference's Manifest
# before relying on it.
#
# How it plugs in: exactly like src/crt.jl:247-275 plugs the RNS NTT in -- by dispatch on the storage
# type parameter of RingElement{ℛ,Field,Storage} (src/pow2_cyc_rings.jl:93-96).  `HipVector` is that
# storage; the methods below override NTT.nntt / NTT.inntt, broadcast `+ - *`, `modswitch`, `modswitch_drop`,
# `apply_galois_element`, `keyswitch`, CKKS encode / decode, BFV `enc_mul` and `rand(::HipRng, ::RingSampler)` for it.
# Everything else in ToyFHE (keygen, encrypt, decrypt, π, π⁻¹, CipherText + -) is generic code over those and runs unchanged.
#
# BATCHES (north_star: "batches of independent ciphertexts").  A HipVector holds `count` polynomials, [count][limbs][N],
# and every entry point of the C ABI takes that count, so a RingElement over a batched HipVector is `count` ring elements
# that move through the reference's generic code (rlwe_she.jl:247-266 enc_mul, :315-349 keyswitch, :355-359 rotate) as one:
# `batch(cts)` stacks ciphertexts of one parameter set into a CipherText over batched elements, `unbatch` splits it again.
# One ciphertext is the batch of one.  (Element access -- getindex / setindex! -- is defined for count == 1 only.)
#
# STREAMS.  Every HipRing context owns a HIP stream.  The ring's own context (hipring(ℛ): NTTs, key switch, rescale, ...)
# and the (T, N)-keyed context of the limb-wise broadcast hooks (modring) are different contexts, so each HipVector remembers
# the context that last wrote it (`last`), and every operation orders its context after the producers of its operands on the
# device (tfhe_ctx_wait_for: one recorded event, no host wait) before it is enqueued.  `.ptr` values are passed to ccall
# under GC.@preserve: a finalizer cannot free a buffer before the call that uses it has been enqueued, and tfhe_free itself
# parks a block until the work enqueued so far has completed (include/toyfhe_hip.h: tfhe_free / tfhe_ctx_destroy /
# tfhe_bfv_plan_destroy may be called from any thread, e.g. the finalizer thread).
module ToyFHEHIP

using ToyFHE
using ToyFHE: NTT, CRTEncoded, moduli, KeySwitchKey, CipherText, BFVParams, ModulusRaised, CKKSEncoding, FixedRational, RingSampler
using ToyFHE.NTT: NegacyclicRing, RingElement, RingCoeffs, degree, coeffs_primal, coeffs_dual
using OffsetArrays, StructArrays, Random, Distributions

const lib = get(ENV, "TOYFHE_HIP_LIB", "libtoyfhe_hip.so")

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:tfhe_last_error, lib), Cstring, ()))
    rc == -1 && throw(AssertionError(msg))              # pow2_cyc_rings.jl:31,61,116; rlwe_she.jl:318
    rc in (-3, -4) && throw(ToyFHE.UsageError(msg))     # rlwe_she.jl:223-225,233-235,248-250
    rc == -7 && error(msg)                              # crt.jl:270,274
    if rc == -5                                         # TFHE_E_NOMEM: give the key caches back before the caller sees it (a retry may fit)
        release_key_caches!(all=isempty(PREPARED_KEYS)); throw(OutOfMemoryError())   # prepared copies first; the packed keys only when none were left
    end
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
# limb-wise operations (+ - * neg, scalar) depend on the moduli only, not on ψ: the broadcast hooks see the coefficient type
# T but not the ring value, so they use a context keyed on (T, N) whose ψ the library derives.  It is a context of its own
# (own stream): `on` below orders it against the ring contexts through the vectors' `last` fields.
const MODRINGS = Dict{Any,HipRing}()
function modring(::Type{T}, N::Integer) where {T<:CRTEncoded}
    get!(MODRINGS, (T, N)) do
        make_ring(N, limb_moduli(T), zeros(UInt64, nlimbs(T)))
    end
end
sync(r::HipRing) = check(ccall((:tfhe_ctx_sync, lib), Cint, (Ptr{Cvoid},), r.handle))
# device-side ordering between two contexts (no host wait): r's later work runs after what `producer` has been given so far
wait_for(r::HipRing, producer::HipRing) = check(ccall((:tfhe_ctx_wait_for, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), r.handle, producer.handle))

# ---- device storage: [count][L][N] UInt64 residues, limb-major like StructArray field arrays (crt.jl:150-156) --------
mutable struct HipVector{T} <: AbstractVector{T}
    ptr::Ptr{UInt64}; limbs::Int; n::Int; count::Int
    last::Union{Nothing,HipRing}                       # the context whose stream last wrote this buffer
    function HipVector{T}(limbs::Integer, n::Integer, count::Integer=1) where T
        p = Ref{Ptr{Cvoid}}()
        check(ccall((:tfhe_malloc, lib), Cint, (Csize_t, Ptr{Ptr{Cvoid}}), 8 * limbs * n * count, p))
        v = new{T}(convert(Ptr{UInt64}, p[]), limbs, n, count, nothing)
        finalizer(x -> ccall((:tfhe_free, lib), Cint, (Ptr{Cvoid},), x.ptr), v)   # any thread: the allocator parks the block
        v
    end
end
Base.size(v::HipVector) = (v.n,)
Base.similar(::Type{HipVector{T}}, ::Type{T′}) where {T,T′} = HipVector{T′}      # crt.jl:196-197
words(v::HipVector) = v.limbs * v.n                      # words of ONE polynomial
allwords(v::HipVector) = v.limbs * v.n * v.count
# `ctx` is about to read `ins` and write `outs`: order its stream after the producers of the inputs; mark the outputs
function on(ctx::HipRing, outs::Tuple, ins::Tuple)
    for v in ins
        v.last === nothing || v.last === ctx || wait_for(ctx, v.last)
    end
    for v in outs
        v.last = ctx
    end
    ctx
end
samecount(a::HipVector, b::HipVector) = (a.count == b.count || throw(ToyFHE.UsageError("operands hold different batch sizes")); a.count)

function upload(sa::StructArray{T}) where {T<:CRTEncoded}
    upload([sa])
end
# a batch of host polynomials -> one [count][limbs][n] device buffer
function upload(sas::Vector{<:StructArray{T}}) where {T<:CRTEncoded}
    cols1 = StructArrays.fieldarrays(sas[1]); v = HipVector{T}(length(cols1), length(sas[1]), length(sas))
    GC.@preserve v begin
        for (k, sa) in enumerate(sas), (l, col) in enumerate(StructArrays.fieldarrays(sa))
            host = UInt64[convert(Integer, x) for x in col]
            check(ccall((:tfhe_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                        v.ptr + 8 * ((k - 1) * v.limbs + (l - 1)) * v.n, host, 8 * v.n))
        end
    end
    v
end
function upload(host::Vector{Float64}, count::Integer=1)
    v = HipVector{Float64}(1, length(host) ÷ count, count)
    GC.@preserve v check(ccall((:tfhe_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), v.ptr, host, 8 * length(host)))
    v
end
# host copies (synchronous: tfhe_memcpy_d2h waits for the device)
function download_all(v::HipVector{T}) where {T<:CRTEncoded}
    host = Array{UInt64,3}(undef, v.n, v.limbs, v.count)
    v.last === nothing || sync(v.last)
    GC.@preserve v check(ccall((:tfhe_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), host, v.ptr, 8 * length(host)))
    [StructArray{T}(tuple((map(F, host[:, l, k]) for (l, F) in enumerate(fieldtypes(moduli(T))))...)) for k in 1:v.count]
end
download(v::HipVector{T}) where {T<:CRTEncoded} = (v.count == 1 || error("batched storage: use download_all"); download_all(v)[1])
function download(v::HipVector{Float64})
    host = Vector{Float64}(undef, v.n * v.count)
    v.last === nothing || sync(v.last)
    GC.@preserve v check(ccall((:tfhe_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), host, v.ptr, 8 * length(host)))
    host
end
# element access goes through the host (getindex / setindex! on RingElement, pow2_cyc_rings.jl:140-145); one polynomial only
Base.getindex(v::HipVector{T}, i::Int) where {T<:CRTEncoded} = download(v)[i]
function Base.setindex!(v::HipVector{T}, x, i::Int) where {T<:CRTEncoded}
    host = download(v); host[i] = convert(T, x); new = upload(host); ctx = on(modring(T, v.n), (v,), (v, new))
    GC.@preserve v new check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx.handle, v.ptr, new.ptr, 8 * allwords(v)))
    sync(ctx); x
end
function Base.zero(o::OffsetVector{T,HipVector{T}}) where {T<:CRTEncoded}
    v = HipVector{T}(o.parent.limbs, o.parent.n, o.parent.count); ctx = on(modring(T, v.n), (v,), ())
    GC.@preserve v check(ccall((:tfhe_memset, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Csize_t), ctx.handle, v.ptr, 0, 8 * allwords(v)))
    OffsetArray(v, axes(o)...)
end
# move host ring elements to the device / back: one element, or a batch of elements of one ring as ONE batched element
todevice(re::RingElement{ℛ,T}) where {ℛ,T<:CRTEncoded} =
    RingElement{ℛ}(OffsetArray(upload(coeffs_primal(re).parent), 0:degree(ℛ)-1), nothing)
todevice(res::Vector{<:RingElement{ℛ,T}}) where {ℛ,T<:CRTEncoded} =
    RingElement{ℛ}(OffsetArray(upload([coeffs_primal(re).parent for re in res]), 0:degree(ℛ)-1), nothing)
tohost(re::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T}} =
    [RingElement{ℛ}(OffsetArray(sa, 0:degree(ℛ)-1), nothing) for sa in download_all(coeffs_primal(re).parent)]
batchsize(re::RingElement{ℛ,T,S}) where {ℛ,T,S<:HipVector} = (re.primal === nothing ? re.dual : re.primal).parent.count
batchsize(c::CipherText) = batchsize(c.cs[1])

# ---- K1/K2: the NTT hooks, same shape as crt.jl:247-267 -------------------------------------------------------------
function NTT.nntt(rcs::RingCoeffs{ℛ,T,OffsetVector{T,S}})::RingCoeffs{ℛ} where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = rcs.coeffs.parent; dst = HipVector{T}(src.limbs, src.n, src.count); ctx = on(hipring(ℛ), (dst,), (src,))
    GC.@preserve src dst check(ccall((:tfhe_nntt, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, src.ptr, dst.ptr, src.count, src.limbs, C_NULL))
    RingCoeffs{ℛ}(OffsetArray(dst, axes(rcs.coeffs)...))
end
function NTT.inntt(rcs::RingCoeffs{ℛ,T,OffsetVector{T,S}})::RingCoeffs{ℛ} where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = rcs.coeffs.parent; dst = HipVector{T}(src.limbs, src.n, src.count); ctx = on(hipring(ℛ), (dst,), (src,))
    GC.@preserve src dst check(ccall((:tfhe_inntt, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, src.ptr, dst.ptr, src.count, src.limbs, C_NULL))
    RingCoeffs{ℛ}(OffsetArray(dst, axes(rcs.coeffs)...))
end

# ---- K3/K4: limb-wise broadcast (pow2_cyc_rings.jl:167,178-179,188-189,200-214) -------------------------------------
const DevVec{T} = OffsetVector{T,HipVector{T}}
# a single polynomial against a batch (the `Ref(c)` / scalar-broadcast patterns, rlwe_she.jl:143): repeat it on the device
function matched(a::HipVector{T}, b::HipVector{T}) where {T}
    a.count == b.count && return a, b
    a.count == 1 && return repeated(a, b.count), b
    b.count == 1 && return a, repeated(b, a.count)
    throw(ToyFHE.UsageError("operands hold different batch sizes"))
end
function repeated(a::HipVector{T}, count::Integer) where {T}
    dst = HipVector{T}(a.limbs, a.n, count); ctx = on(modring(T, a.n), (dst,), (a,))
    GC.@preserve a dst check(ccall((:tfhe_broadcast_poly, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Csize_t, Int64),
                ctx.handle, dst.ptr, a.ptr, words(a), count))
    dst
end
function Base.broadcasted(::typeof(+), a::DevVec{T}, b::DevVec{T}) where {T<:CRTEncoded}
    x, y = matched(a.parent, b.parent); dst = HipVector{T}(x.limbs, x.n, x.count); ctx = on(modring(T, dst.n), (dst,), (x, y))
    GC.@preserve x y dst check(ccall((:tfhe_add, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, x.ptr, y.ptr, dst.ptr, dst.count, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
function Base.broadcasted(::typeof(-), a::DevVec{T}, b::DevVec{T}) where {T<:CRTEncoded}
    x, y = matched(a.parent, b.parent); dst = HipVector{T}(x.limbs, x.n, x.count); ctx = on(modring(T, dst.n), (dst,), (x, y))
    GC.@preserve x y dst check(ccall((:tfhe_sub, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, x.ptr, y.ptr, dst.ptr, dst.count, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
function Base.broadcasted(::typeof(*), a::DevVec{T}, b::DevVec{T}) where {T<:CRTEncoded}
    x, y = matched(a.parent, b.parent); dst = HipVector{T}(x.limbs, x.n, x.count); ctx = on(modring(T, dst.n), (dst,), (x, y))
    GC.@preserve x y dst check(ccall((:tfhe_mul, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, x.ptr, y.ptr, dst.ptr, dst.count, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
# sum_k as[k] .* bs[k] in one device pass (tfhe_dot): the accumulation of the diagonal matrix-vector product, infer.jl:140-149
function dot(as::Vector{<:DevVec{T}}, bs::Vector{<:DevVec{T}}) where {T<:CRTEncoded}
    @assert length(as) == length(bs) && !isempty(as)
    xs = HipVector{T}[a.parent for a in as]; ys = HipVector{T}[b.parent for b in bs]
    cnt = samecount(xs[1], ys[1]); dst = HipVector{T}(xs[1].limbs, xs[1].n, cnt)
    ctx = on(modring(T, dst.n), (dst,), (xs..., ys...))
    ap = Ptr{UInt64}[x.ptr for x in xs]; bp = Ptr{UInt64}[y.ptr for y in ys]
    GC.@preserve xs ys dst check(ccall((:tfhe_dot, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{Ptr{UInt64}}, Ptr{Ptr{UInt64}}, Cint, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, C_NULL, ap, bp, length(as), dst.ptr, cnt, dst.limbs, C_NULL))
    OffsetArray(dst, axes(as[1])...)
end
function Base.broadcasted(::typeof(-), a::DevVec{T}) where {T<:CRTEncoded}
    x = a.parent; dst = HipVector{T}(x.limbs, x.n, x.count); ctx = on(modring(T, dst.n), (dst,), (x,))
    GC.@preserve x dst check(ccall((:tfhe_neg, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, x.ptr, dst.ptr, x.count, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
# scalar_mul (pow2_cyc_rings.jl:177-185): `scalar .* coeffs`
function Base.broadcasted(::typeof(*), s::Union{Integer,CRTEncoded}, a::DevVec{T}) where {T<:CRTEncoded}
    scal = UInt64[convert(Integer, c) for c in convert(T, s).c]
    x = a.parent; dst = HipVector{T}(x.limbs, x.n, x.count); ctx = on(modring(T, dst.n), (dst,), (x,))
    GC.@preserve x dst check(ccall((:tfhe_scalar_mul, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, scal, x.ptr, dst.ptr, x.count, dst.limbs, C_NULL))
    OffsetArray(dst, axes(a)...)
end
Base.broadcasted(::typeof(*), a::DevVec{T}, s::Union{Integer,CRTEncoded}) where {T<:CRTEncoded} = Base.broadcasted(*, s, a)

# [sum_k scalars[o][k] .* as[k] for o in outputs]: several scalar-weighted sums of the SAME operands in one pass over them
# (tfhe_lincomb_many) -- the output channels of a convolution with plaintext scalar weights, infer.jl:127-131 (each channel is
# `sum(C[i][j] * w[i,j,ch])`: per term one scalar_mul, pow2_cyc_rings.jl:177-185, and one +, :200-214).  One row of scalars per output.
function lincomb_many(scalars::Vector{<:Vector}, as::Vector{<:DevVec{T}}) where {T<:CRTEncoded}
    @assert !isempty(as) && !isempty(scalars)
    xs = HipVector{T}[a.parent for a in as]; nterms = length(xs); nout = length(scalars)
    for row in scalars
        length(row) == nterms || throw(AssertionError("lincomb_many: one scalar per operand in every row"))
    end
    cnt = xs[1].count; limbs = xs[1].limbs; n = xs[1].n
    for x in xs
        samecount(x, xs[1])
    end
    flat = UInt64[convert(Integer, c) for row in scalars for s in row for c in convert(T, s).c]      # [nout][nterms][limbs]
    dsts = HipVector{T}[HipVector{T}(limbs, n, cnt) for _ in 1:nout]; ctx = on(modring(T, n), (dsts...,), (xs...,))
    ap = Ptr{UInt64}[x.ptr for x in xs]; dp = Ptr{UInt64}[d.ptr for d in dsts]
    GC.@preserve xs dsts check(ccall((:tfhe_lincomb_many, lib), Cint,
                (Ptr{Cvoid}, Ptr{UInt64}, Ptr{Ptr{UInt64}}, Cint, Ptr{Ptr{UInt64}}, Cint, Int64, Cint, Ptr{Int32}),
                ctx.handle, flat, ap, nterms, dp, nout, cnt, limbs, C_NULL))
    [OffsetArray(d, axes(as[1])...) for d in dsts]
end
# sum_k scalars[k] .* as[k] in one pass (tfhe_lincomb; r06: bound directly -- it was reached through lincomb_many with one row):
# per term one scalar_mul (pow2_cyc_rings.jl:177-185) and one + (:200-214) in the reference, the same canonical residues here
function lincomb(scalars::Vector, as::Vector{<:DevVec{T}}) where {T<:CRTEncoded}
    @assert !isempty(as)
    xs = HipVector{T}[a.parent for a in as]; nterms = length(xs)
    length(scalars) == nterms || throw(AssertionError("lincomb: one scalar per operand"))
    cnt = xs[1].count; limbs = xs[1].limbs; n = xs[1].n
    for x in xs
        samecount(x, xs[1])
    end
    flat = UInt64[convert(Integer, c) for s in scalars for c in convert(T, s).c]                      # [nterms][limbs]
    dst = HipVector{T}(limbs, n, cnt); ctx = on(modring(T, n), (dst,), (xs...,))
    ap = Ptr{UInt64}[x.ptr for x in xs]
    GC.@preserve xs dst check(ccall((:tfhe_lincomb, lib), Cint,
                (Ptr{Cvoid}, Ptr{UInt64}, Ptr{Ptr{UInt64}}, Cint, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, flat, ap, nterms, dst.ptr, cnt, limbs, C_NULL))
    OffsetArray(dst, axes(as[1])...)
end

# ---- K6/K7: modswitch / modswitch_drop / crtselect (crt.jl:185-236) --------------------------------------------------
function ToyFHE.modswitch(re::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    src = coeffs_primal(re).parent; ℛ′ = ToyFHE.drop_last(ℛ); T′ = eltype(ℛ′)
    dst = HipVector{T′}(src.limbs - 1, src.n, src.count); ctx = on(hipring(ℛ), (dst,), (src,))
    GC.@preserve src dst check(ccall((:tfhe_rescale, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, src.ptr, dst.ptr, src.count, src.limbs, C_NULL))
    RingElement{ℛ′}(OffsetArray(dst, 0:src.n-1), nothing)
end
function select_limbs(ℛ, src::HipVector, ::Type{T′}, which) where {T′}
    idx = Int32[w - 1 for w in which]; dst = HipVector{T′}(length(idx), src.n, src.count); ctx = on(hipring(ℛ), (dst,), (src,))
    GC.@preserve src dst check(ccall((:tfhe_select_limbs, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}, Cint),
                ctx.handle, src.ptr, dst.ptr, src.count, src.limbs, idx, length(idx)))
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
    src = coeffs_primal(re).parent; dst = HipVector{T}(src.limbs, src.n, src.count); ctx = on(hipring(ℛ), (dst,), (src,))
    GC.@preserve src dst check(ccall((:tfhe_galois, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, UInt64, Int64, Cint, Ptr{Int32}),
                ctx.handle, src.ptr, dst.ptr, g, src.count, src.limbs, C_NULL))
    RingElement{ℛ}(OffsetArray(dst, 0:src.n-1), nothing)
end

# ---- ciphertext staging: the C ABI takes [batch][polys][limbs][N] contiguously -----------------------------------------
# pack: component p of every ciphertext of the batch with one strided copy each (tfhe_pack_poly); the parts are
# [count][limbs][N] buffers of the coefficient domain (or the NTT domain, for keys and the tensor)
function pack(ctx::HipRing, parts::Vector{<:HipVector}, ::Type{T}) where {T}
    limbs, n, cnt = parts[1].limbs, parts[1].n, parts[1].count; polys = length(parts)
    out = HipVector{T}(limbs * polys, n, cnt); on(ctx, (out,), (parts...,))
    GC.@preserve parts out begin
        for (k, p) in enumerate(parts)
            samecount(p, parts[1])
            check(ccall((:tfhe_pack_poly, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Cint, Cint, Csize_t, Int64),
                        ctx.handle, out.ptr, p.ptr, polys, k - 1, limbs * n, cnt))
        end
    end
    out
end
pack(ctx::HipRing, c::CipherText{Enc,P,<:RingElement{ℛ,T}}) where {Enc,P,ℛ,T} =
    pack(ctx, HipVector[coeffs_primal(x).parent for x in c.cs], T)
# evaluation key: [digit][mask, masked][Lk][N], NTT domain (rlwe_she.jl:297, 340-344), packed once per key (one polynomial each)
# Both key caches are guarded by one lock (rotate_many / matmul_diag may run from several tasks; get! on an IdDict is not
# task-safe) and can be given back: release_key_caches!() empties them (the HipVector finalizers tfhe_free the buffers), and
# `check` calls it once before reporting an out-of-memory error, so the copies are not invisible to the allocator's retry.
const KEY_LOCK = ReentrantLock()
const PACKED_KEYS = IdDict{Any,HipVector}()
function pack(ek::KeySwitchKey)
    lock(KEY_LOCK) do
        get!(PACKED_KEYS, ek) do
            ℛk = NTT.ring(ek.key[1].mask); parts = HipVector[]
            for kc in ek.key
                push!(parts, coeffs_dual(kc.mask).parent); push!(parts, coeffs_dual(kc.masked).parent)
            end
            pack(hipring(ℛk), parts, eltype(ℛk))
        end
    end
end
# Galois keys PREPARED for the hoisted rotations (tfhe_rotate_many with prepared = 1, tfhe_matmul_diag): every NTT-domain row
# permuted by g^-1 (tfhe_galois_key_prepare), once per key -- what GaloisKey.prepared() is in the Python mirror.
# The prepared copy is a SECOND full device copy of the key (2.8 GB more for 63 rotations at N = 2^16): the cache is bounded
# (TOYFHE_HIP_PREPARED_GIB, default 8) and evicts in insertion order; an evicted key is prepared again from the packed one on its
# next use (one permutation pass).  Callers hold the returned vectors for the duration of their ccall, so eviction never frees
# a buffer in use.
const PREPARED_KEYS = IdDict{Any,HipVector}()
const PREPARED_ORDER = Any[]
const PREPARED_BYTES = Ref{Int}(0)
prepared_limit() = round(Int, parse(Float64, get(ENV, "TOYFHE_HIP_PREPARED_GIB", "8")) * 2.0^30)
function prepared(gk::ToyFHE.GaloisKey)
    lock(KEY_LOCK) do
        haskey(PREPARED_KEYS, gk) && return PREPARED_KEYS[gk]
        ek = gk.key; key = pack(ek); keyring = NTT.ring(ek.key[1].mask); ctx = hipring(keyring)
        out = HipVector{eltype(keyring)}(key.limbs, key.n, key.count); on(ctx, (out,), (key,))
        GC.@preserve key out check(ccall((:tfhe_galois_key_prepare, lib), Cint,
                    (Ptr{Cvoid}, Cint, Cint, UInt64, Ptr{UInt64}, Ptr{UInt64}),
                    ctx.handle, nlimbs(eltype(keyring)), length(ek.key), gk.galois_element, key.ptr, out.ptr))
        PREPARED_KEYS[gk] = out; push!(PREPARED_ORDER, gk); PREPARED_BYTES[] += 8 * allwords(out)
        while PREPARED_BYTES[] > prepared_limit() && length(PREPARED_ORDER) > 1
            old = popfirst!(PREPARED_ORDER); v = pop!(PREPARED_KEYS, old); PREPARED_BYTES[] -= 8 * allwords(v)
        end
        out
    end
end
# out-of-memory recovery (check): the PREPARED copies go first (they are re-made from the packed keys by one permutation pass);
# `all = true` drops the packed keys too.  The dropped vectors are long-lived, old-generation objects: an incremental collection
# would not reach them, so a FULL collection runs their finalizers (tfhe_free: parked, then reusable) -- and only theirs if no
# other task still holds one for a call it is about to make (no explicit finalize: that would free under such a caller) (r06,
# ADVICE r05).
function release_key_caches!(; all::Bool=false)
    lock(KEY_LOCK) do
        empty!(PREPARED_KEYS); empty!(PREPARED_ORDER); PREPARED_BYTES[] = 0
        all && empty!(PACKED_KEYS)
    end
    GC.gc(true)
    ccall((:tfhe_alloc_trim, lib), Cint, ())
    nothing
end
# unpack: `polys` (batched) ring elements of ℛ from a packed [count][polys][limbs][N] buffer; dual = true: NTT-domain results
function unpack(ctx::HipRing, buf::HipVector, ℛ, polys::Integer; dual::Bool=false)
    T = eltype(ℛ); limbs = nlimbs(T); n = degree(ℛ); cnt = buf.count
    els = map(1:polys) do k
        v = HipVector{T}(limbs, n, cnt); on(ctx, (v,), (buf,))
        GC.@preserve v buf check(ccall((:tfhe_unpack_poly, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Cint, Cint, Csize_t, Int64),
                    ctx.handle, v.ptr, buf.ptr, polys, k - 1, limbs * n, cnt))
        dual ? RingElement{ℛ}(nothing, OffsetArray(v, 0:n-1)) : RingElement{ℛ}(OffsetArray(v, 0:n-1), nothing)
    end
    tuple(els...)                                    # `buf` may be finalised now: tfhe_free parks it behind the copies
end
# ciphertexts of one parameter set as ONE ciphertext over batched ring elements, and back
function batch(cts::Vector{<:CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}}) where {Enc,P,ℛ,T}
    polys = length(cts[1].cs); ctx = hipring(ℛ)
    els = map(1:polys) do k
        parts = HipVector[coeffs_primal(c.cs[k]).parent for c in cts]
        limbs, n = parts[1].limbs, parts[1].n; v = HipVector{T}(limbs, n, length(cts)); on(ctx, (v,), (parts...,))
        GC.@preserve parts v begin
            for (b, p) in enumerate(parts)
                check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                            ctx.handle, v.ptr + 8 * (b - 1) * limbs * n, p.ptr, 8 * limbs * n))
            end
        end
        RingElement{ℛ}(OffsetArray(v, 0:n-1), nothing)
    end
    CipherText{Enc}(cts[1].params, tuple(els...))
end
function unbatch(c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    ctx = hipring(ℛ); cnt = batchsize(c)
    map(1:cnt) do b
        els = map(c.cs) do x
            src = coeffs_primal(x).parent; v = HipVector{T}(src.limbs, src.n); on(ctx, (v,), (src,))
            GC.@preserve src v check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                        ctx.handle, v.ptr, src.ptr + 8 * (b - 1) * words(src), 8 * words(src)))
            RingElement{ℛ}(OffsetArray(v, 0:src.n-1), nothing)
        end
        CipherText{Enc}(c.params, tuple(els...))
    end
end

# ---- K9-K11 / K14: keyswitch (rlwe_she.jl:315-347) as one fused call ---------------------------------------------------
function ToyFHE.keyswitch(ek::KeySwitchKey, c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) in (2, 3)                                                    # rlwe_she.jl:318
    keyring = NTT.ring(ek.key[1].mask); Lk = nlimbs(eltype(keyring)); level = nlimbs(T); cnt = batchsize(c)
    w = ToyFHE.relin_window(ek.params); key = pack(ek)
    ctx = hipring(keyring)                              # the key ring's context takes the ciphertext ring's limbs as a prefix
    ct = pack(ctx, c); out = HipVector{T}(2 * level, degree(ℛ), cnt); on(ctx, (out,), (ct, key))
    if w != 0                                                                         # K14, rlwe_she.jl:330-338 (+ modulusraising.jl:35-49)
        GC.@preserve key ct out check(ccall((:tfhe_keyswitch_window, lib), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Int64),
                    ctx.handle, Lk, level, ek.params isa ModulusRaised ? 1 : 0, w, key.ptr, length(ek.key), ct.ptr, length(c.cs), out.ptr, cnt))
        return CipherText{Enc}(c.params, unpack(ctx, out, ℛ, 2))
    end
    GC.@preserve key ct out check(ccall((:tfhe_keyswitch, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Int64),
                ctx.handle, Lk, level, ek.params isa ModulusRaised ? 1 : 0, key.ptr, length(ek.key),
                ct.ptr, length(c.cs), out.ptr, cnt))
    CipherText{Enc}(c.params, unpack(ctx, out, ℛ, 2))
end
# rotate(gk, c) = keyswitch(gk, apply_galois_element(c, g)) (rlwe_she.jl:355-359), fused on the device
function ToyFHE.rotate(gk::ToyFHE.GaloisKey, c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) == 2
    ek = gk.key; ToyFHE.relin_window(ek.params) != 0 && return ToyFHE.keyswitch(ek, ToyFHE.NTT.apply_galois_element(c, gk.galois_element))
    keyring = NTT.ring(ek.key[1].mask); Lk = nlimbs(eltype(keyring)); level = nlimbs(T); cnt = batchsize(c)
    ctx = hipring(keyring); ct = pack(ctx, c); out = HipVector{T}(2 * level, degree(ℛ), cnt)
    if degree(ℛ) >= 2^15 && cnt >= 8
        # the rotation is finished in the key switch's tail on the key as tfhe_galois_key_prepare leaves it: prepared once per key
        # (PREPARED_KEYS), not once per call -- infer.jl:140-149 rotates by ONE key 63 times per matrix product (r06)
        key = prepared(gk); on(ctx, (out,), (ct, key))
        GC.@preserve key ct out check(ccall((:tfhe_rotate_prepared, lib), Cint,
                    (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt64}, Cint, UInt64, Ptr{UInt64}, Ptr{UInt64}, Int64),
                    ctx.handle, Lk, level, ek.params isa ModulusRaised ? 1 : 0, key.ptr, length(ek.key),
                    gk.galois_element, ct.ptr, out.ptr, cnt))
        return CipherText{Enc}(c.params, unpack(ctx, out, ℛ, 2))
    end
    key = pack(ek); on(ctx, (out,), (ct, key))
    GC.@preserve key ct out check(ccall((:tfhe_rotate, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt64}, Cint, UInt64, Ptr{UInt64}, Ptr{UInt64}, Int64),
                ctx.handle, Lk, level, ek.params isa ModulusRaised ? 1 : 0, key.ptr, length(ek.key),
                gk.galois_element, ct.ptr, out.ptr, cnt))
    CipherText{Enc}(c.params, unpack(ctx, out, ℛ, 2))
end

# hoisted rotations: [rotate(gk, c) for gk in gks] from one digit decomposition of c (tfhe_rotate_many)
function rotate_many(gks::Vector{<:ToyFHE.GaloisKey}, c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) == 2
    ek1 = gks[1].key; keyring = NTT.ring(ek1.key[1].mask); Lk = nlimbs(eltype(keyring)); level = nlimbs(T); cnt = batchsize(c)
    ctx = hipring(keyring); ct = pack(ctx, c); n = degree(ℛ); nrot = length(gks)
    packed = HipVector[prepared(gk) for gk in gks]; out = HipVector{T}(2 * level, n, cnt * nrot); on(ctx, (out,), (ct, packed...))
    keys = Ptr{UInt64}[k.ptr for k in packed]; gs = UInt64[gk.galois_element for gk in gks]
    GC.@preserve packed ct out check(ccall((:tfhe_rotate_many, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{UInt64}}, Cint, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Ptr{UInt64}, Int64),
                ctx.handle, Lk, level, ek1.params isa ModulusRaised ? 1 : 0, keys, length(ek1.key), 1, gs, nrot,
                ct.ptr, out.ptr, cnt))
    map(1:nrot) do r                                             # out: [nrot][cnt][2][level][N]
        part = HipVector{T}(2 * level, n, cnt); on(ctx, (part,), (out,))
        GC.@preserve part out check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                    ctx.handle, part.ptr, out.ptr + 8 * (r - 1) * cnt * 2 * level * n, 8 * cnt * 2 * level * n))
        CipherText{Enc}(c.params, unpack(ctx, part, ℛ, 2))
    end
end

# diags[1] .* c + sum_k diags[k+1] .* rotate(gks[k], c) -- the diagonal matrix-vector product of infer.jl:140-149 /
# test/ckks_matmul.jl:33-41 -- in one device call (tfhe_matmul_diag): hoisted rotations, with a special prime finished in the
# evaluation domain, accumulation included; the result is in the NTT domain like that of `dot`.  diags: plaintext elements of c's
# ring (one polynomial each, shared by the batch).  Same words as rotate_many -> coeffs_dual -> dot.
function matmul_diag(gks::Vector{<:ToyFHE.GaloisKey}, diags::Vector{<:RingElement{ℛ,T,<:HipVector}},
                     c::CipherText{Enc,P,<:RingElement{ℛ,T,<:HipVector}}) where {Enc,P,ℛ,T}
    @assert length(c.cs) == 2 && !isempty(gks) && length(diags) == length(gks) + 1
    length(gks) <= 64 || throw(ToyFHE.UsageError("matmul_diag: at most 64 rotations per call (TFHE_DOT_MAX); split the product and add the parts"))
    ek1 = gks[1].key; keyring = NTT.ring(ek1.key[1].mask); Lk = nlimbs(eltype(keyring)); level = nlimbs(T); cnt = batchsize(c)
    ctx = hipring(keyring); ct = pack(ctx, c); n = degree(ℛ); nrot = length(gks)
    packed = HipVector[prepared(gk) for gk in gks]                # tfhe_matmul_diag takes PREPARED keys only (tfhe_galois_key_prepare)
    dparts = HipVector[coeffs_dual(d).parent for d in diags]
    all(d -> d.count == 1, dparts) || throw(ToyFHE.UsageError("matmul_diag: one polynomial per diagonal"))
    dg = HipVector{T}(level, n, nrot + 1); on(ctx, (dg,), (dparts...,))
    GC.@preserve dparts dg for (k, d) in enumerate(dparts)                    # dg: [nrot + 1][level][N]
        check(ccall((:tfhe_memcpy_d2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
              ctx.handle, dg.ptr + 8 * (k - 1) * level * n, d.ptr, 8 * level * n))
    end
    out = HipVector{T}(2 * level, n, cnt); on(ctx, (out,), (ct, dg, packed...))
    keys = Ptr{UInt64}[k.ptr for k in packed]; gs = UInt64[gk.galois_element for gk in gks]
    GC.@preserve packed ct dg out check(ccall((:tfhe_matmul_diag, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{UInt64}}, Cint, Ptr{UInt64}, Cint, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                ctx.handle, Lk, level, ek1.params isa ModulusRaised ? 1 : 0, keys, length(ek1.key), gs, nrot,
                dg.ptr, ct.ptr, out.ptr, cnt))
    # a ciphertext-by-plaintext product: the result sits at the SQUARED scale, as the reference's `.*` returns it
    # (ckksencoding.jl:106-111: CipherText{CKKSEncoding{Tscale^2}}); other encodings carry no scale
    CipherText{squared_encoding(Enc)}(c.params, unpack(ctx, out, ℛ, 2; dual=true))
end
squared_encoding(::Type{CKKSEncoding{Tscale}}) where {Tscale} = CKKSEncoding{Tscale^2}
squared_encoding(::Type{E}) where {E} = E

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
# `slots`: N/2 complex values per polynomial, `count` polynomials back to back -> one batched plaintext element
function encode(::Type{<:RingElement{ℛ,T,S}}, slots::Vector{ComplexF64}, denom, count::Integer=1) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    mant, exp2 = scale_parts(denom); sl = upload(collect(reinterpret(Float64, slots)), count)
    out = HipVector{T}(nlimbs(T), degree(ℛ), count); ctx = on(hipring(ℛ), (out,), (sl,))
    GC.@preserve sl out check(ccall((:tfhe_ckks_encode, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, Cint, Ptr{Float64}, Ptr{UInt64}, Int64),
                ctx.handle, out.limbs, mant, exp2, sl.ptr, out.ptr, count))
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end
Base.convert(R::Type{<:RingElement{ℛ,T,S}}, s::CKKSEncoding{FixedRational{denom}}) where {ℛ,T<:CRTEncoded,S<:HipVector{T},denom} =
    encode(R, collect(s.data), denom)
function decode(plain::RingElement{ℛ,T,S}, denom) where {ℛ,T<:CRTEncoded,S<:HipVector{T}}
    mant, exp2 = scale_parts(denom); src = coeffs_primal(plain).parent; sl = HipVector{Float64}(1, degree(ℛ), src.count)
    ctx = on(hipring(ℛ), (sl,), (src,))
    GC.@preserve src sl check(ccall((:tfhe_ckks_decode, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, Cint, Ptr{UInt64}, Ptr{Float64}, Int64),
                ctx.handle, src.limbs, mant, exp2, src.ptr, sl.ptr, src.count))
    collect(reinterpret(ComplexF64, download(sl)))                # [count][N/2] slots, polynomial-major
end
function ToyFHE.CKKSEncoding{FixedRational{denom}}(plain::RingElement{ℛ,T,S}) where {ℛ,T<:CRTEncoded,S<:HipVector{T},denom}
    batchsize(plain) == 1 || error("batched plaintext: use decode(plain, denom)")
    CKKSEncoding{FixedRational{denom}}(typeof(plain), OffsetArray(decode(plain, denom), 0:degree(ℛ)÷2-1))
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
    # any other component counts: the reference's generic convolution (rlwe_she.jl:247-262) over the device hooks below
    (length(c1.cs) == 2 && length(c2.cs) == 2) || return invoke(ToyFHE.enc_mul, Tuple{Any,Any}, c1, c2)
    ctx = hipring(ℛ); a, b = pack(ctx, c1), pack(ctx, c2); cnt = samecount(a, b)   # the plan's results are ordered on ℛ's stream
    out = HipVector{T}(3 * nlimbs(T), degree(ℛ), cnt); on(ctx, (out,), (a, b))
    GC.@preserve a b out check(ccall((:tfhe_bfv_mul, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                plan(c1.params).handle, a.ptr, b.ptr, out.ptr, cnt))
    unpack(ctx, out, ℛ, 3)
end
# mul_expand / mul_contract (bfv.jl:34-40) for device storage: switch(ℛbig, c) and switch(ℛ, multround(e, t, q)) per component
# (tfhe_bfv_expand / tfhe_bfv_contract: exact, bit-identical to the BigInt path); the ring products in between are ℛbig's own
# device NTTs.  The conversions order ℛ's and ℛbig's streams themselves.
function bfv_expand(params::BFVParams, x::RingElement{ℛ,T,<:HipVector}) where {ℛ,T}
    ℛb = params.ℛbig; Tb = eltype(ℛb); src = coeffs_primal(x).parent
    out = HipVector{Tb}(nlimbs(Tb), degree(ℛb), src.count); on(hipring(ℛb), (out,), (src,))
    GC.@preserve src out check(ccall((:tfhe_bfv_expand, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                plan(params).handle, src.ptr, out.ptr, src.count))
    RingElement{ℛb}(OffsetArray(out, 0:degree(ℛb)-1), nothing)
end
function bfv_contract(params::BFVParams, e::RingElement{ℛb,Tb,<:HipVector}) where {ℛb,Tb}
    ℛ = params.ℛ; T = eltype(ℛ); src = coeffs_primal(e).parent
    out = HipVector{T}(nlimbs(T), degree(ℛ), src.count); on(hipring(ℛ), (out,), (src,))
    GC.@preserve src out check(ccall((:tfhe_bfv_contract, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                plan(params).handle, src.ptr, out.ptr, src.count))
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end
ToyFHE.mul_expand(params::BFVParams, c::CipherText{E,BFVParams,<:RingElement{ℛ,T,<:HipVector}}) where {E,ℛ,T} = map(x -> bfv_expand(params, x), c.cs)
function ToyFHE.mul_contract(params::BFVParams, c::Vector{<:RingElement{ℛb,Tb,<:HipVector}}) where {ℛb,Tb}
    map(e -> bfv_contract(params, e), c)
end
# c1*c2 followed by keyswitch(ek, .) in one call (the BASELINE.json unit) for RNS-gadget keys on ℛ itself
function mul_relin(ek::KeySwitchKey, c1::CipherText{E,BFVParams,<:RingElement{ℛ,T,<:HipVector}}, c2::CipherText{E,BFVParams}) where {E,ℛ,T}
    c1.params !== c2.params && throw(ToyFHE.UsageError("Attempting to multiply ciphertexts with differing parameters"))
    ctx = hipring(ℛ); key = pack(ek); a, b = pack(ctx, c1), pack(ctx, c2); cnt = samecount(a, b)
    out = HipVector{T}(2 * nlimbs(T), degree(ℛ), cnt); on(ctx, (out,), (a, b, key))
    GC.@preserve key a b out check(ccall((:tfhe_bfv_mul_relin, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Cint, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64),
                plan(c1.params).handle, key.ptr, length(ek.key), a.ptr, b.ptr, out.ptr, cnt))
    CipherText{E}(c1.params, unpack(ctx, out, ℛ, 2))
end

# ---- tensor for schemes whose mul_expand / mul_contract are the identity (BGV, CKKS; rlwe_she.jl:39-40,255-258) -----------
function ToyFHE.enc_mul(c1::CipherText{E,P,<:RingElement{ℛ,T,<:HipVector}}, c2::CipherText{E,P}) where {E,P,ℛ,T}
    c1.params !== c2.params && throw(ToyFHE.UsageError("Attempting to multiply ciphertexts with differing parameters"))
    (length(c1.cs) == 2 && length(c2.cs) == 2) || return invoke(ToyFHE.enc_mul, Tuple{CipherText,CipherText}, c1, c2)
    ctx = hipring(ℛ)
    a = pack(ctx, HipVector[coeffs_dual(x).parent for x in c1.cs], T); b = pack(ctx, HipVector[coeffs_dual(x).parent for x in c2.cs], T)
    cnt = samecount(a, b); out = HipVector{T}(3 * nlimbs(T), degree(ℛ), cnt); on(ctx, (out,), (a, b))
    GC.@preserve a b out check(ccall((:tfhe_tensor, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Int64, Cint, Ptr{Int32}),
                ctx.handle, a.ptr, b.ptr, out.ptr, cnt, nlimbs(T), C_NULL))
    unpack(ctx, out, ℛ, 3; dual=true)                          # NTT-domain results: dual-only ring elements
end

# ---- device samplers behind RingSampler (poly.jl:7-23, crt.jl:277-279) -----------------------------------------------------
# A counter-based generator (Philox4x32-10 on the device, keyed by (seed, stream, polynomial index)): an AbstractRNG, so that
# `rand(rng, 𝒰)` / `rand(rng, 𝒩(params))` in keygen / encrypt (rlwe_she.jl:155-217) dispatch to the methods below when the
# sampler's ring is device-backed.  `count` polynomials per draw (one batched element).  The stream is the library's own, not
# Julia's MersenneTwister: RNG parity with the reference is not a goal (SURVEY.md §7).
mutable struct HipRng <: Random.AbstractRNG
    seed::UInt64; next_poly::UInt64; count::Int
end
HipRng(seed::Integer) = HipRng(UInt64(seed), UInt64(0), 1)
function sample_uniform(rng::HipRng, ℛ)
    T = eltype(ℛ); out = HipVector{T}(nlimbs(T), degree(ℛ), rng.count); ctx = on(hipring(ℛ), (out,), ())
    GC.@preserve out check(ccall((:tfhe_sample_uniform, lib), Cint, (Ptr{Cvoid}, Cint, UInt64, UInt32, UInt64, Ptr{UInt64}, Int64),
                ctx.handle, out.limbs, rng.seed, 0, rng.next_poly, out.ptr, rng.count))
    rng.next_poly += rng.count
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end
function sample_gaussian(rng::HipRng, ℛ, σ::Real, multiplier::Integer=1)
    T = eltype(ℛ); out = HipVector{T}(nlimbs(T), degree(ℛ), rng.count); ctx = on(hipring(ℛ), (out,), ())
    GC.@preserve out check(ccall((:tfhe_sample_gaussian, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble, UInt64, UInt64, UInt32, UInt64, Ptr{UInt64}, Int64),
                ctx.handle, out.limbs, σ, multiplier, rng.seed, 1, rng.next_poly, out.ptr, rng.count))
    rng.next_poly += rng.count
    RingElement{ℛ}(OffsetArray(out, 0:degree(ℛ)-1), nothing)
end
# the seam: Random.rand(rng, ::RingSampler) (poly.jl:18-23) with the coefficient distribution the schemes use --
#   DiscreteUniform(coefftype(ℛ)) = the coefficient type itself (utils.jl:32; rlwe_she.jl:156,278)    -> uniform residues
#   DiscreteNormal(0, σ) (bfv.jl:31-32, bgv.jl:34, ckks.jl:24-25)                                    -> rounded Gaussian
#   ShiftedDiscreteNormal(p, DiscreteNormal(0, σ)) (bgv.jl:27-33)                                    -> p times a rounded Gaussian
# (Distributions.params(d) = (μ, σ) for the fork's DiscreteNormal, Manifest.toml:182-188: assumed, unexecuted.)
function Random.rand(rng::HipRng, r::RingSampler{ℛ}) where {ℛ<:NegacyclicRing{<:CRTEncoded}}
    d = r.coeff_distribution
    d isa Type && return sample_uniform(rng, NTT.ring(r))
    d isa ToyFHE.ShiftedDiscreteNormal && return sample_gaussian(rng, NTT.ring(r), Distributions.params(d.dn)[2], d.p)
    sample_gaussian(rng, NTT.ring(r), Distributions.params(d)[2])
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
    on(ring, (dst,), (src,))
    GC.@preserve dst src check(ccall((:tfhe_gather, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Csize_t),
                comm.handle, ring.handle, src.ptr, dst.ptr, allwords(src)))
    dst
end

end # module
