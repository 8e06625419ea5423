"""Static check of the Julia shim (toyfhe.jl_amd/julia/ToyFHEHIP.jl): the image has no Julia, so the shim cannot be
executed here -- but every `ccall((:sym, lib), Ret, (argtypes...), args...)` in it is parsed and compared, symbol by
symbol, with include/toyfhe_hip.h (arity + C type class of every argument + return type) and with the ctypes table the
Python mirror actually runs through (toyfhe.jl_amd/native.py), and every function the shim calls must be defined in the
file, imported by it, module-qualified, or a Julia Base name.  This keeps the shim from rotting into pseudo-code."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "toyfhe.jl_amd", "julia", "ToyFHEHIP.jl")
HEADER = os.path.join(ROOT, "include", "toyfhe_hip.h")


def c_class(t):
    t = t.strip()
    if "*" in t:
        return "ptr"
    t = re.sub(r"\b(const|unsigned)\b", "", t).strip()
    base = t.split()[0]
    return {"int": "int", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "size_t": "size", "double": "f64",
            "int32_t": "i32", "void": "void"}[base]


def header_prototypes():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char \*)\s*(tfhe_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argl = [] if args in ("", "void") else [c_class(a) for a in args.split(",")]
        protos[name] = ("cstr" if "char" in ret else "int", argl)
    return protos


def jl_class(t):
    t = t.strip()
    if t.startswith("Ptr{") or t in ("Ptr",):
        return "ptr"
    return {"Cint": "int", "Int64": "i64", "UInt64": "u64", "UInt32": "u32", "Csize_t": "size", "Cdouble": "f64",
            "Float64": "f64", "Int32": "i32", "Cstring": "cstr"}[t]


def split_top(s):
    """split at top-level commas (parentheses / braces / brackets balanced)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def shim_ccalls():
    src = open(SHIM).read()
    src = "\n".join(l.split("#")[0] if not l.lstrip().startswith("#") else "" for l in src.splitlines())
    calls = []
    for m in re.finditer(r"ccall\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        parts = split_top(src[m.end():i - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*lib\s*\)", parts[0])
        assert sym, f"ccall with a computed symbol: {parts[0]}"
        assert parts[2].startswith("(") and parts[2].endswith(")")
        inner = parts[2][1:-1].strip().rstrip(",")
        argtypes = [jl_class(t) for t in split_top(inner)] if inner else []
        calls.append((sym.group(1), jl_class(parts[1]), argtypes, len(parts) - 3))
    return calls


def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    assert len(protos) >= 48
    calls = shim_ccalls()
    assert len(calls) >= 30
    for name, ret, argtypes, nargs in calls:
        assert name in protos, f"{name} is not declared in include/toyfhe_hip.h"
        hret, hargs = protos[name]
        assert ret == hret, (name, ret, hret)
        assert nargs == len(argtypes), f"{name}: {nargs} arguments passed for {len(argtypes)} declared types"
        assert argtypes == hargs, f"{name}: shim {argtypes} vs header {hargs}"


def test_shim_binds_the_whole_hot_path():
    bound = {c[0] for c in shim_ccalls()}
    need = {"tfhe_ctx_create", "tfhe_ctx_destroy", "tfhe_malloc", "tfhe_free", "tfhe_memcpy_h2d", "tfhe_memcpy_d2h", "tfhe_memcpy_d2d",
            "tfhe_nntt", "tfhe_inntt", "tfhe_add", "tfhe_sub", "tfhe_neg", "tfhe_mul", "tfhe_scalar_mul", "tfhe_tensor", "tfhe_rescale",
            "tfhe_select_limbs", "tfhe_galois", "tfhe_keyswitch", "tfhe_rotate", "tfhe_keyswitch_window", "tfhe_ckks_encode",
            "tfhe_ckks_decode", "tfhe_bfv_plan_create", "tfhe_bfv_plan_destroy", "tfhe_bfv_mul", "tfhe_bfv_mul_relin",
            "tfhe_sample_uniform", "tfhe_sample_gaussian", "tfhe_last_error", "tfhe_ctx_sync", "tfhe_set_device",
            # r06: the weighted sums bound directly, rotations on prepared keys
            "tfhe_lincomb", "tfhe_lincomb_many", "tfhe_rotate_prepared", "tfhe_rotate_many", "tfhe_matmul_diag", "tfhe_galois_key_prepare"}
    assert need <= bound, sorted(need - bound)


def test_ccalls_agree_with_the_ctypes_table_the_tests_run_through():
    import ctypes as C

    import toyfhe_jl_amd as tf
    lib = tf.native.lib()

    def ct_class(t):
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) not in (None, "i", "l", "L", "I", "d", "Q", "q"):
            return "ptr"
        return {C.c_int: "int", C.c_int64: "i64", C.c_uint64: "u64", C.c_uint32: "u32", C.c_size_t: "size", C.c_double: "f64",
                C.c_int32: "int"}[t]

    same_width = {"size": "u64"}                       # c_size_t is c_ulong is c_uint64 on this ABI
    for name, ret, argtypes, _ in shim_ccalls():
        if name == "tfhe_last_error":
            continue
        got = [ct_class(t) for t in getattr(lib, name).argtypes]
        norm = lambda xs: [same_width.get(x, x) for x in xs]
        assert norm(got) == norm(argtypes), (name, got, argtypes)


JULIA_BASE = set("""
get get! haskey all isempty push! map tuple length size axes zeros zero collect reinterpret convert error throw finalizer unsafe_string
enumerate fieldtypes round ispow2 trailing_zeros ndigits numerator denominator big invoke typeof eltype isa new similar string
Ref Dict IdDict Vector Matrix UInt64 UInt32 Int32 Int64 Cint Float64 ComplexF64 Rational BigInt AssertionError OutOfMemoryError
OffsetArray StructArray Ptr in ccall tuple first last min max setindex! getindex sel x HipVector Int
ReentrantLock lock empty! parse pop! popfirst!
""".split())


def test_every_called_name_is_defined_imported_or_base():
    src = open(SHIM).read()
    code = "\n".join(l.split("#")[0] if not l.lstrip().startswith("#") else "" for l in src.splitlines())
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', code)
    defined = set(re.findall(r"\bfunction\s+(?:[\w.]+\.)?([\w!]+)", code))
    defined |= set(re.findall(r"^\s*(?:[\w.]+\.)?([\w!]+)\([^=\n]*\)(?:\s+where\s+[^=\n]+)?\s*=(?!=)", code, flags=re.M))
    defined |= set(re.findall(r"\b(?:mutable\s+)?struct\s+(\w+)", code))
    defined |= set(re.findall(r"^\s*const\s+(\w+)", code, flags=re.M))
    defined |= set(re.findall(r"\b(\w+)\s*\([^()]*\)\s*=\s", code))         # local one-line closures: sel(o) = ...
    imported = set()
    for m in re.finditer(r"^using\s+[\w.]+:\s*(.+)$", code, flags=re.M):
        imported |= {x.strip() for x in m.group(1).split(",")}
    called = set()
    for m in re.finditer(r"(?<![\w.!:{])([A-Za-z_ℛ][\w!]*)\s*\(", code):
        called.add(m.group(1))
    keywords = {"if", "for", "while", "function", "where", "return", "do", "let", "end", "isa", "in", "using", "module", "const", "ccall"}
    unknown = {c for c in called if c not in defined | imported | JULIA_BASE | keywords}
    assert not unknown, sorted(unknown)
    for name in ("pack", "unpack", "plan", "scale_parts", "hipring", "modring", "upload", "download"):
        assert name in defined, name
    assert "CURRENT_RING" not in code


def header_param_names():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(tfhe_\w+)\s*\(([^;{]*?)\)\s*;", src):
        args = m.group(2).strip()
        names[m.group(1)] = [] if args in ("", "void") else [re.findall(r"\w+", a)[-1] for a in args.split(",")]
    return names


def shim_ccall_args():
    """(symbol, [argument expressions]) of every ccall"""
    src = open(SHIM).read()
    src = "\n".join(l.split("#")[0] if not l.lstrip().startswith("#") else "" for l in src.splitlines())
    out = []
    for m in re.finditer(r"ccall\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        parts = split_top(src[m.end():i - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*lib\s*\)", parts[0]).group(1)
        out.append((sym, parts[3:], src[max(0, m.start() - 200):m.start()]))
    return out


def test_batch_dimension_reaches_every_entry_point():
    """north_star: batches of independent ciphertexts.  No ccall may pass the literal 1 where the C ABI takes a `count` /
    `batch` (round-2 state: every call did); the value must come from the storage's `count` field."""
    names = header_param_names()
    checked = 0
    for sym, args, _ in shim_ccall_args():
        for k, pname in enumerate(names[sym]):
            if pname in ("count", "batch"):
                assert args[k].strip() != "1", f"{sym}: literal 1 passed for `{pname}`"
                checked += 1
    assert checked >= 20
    src = open(SHIM).read()
    assert re.search(r"mutable struct HipVector\{T\}.*?count::Int", src, flags=re.S)
    for name in ("batch", "unbatch", "batchsize"):
        assert re.search(rf"^function {name}\(|^{name}\(", src, flags=re.M), name


def test_device_pointers_are_gc_preserved_and_ordered_across_contexts():
    """`.ptr` operands under GC.@preserve, and no operation enqueued on one context while its operands were
    produced on another without `on(ctx, outs, ins)` (which issues tfhe_ctx_wait_for)."""
    for sym, args, before in shim_ccall_args():
        if any(re.search(r"\b\w+\.ptr\b", a) for a in args) and sym not in ("tfhe_free",):
            assert "GC.@preserve" in before, f"{sym}: device pointers passed without GC.@preserve"
    src = open(SHIM).read()
    assert "function on(ctx::HipRing, outs::Tuple, ins::Tuple)" in src and "wait_for(ctx, v.last)" in src
    # every entry point that enqueues work takes its context through `on`
    for fn in ("NTT.nntt", "NTT.inntt", "ToyFHE.modswitch", "NTT.apply_galois_element", "ToyFHE.keyswitch", "ToyFHE.rotate"):
        body = src[src.index(f"function {fn}("):]
        body = body[:body.index("\nend")]
        assert re.search(r"\bon\(", body) or "pack(ctx" in body, fn


def test_sampler_hook_is_a_rand_method():
    """INTEGRATION.md section 2 names Random.rand(::HipRng, ::RingSampler) as the sampler seam (poly.jl:18-23)."""
    src = open(SHIM).read()
    assert re.search(r"function Random\.rand\(rng::HipRng, r::RingSampler\{ℛ\}\)", src)
    assert "mutable struct HipRng <: Random.AbstractRNG" in src


def test_hoisted_rotation_callers_prepare_their_keys():
    """tfhe_matmul_diag has no `prepared = 0` fallback -- every shim function that calls it must take its key
    pointers from `prepared(gk)` (tfhe_galois_key_prepare), and tfhe_rotate_many must say prepared = 1 when it does the same."""
    src = open(SHIM).read()
    assert "tfhe_galois_key_prepare" in {c[0] for c in shim_ccalls()}
    for m in re.finditer(r"^function (\w+)\(.*?^end", src, flags=re.S | re.M):
        body = m.group(0)
        if ":tfhe_matmul_diag" in body:
            assert "prepared(gk)" in body and "pack(gk.key)" not in body, m.group(1)
            assert "squared_encoding" in body, "a ciphertext-by-plaintext product leaves at the squared scale (ckksencoding.jl:106-111)"
        if ":tfhe_rotate_many" in body:
            uses_prepared = "prepared(gk)" in body
            flag = re.search(r"length\(ek1\.key\),\s*(\d),\s*gs", body)
            assert flag and (flag.group(1) == "1") == uses_prepared, "the `prepared` flag must match the keys passed"


def test_validation_kit_refers_to_things_that_exist():
    """toyfhe.jl_amd/julia/test/runtests.jl and tools/gen_reference_fixtures.jl cannot be run here either: at least every
    `H.name` they use is defined in the shim, every fixture case the generator handles has inputs, and the generator reads
    exactly the header layout toyfhe.jl_amd/wire.py writes."""
    shim = open(SHIM).read()
    defined = set(re.findall(r"\bfunction\s+(?:[\w.]+\.)?([\w!]+)", shim)) | set(re.findall(r"^([\w!]+)\([^=\n]*\)(?:\s+where\s+[^=\n]+)?\s*=(?!=)", shim, flags=re.M))
    kit = open(os.path.join(ROOT, "toyfhe.jl_amd", "julia", "test", "runtests.jl")).read()
    used = set(re.findall(r"\bH\.([\w!]+)", kit))
    assert used and used <= defined, sorted(used - defined)
    gen = open(os.path.join(ROOT, "tools", "gen_reference_fixtures.jl")).read()
    ops = set(re.findall(r'op == "(\w+)"', gen)) | set(re.findall(r'"(\w+)"', " ".join(re.findall(r"op in \(([^)]*)\)", gen))))
    import glob
    import json
    have = {json.load(open(p))["op"] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "ref_julia", "*", "case.json"))}
    assert have and have <= ops, sorted(have - ops)
    from toyfhe_jl_amd import wire
    assert wire._HDR.format == "<8sIIIIIIQQiI" and wire._HDR.size == 56     # magic, 6 x u32, 2 x u64, i32, u32: what readblob / writeblob move
    assert gen.count("read(io, UInt32)") >= 2 and "TFHEWIRE" in gen
