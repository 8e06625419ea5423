"""No-GPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/toyfhe_hip.h declares, and fails loudly (never silently falls back) without a device."""
import ctypes
import os
import re

import pytest

import toyfhe_jl_amd as tf
from toyfhe_jl_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "toyfhe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    lib = native.lib()
    decl = _declared_symbols()
    assert len(decl) >= 40
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/toyfhe_hip.h but not exported"
    assert sorted(native.EXPORTED_SYMBOLS) == decl


def test_product_does_not_touch_oracle():
    pkg = os.path.join(ROOT, "toyfhe.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "from oracle" not in src and "import oracle" not in src and "ref_cpu" not in src, f
                assert "libemul" not in src, f


@pytest.mark.skipif(native.device_count() > 0, reason="only meaningful without a GPU")
def test_fails_loudly_without_device():
    with pytest.raises(tf.HipError):
        tf.Context(16, [1099511627873])
    with pytest.raises(tf.HipError):
        tf.DeviceBuffer(16)


def test_argument_validation_precedes_device_use():
    # these are rejected on the host before any HIP call, so they behave the same with or without a GPU
    with pytest.raises(AssertionError):
        tf.Context(12, [1099511627873])            # N not a power of two
    with pytest.raises(AssertionError):
        tf.Context(16, [1099511627873 + 2])        # not prime
    with pytest.raises(AssertionError):
        tf.Context(64, [97])                       # 2N does not divide q-1
    with pytest.raises(AssertionError):
        tf.Context(4, [97, 97])                    # repeated modulus
