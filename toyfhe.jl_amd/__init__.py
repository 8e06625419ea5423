"""toyfhe.jl_amd -- MI355X-native engine for the power-of-two-cyclotomic RNS path of ToyFHE.jl.

The directory name contains a dot, so import it through the repo-root loader module:
``import toyfhe_jl_amd``.  Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI),
``native`` (ctypes binding) and the host-side mirror of the reference's ring/ciphertext interface.
"""
from . import native  # noqa: F401
from .native import BfvPlan, Comm, Context, DeviceBuffer, Event, HipError, UsageError  # noqa: F401
from . import ring, she  # noqa: F401,E402
from .ring import NegacyclicRing, PlainElement, PlainRing, RingElement, nextprime, plaintext_space  # noqa: F401,E402
from . import wire  # noqa: F401,E402
from .she import (DeviceRng, BFVParams, BGVParams, CKKSParams, CipherText, ModulusRaised, apply_galois_element,  # noqa: F401,E402
                  ckks_decode, ckks_encode, decrypt, enc_mul, encrypt, invariant_noise_budget, keygen, keygen_evalmult, keygen_galois,
                  keyswitch, make_eval_key, matmul_diag, modswitch, rotate, rotate_many)
