"""TEST INFRASTRUCTURE -- not part of the product (see oracle/spec.py).

The same restatement over the Goldilocks field p = 2^64 - 2^32 + 1.  The
reference instantiates its generic code for `GoldilocksField = Fp64<MontBackend<GoldilocksMontConfig, 1>>`
with modulus 18446744069414584321 and generator 7 (mpc/src/common/math/goldilocks.rs:4-13); ark derives the
two-adicity (32) and TWO_ADIC_ROOT_OF_UNITY = GENERATOR^((p-1)/2^32) from those, exactly as for bls12-381 Fr.
So this module is oracle/spec.py loaded a second time with the field constants swapped: every function
(and the reference file:line it cites) is unchanged.  PARITY UNPINNED, as for Fr: the reference holds no
stored vectors for this path and cannot be built here.
"""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("oracle._spec_goldilocks",
                                               os.path.join(os.path.dirname(os.path.abspath(__file__)), "spec.py"))
S = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(S)
S.R_MOD = 18446744069414584321  # goldilocks.rs:6
S.GENERATOR = 7                 # goldilocks.rs:7
S.TWO_ADICITY = 32              # p - 1 = 2^32 * (2^32 - 1)
S.TWO_ADIC_ROOT = pow(S.GENERATOR, (S.R_MOD - 1) >> S.TWO_ADICITY, S.R_MOD)
assert S.R_MOD == 2 ** 64 - 2 ** 32 + 1 and (S.R_MOD - 1) % (1 << 32) == 0 and ((S.R_MOD - 1) >> 32) % 2 == 1
P = S.R_MOD
