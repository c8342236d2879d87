"""CPU tests of the "hbmpc-chacha20-v1" coefficient stream restated in oracle/spec.py (include/hbmpc_hip.h,
seeded compute_shares).  The block function is pinned by the RFC 8439 section 2.3.2 test vector; the sampler by its
defining properties (range, determinism, position independence)."""
from oracle import spec as S
from oracle import spec_gl


def test_chacha20_block_rfc8439_vector():
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    # RFC 8439 2.3.2: block counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00 -> state words 12..15
    blk = S._chacha20_block(key, (0x09000000 << 32) + 1, 0x4A000000)
    want = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
            0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert blk == want


def test_seeded_coefficient_properties():
    seed = list(range(1, 9))
    for mod, sp in ((S.R_MOD, S), (spec_gl.P, spec_gl.S)):
        vals = [sp.seeded_coefficient(seed, b, k) for b in range(20) for k in range(1, 6)]
        assert all(0 <= v < mod for v in vals)
        assert len(set(vals)) == len(vals)
        assert vals == [sp.seeded_coefficient(seed, b, k) for b in range(20) for k in range(1, 6)]
        assert sp.seeded_coefficient(seed, 3, 2) != sp.seeded_coefficient([9] + seed[1:], 3, 2)
        poly = sp.seeded_polynomial(seed, 7, mod + 5, 4)
        assert poly[0] == 5 and poly[1:] == [sp.seeded_coefficient(seed, 7, k) for k in range(1, 5)]
        assert sp.seeded_polynomial(seed, 7, None, 4) == [sp.seeded_coefficient(seed, 7, k) for k in range(5)]


def test_goldilocks_rejection_is_exercised():
    # candidates >= p have probability 2^-32 per draw, too rare to meet by search: feed the sampler a block function
    # whose first candidate is p itself and check that the next one is taken
    sp = spec_gl.S
    real = sp._chacha20_block
    p = spec_gl.P
    try:
        sp._chacha20_block = lambda key, ctr, nonce: [p & 0xFFFFFFFF, p >> 32, 0xFFFFFFFF, 0xFFFFFFFF, 5, 0] + [0] * 10
        assert sp.seeded_coefficient([0] * 8, 0, 1) == 5
    finally:
        sp._chacha20_block = real
