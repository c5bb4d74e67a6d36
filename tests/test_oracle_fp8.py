"""CPU: the fp_8bit<5, Signed> restatement (oracle.c) against an independent numpy transcription of the reference's
encode / decode expressions (cpp/src/neighbors/ivf_pq/ivf_pq_fp_8bit.cuh:52-102)."""
import numpy as np

import oracle

EXP_BITS, VAL_BITS, EXP_MASK = 5, 3, 15


def ref_encode(v, signed):
    def enc_u(a):
        kmin, kmax = 1.0 / (1 << EXP_MASK), float(1 << (EXP_MASK + 1)) * (2.0 - 1.0 / (1 << VAL_BITS))
        if a < kmin:
            return 0
        if a >= kmax:
            return 0xFF
        bits = int(np.float32(a).view(np.uint32))
        return ((bits + (EXP_MASK << 23) - 0x3F800000) >> (15 + EXP_BITS)) & 0xFF

    if not signed:
        return enc_u(v)
    return (enc_u(abs(v)) & 0xFE) | (1 if v < 0 else 0)


def ref_decode_f32(b, signed):
    u = (b & ~1) if signed else b
    base = (0x3F800000 | (0x00400000 >> VAL_BITS)) - (EXP_MASK << 23)
    r = np.uint32(base + (u << (15 + EXP_BITS))).view(np.float32)
    return -r if (signed and (b & 1)) else r


def ref_decode_f16(b, signed):
    u = (b & ~1) if signed else b
    base = (0x3C00 | (0x0200 >> VAL_BITS)) - (EXP_MASK << 10)
    r = np.uint16(base + (u << (2 + EXP_BITS))).view(np.float16)
    return -r if (signed and (b & 1)) else r


def test_fp8_round_trip_matches_the_reference_expressions():
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.random(300, dtype=np.float32) * 4, 10.0 ** rng.uniform(-7, 5.5, 400).astype(np.float32),
                           np.array([0, 2.0 ** -15, 2.0 ** -15 * 0.999, 65536 * 1.875, 1e9], np.float32)])
    for signed in (False, True):
        v = vals if not signed else vals * np.where(rng.random(vals.size) < 0.5, -1, 1).astype(np.float32)
        got32 = oracle.fp8_round_trip(v, signed=signed)
        got16 = oracle.fp8_round_trip(v, signed=signed, to_half=True)
        for x, g32, g16 in zip(v, got32, got16):
            b = ref_encode(float(x), signed)
            assert np.float32(ref_decode_f32(b, signed)) == g32, (x, b)
            w = np.float32(ref_decode_f16(b, signed))
            assert (np.isnan(w) and np.isnan(g16)) or w == g16, (x, b)


def test_fp8_relative_error_and_monotone():
    v = np.sort(10.0 ** np.random.default_rng(1).uniform(-4, 4, 2000)).astype(np.float32)
    r = oracle.fp8_round_trip(v)
    assert (np.abs(r - v) <= v / 16 + 1e-12).all()  # 3 value bits, centred: |err| <= 2^-4 relative
    assert (np.diff(r) >= 0).all()
