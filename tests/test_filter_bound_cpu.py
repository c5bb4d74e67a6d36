"""CPU check of the rounding margins of the matrix-core filter (cuvs_amd/csrc/ivf_pq_scan3.hip: filter_threshold,
filter_threshold_ip). The filter may drop a (row, query) pair only when its exact score - the reference's LUT arithmetic
(create_lut_impl.cuh:17-78, compute_score_impl.cuh:52-79), restated here in numpy for the fp32 and fp16 LUT / score types -
is above the query's bound. The weakest case is a row exactly AT the bound: with bound := its exact score it has to
survive, i.e. the screened value must not exceed the threshold. The GEMM is emulated the way the hardware defines it:
operands scaled by a power of two and rounded to fp16, products exact, accumulation in fp32 (any order)."""
import numpy as np
import pytest

F32, F16 = np.float32, np.float16
pytestmark = [pytest.mark.filterwarnings("ignore:overflow encountered"), pytest.mark.filterwarnings("ignore:invalid value")]


def fp8_round_trip(v, signed, half):
    """pq_lut_math.hpp fp8_round_trip (the reference's fp_8bit<5, Signed>, ivf_pq_fp_8bit.cuh:32-100): truncating encode,
    half an ulp added back on decode, float or half decoder"""
    v = F32(v)
    av = abs(v) if signed else v
    if av < F32(1.0 / 32768.0):
        u = 0
    elif av >= F32(65536.0 * 1.875):
        u = 0xFF
    else:
        bits = int(np.array(av, F32).view(np.uint32))
        u = ((bits + (15 << 23) - 0x3F800000) >> 20) & 0xFF
    neg = signed and v < 0
    if signed:
        u &= 0xFE
    if half:
        hb = ((0x3C00 | (0x0200 >> 3)) - (15 << 10)) + (u << 7)
        r = F32(np.array(hb & 0xFFFF, np.uint16).view(F16))
    else:
        r = np.array((((0x3F800000 | (0x00400000 >> 3)) - (15 << 23)) + (u << 20)) & 0xFFFFFFFF, np.uint32).view(F32)
    return F32(-r if neg else r)


def test_fp8_emulation_equals_the_oracle():
    import oracle

    rng = np.random.default_rng(3)
    v = np.concatenate([rng.standard_normal(500) * 10.0 ** rng.uniform(-6, 6, 500), [0.0, 1e-9, 122880.0, 1e7]]).astype(F32)
    for signed in (False, True):
        for half in (False, True):
            vv = v if signed else np.abs(v)
            want = oracle.fp8_round_trip(vv, signed=signed, to_half=half)
            got = np.array([fp8_round_trip(x, signed, half) for x in vv], F32)
            ok = (got == want) | (np.isnan(got) & np.isnan(want))
            assert ok.all(), (signed, half, vv[~ok][:5], got[~ok][:5], want[~ok][:5])


def lut_score(r, cb, codes, lut, acc, ip=False, q=None, c=None):
    """exact score of one row: entries fma(d1, d1, fma(d0, d0, 0)) (L2) / the four-fma chain (IP), rounded to the LUT type,
    summed in subspace order in the score type"""
    s16, s32 = F16(0), F32(0)
    for s in range(cb.shape[0]):
        p0, p1 = cb[s, codes[s]]
        if not ip:
            d0, d1 = F32(r[2 * s] - p0), F32(r[2 * s + 1] - p1)
            v = F32(np.float64(d1) * np.float64(d1) + np.float64(F32(np.float64(d0) * np.float64(d0))))  # fma(d1, d1, fma(d0, d0, 0))
        else:
            v = F32(np.float64(-q[2 * s]) * np.float64(c[2 * s]))
            v = F32(np.float64(-q[2 * s]) * np.float64(p0) + np.float64(v))
            v = F32(np.float64(-q[2 * s + 1]) * np.float64(c[2 * s + 1]) + np.float64(v))
            v = F32(np.float64(-q[2 * s + 1]) * np.float64(p1) + np.float64(v))
        if lut == "fp8":
            v = fp8_round_trip(v, ip, acc == "f16")
        e = v if (lut == "f32" or (lut == "fp8" and acc != "f16")) else F16(v)
        if acc == "f16":
            s16 = F16(s16 + e)
        else:
            s32 = F32(s32 + F32(e))
    return float(s16 if acc == "f16" else s32)


def gemm16(x, y, sc, rng):
    """sum of fl16(sc x_i) * fl16(sc y_i) accumulated in fp32, in a random order"""
    xa, ya = F16(F32(sc) * x).astype(np.float64), F16(F32(sc) * y).astype(np.float64)
    acc = F32(0)
    for i in rng.permutation(len(x)):
        acc = F32(np.float64(acc) + xa[i] * ya[i])
    return float(acc)


def eps_alpha(lut, acc, ip=False):
    if lut == "f32":
        return 1.0 / 65536.0, 0.0
    if lut == "fp8":
        if ip:
            return (0.18 if acc == "f16" else 0.14), 64.0 / 32768.0
        return (0.11 if acc == "f16" else 0.07), 64.0 / 32768.0
    return (0.04 if acc == "f16" else 1.0 / 1024.0), 64.0 / 16777216.0


@pytest.mark.parametrize("lut,acc", [("f32", "f32"), ("f16", "f32"), ("f16", "f16"), ("fp8", "f16"), ("fp8", "f32")])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 37.0, 2.5e3])
def test_l2_row_at_the_bound_survives(lut, acc, scale):
    rng = np.random.default_rng(int(scale * 7) + len(lut) + len(acc))
    pq_dim, D = 64, 128
    cb = (rng.standard_normal((pq_dim, 256, 2)) * scale).astype(F32)
    cbmax = float(np.abs(cb).max())
    sc = 2.0 ** np.floor(np.log2(16.0 / cbmax))
    eps, alpha = eps_alpha(lut, acc)
    worst = -np.inf
    for trial in range(300):
        codes = rng.integers(0, 256, pq_dim)
        d = cb[np.arange(pq_dim), codes].reshape(-1)  # decoded residual
        # residuals from "right on top of the row" to far away, incl. tiny and lopsided ones
        kind = trial % 4
        if kind < 2:
            r = (d + rng.standard_normal(D).astype(F32) * F32(scale * 10.0 ** rng.uniform(-4, 1))).astype(F32)
        else:
            r = (rng.standard_normal(D) * scale * 10.0 ** rng.uniform(-3, 1.5)).astype(F32)
        if kind == 3:
            r[rng.integers(0, D, 100)] = 0
        s_exact = lut_score(r, cb, codes, lut, acc)
        if not np.isfinite(s_exact) or s_exact > (30000 if lut == "fp8" else 60000):  # bound_max: not served by the filter
            continue
        rn = float(np.sum(r.astype(np.float64) ** 2, dtype=np.float64))
        dn = F32(0)
        for s in range(pq_dim):  # row_term_kernel's fp32 chain
            dn = F32(np.float64(cb[s, codes[s], 0]) ** 2 + np.float64(dn))
            dn = F32(np.float64(cb[s, codes[s], 1]) ** 2 + np.float64(dn))
        x = F32(-0.5) * F32(sc) * F32(sc) * F32(dn * F32(1.0 - 1.0 / 512.0))
        hi = F16(x)
        lo = F16(F32(x) - F32(hi))
        accv = F32(np.float64(F32(gemm16(r, d, sc, rng))) + np.float64(hi) + np.float64(lo))  # K-extension step
        c1 = -2.0 / (sc * sc)
        screened = float(accv) * c1
        # filter_threshold(bound = s_exact)
        mabs = 2.0 ** -23 / sc * (np.sqrt(D * rn) + D * cbmax)
        thr = (s_exact + alpha) * (1.0 + 2.0 * eps) + mabs - rn * (1.0 - 1.0 / 512.0)
        thr = float(F32(thr)) + abs(float(F32(thr))) * 2.4e-7 + 1e-37
        worst = max(worst, screened - thr)
        assert screened <= thr, (trial, screened, thr, s_exact, rn, float(dn))
    assert worst < 0


@pytest.mark.parametrize("lut,acc", [("f32", "f32"), ("f16", "f32"), ("f16", "f16"), ("fp8", "f16"), ("fp8", "f32")])
def test_inner_product_row_at_the_bound_survives(lut, acc):
    rng = np.random.default_rng(11 + len(lut) + len(acc))
    pq_dim, D = 64, 128
    for scale in (0.05, 1.0, 40.0):
        cb = (rng.standard_normal((pq_dim, 256, 2)) * scale).astype(F32)
        cbmax = float(np.abs(cb).max())
        sc = 2.0 ** np.floor(np.log2(16.0 / cbmax))
        dmax = float(np.sqrt((np.max(np.sum(cb.astype(np.float64) ** 2, axis=2), axis=1)).sum())) * 1.0001
        eps, alpha = eps_alpha(lut, acc, ip=True)
        for trial in range(150):
            codes = rng.integers(0, 256, pq_dim)
            d = cb[np.arange(pq_dim), codes].reshape(-1)
            c = (rng.standard_normal(D) * scale * 10.0 ** rng.uniform(-1, 1)).astype(F32)
            q = (rng.standard_normal(D) * 10.0 ** rng.uniform(-2, 1)).astype(F32)
            s_exact = lut_score(None, cb, codes, lut, acc, ip=True, q=q, c=c)
            if not np.isfinite(s_exact) or abs(s_exact) > (30000 if lut == "fp8" else 60000):
                continue
            qn = float(np.sum(q.astype(np.float64) ** 2))
            cn = float(np.sum(c.astype(np.float64) ** 2))
            qc = F32(0)
            for i in range(D):
                qc = F32(np.float64(q[i]) * np.float64(c[i]) + np.float64(qc))
            screened = gemm16(q, d, sc, rng) * (-1.0 / (sc * sc))
            nq, nc = np.sqrt(qn), np.sqrt(cn)
            mabs = 2.0 ** -23 / sc * (np.sqrt(D) * nq + D * cbmax)
            m = nq * ((eps + 7.63e-6) * nc + (eps + 1.0 / 512.0) * dmax)
            thr = s_exact + float(qc) + (abs(float(qc)) + abs(s_exact)) * 1e-6 + m + mabs + alpha
            assert screened <= thr, (scale, trial, screened, thr, s_exact)


# ---------------------------------------------------------------------------------------------- IVF-Flat (FLAT builds of the filter)
# ivf_pq_scan3.hip flat3_prepare / flat3_tail: A operands = fl16(sc (x - c)) (cosine: x / |x| - c), B operands = fl16(sc (q - c))
# (dot-product metrics: fl16(sc q)), the exact score is the scan kernel's fp32 fma chain in dimension order.
def chain_l2(q, x):
    acc = F32(0)
    for i in range(len(q)):
        t = F32(q[i] - x[i])
        acc = F32(np.float64(t) * np.float64(t) + np.float64(acc))
    return acc


def chain_dot(x, q):
    acc = F32(0)
    for i in range(len(q)):
        acc = F32(np.float64(x[i]) * np.float64(q[i]) + np.float64(acc))
    return acc


def flat_rows(rng, n, D, kind):
    """rows of one list around a centre: floats of any scale, or integer-valued rows as int8 / uint8 indexes hold them"""
    if kind == "int8":
        c = rng.uniform(-40, 40, D).astype(F32)
        x = np.clip(np.rint(c + rng.standard_normal((n, D)) * rng.uniform(1, 60)), -128, 127).astype(F32)
    elif kind == "uint8":
        c = rng.uniform(60, 200, D).astype(F32)
        x = np.clip(np.rint(c + rng.standard_normal((n, D)) * rng.uniform(1, 60)), 0, 255).astype(F32)
    else:
        scale = 10.0 ** rng.uniform(-3, 3)
        c = (rng.standard_normal(D) * scale).astype(F32)
        x = (c + rng.standard_normal((n, D)).astype(F32) * F32(scale * 10.0 ** rng.uniform(-2, 0.5))).astype(F32)
    return c, x


def flat_tables(c, x):
    """what flat_residual_stats_kernel leaves: the scaling, the largest |component| and the largest norm of a residual"""
    d = (x - c).astype(F32)
    maxres = float(np.abs(d).max())
    sc = 2.0 ** np.floor(np.log2(16.0 / maxres)) if maxres > 0 else 1.0
    dn = np.array([float(chain_dot(r, r)) for r in d], F32)  # the kernel's fp32 chain of squares
    return d, sc, maxres, dn


@pytest.mark.parametrize("kind", ["float", "int8", "uint8"])
def test_flat_l2_row_at_the_bound_survives(kind):
    rng = np.random.default_rng({"float": 21, "int8": 22, "uint8": 23}[kind])
    D = 128
    for rep in range(12):
        c, x = flat_rows(rng, 24, D, kind)
        d, sc, maxres, dn = flat_tables(c, x)
        for i in range(len(x)):
            # queries from "on top of the row" to far away (integer-valued for the integer kinds: the queries' type is the rows')
            q = (x[i] + rng.standard_normal(D) * maxres * 10.0 ** rng.uniform(-3, 1)).astype(F32)
            if kind != "float":
                q = np.clip(np.rint(q), -128 if kind == "int8" else 0, 127 if kind == "int8" else 255).astype(F32)
            s_exact = float(chain_l2(q, x[i]))
            r = (q - c).astype(F32)
            if np.abs(F32(sc) * r).max() >= 60000:  # not served: the query survives everything
                continue
            rn = float(chain_dot(r, r))
            t = F32(-0.5) * F32(sc) * F32(sc) * F32(dn[i] * F32(1.0 - 1.0 / 512.0))
            hi = F16(t)
            lo = F16(F32(t) - F32(hi))
            acc = F32(np.float64(F32(gemm16(r, d[i], sc, rng))) + np.float64(hi) + np.float64(lo))
            screened = float(acc) * (-2.0 / (sc * sc))
            eps = 1.0 / 65536.0
            mabs = 2.0 ** -23 / sc * (np.sqrt(D * rn) + D * maxres)
            thr = s_exact * (1.0 + 2.0 * eps) + mabs - rn * (1.0 - 1.0 / 512.0)
            thr = float(F32(thr)) + abs(float(F32(thr))) * 2.4e-7 + 1e-37
            assert screened <= thr, (kind, rep, i, screened, thr, s_exact, rn, float(dn[i]))


@pytest.mark.parametrize("cosine", [False, True])
@pytest.mark.parametrize("kind", ["float", "int8"])
def test_flat_dot_product_row_at_the_bound_survives(kind, cosine):
    """inner product: score -(x . q) by the chain acc = fma(x, q, acc); cosine: -(x . q) / (|q| |x|) with the norms' own chains,
    screened on unit-length operands (flat3_view::unit_rows, ivf_flat.hip q_unit)"""
    rng = np.random.default_rng(31 + (kind == "int8") + 2 * cosine)
    D = 128
    for rep in range(12):
        c, x = flat_rows(rng, 24, D, kind)
        if cosine:
            inv = np.array([F32(1.0) / np.sqrt(chain_dot(r, r)) if chain_dot(r, r) > 0 else F32(0) for r in x], F32)
            xu = (x * inv[:, None]).astype(F32)
            c = xu.mean(0).astype(F32)  # the centres of a cosine index are means of unit-length rows
            d, sc, maxres, dn = flat_tables(c, xu)
        else:
            d, sc, maxres, dn = flat_tables(c, x)
        dmax = float(np.sqrt(dn.max())) * (1.0 + 1e-6)
        eps = D / 4194304.0
        for i in range(len(x)):
            q = (x[i] * 10.0 ** rng.uniform(-1, 1) + rng.standard_normal(D) * np.abs(x[i]).max() * 10.0 ** rng.uniform(-2, 1)).astype(F32)
            if kind != "float":
                q = np.clip(np.rint(q), -128, 127).astype(F32)
            if not np.any(q):
                continue
            dot = chain_dot(x[i], q)
            if cosine:
                s_exact = -float(F32(dot / F32(np.sqrt(chain_dot(q, q)) * np.sqrt(chain_dot(x[i], x[i])))))
                qf = (q / F32(np.sqrt(np.sum(q.astype(np.float64) ** 2)))).astype(F32)  # normalize_rows
            else:
                s_exact = -float(dot)
                qf = q
            if np.abs(F32(sc) * qf).max() >= 60000:
                continue
            qn = float(np.sum(qf.astype(np.float64) ** 2))
            cn = float(np.sum(c.astype(np.float64) ** 2))
            qc = float(chain_dot(qf, c))
            screened = gemm16(qf, d[i], sc, rng) * (-1.0 / (sc * sc))
            nq, nc = np.sqrt(qn), np.sqrt(cn)
            mabs = 2.0 ** -23 / sc * (np.sqrt(D) * nq + D * maxres)
            m = nq * ((eps + 7.63e-6) * nc + (eps + 1.0 / 512.0) * dmax)
            thr = s_exact + qc + (abs(qc) + abs(s_exact)) * 1e-6 + m + mabs
            thr = float(F32(thr)) + abs(float(F32(thr))) * 2.4e-7 + 1e-37
            assert screened <= thr, (kind, cosine, rep, i, screened, thr, s_exact, qc)
