"""Sum-product (algo 1).  PARITY UNPINNED by the reference: it has no sum-product rule,
so the only oracle is this build's own CPU statement (oracle/spa_impl.h).  CPU part:
that statement against an independent pure-Python one and against decoding sanity.
GPU part: the HIP kernel against the CPU statement in the same precision."""
import math

import numpy as np
import pytest

from conftest import build_code, make_llrs


def spa_python(H, llr, max_iters):
    """Independent float64 statement (Python floats), edge lists built from the dense matrix."""
    m, n = H.shape
    rows = [np.flatnonzero(H[r]).tolist() for r in range(m)]
    phi = lambda x: math.log1p(2.0 / math.expm1(x))
    clamp = lambda x: 1e-12 if x < 1e-12 else (80.0 if x > 80.0 else x)
    post = [float(v) + 0.0 for v in llr]
    c2v = [[0.0] * len(rw) for rw in rows]
    synd = lambda: all(sum(post[c] > 0 for c in rw) % 2 == 0 for rw in rows)
    it, ok = 0, synd()
    while not ok and it < max_iters:
        it += 1
        for r, rw in enumerate(rows):
            v = [post[c] - c2v[r][k] for k, c in enumerate(rw)]
            F = [phi(clamp(abs(x))) for x in v]
            F1, arg, rest = -1.0, -1, 0.0
            for k, f in enumerate(F):
                if f > F1:
                    if arg >= 0:
                        rest = rest + F1
                    F1, arg = f, k
                else:
                    rest = rest + f
            neg = sum(x < 0 for x in v) % 2
            for k in range(len(rw)):
                x = rest if k == arg else (rest - F[k]) + F1
                mag = phi(clamp(x))
                c2v[r][k] = -mag if ((v[k] < 0) ^ bool(neg)) else mag
        acc = [0.0] * n
        for r, rw in enumerate(rows):                 # ascending check index per column
            for k, c in enumerate(rw):
                acc[c] = acc[c] + c2v[r][k]
        post = [acc[c] + (float(llr[c]) + 0.0) for c in range(n)]
        ok = synd()
    return np.array([p > 0 for p in post], np.uint8), it, ok, np.array(post)


def test_oracle_spa_matches_python_statement(oracle_mod):
    _, H = build_code('rand_24_48')
    orc = oracle_mod.Oracle(H)
    y = make_llrs(11, 2.0, H.shape[1], 12)
    scale = 4 * 10 ** 0.2
    hard, iters, conv, soft = orc.spa(y, scale, 20, prec=64)
    for k in range(len(y)):
        h, it, ok, post = spa_python(H, y[k].astype(np.float64) * scale, 20)
        assert it == iters[k] and bool(ok) == bool(conv[k])
        assert np.array_equal(h, hard[k])
        assert np.array_equal(post, soft[k])          # same libm, same order: bit-identical


def test_oracle_spa_decodes_better_than_minsum(oracle_mod):
    _, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    y = make_llrs(5, 3.1, H.shape[1], 48)
    scale = 4 * 10 ** 0.31
    _, it_ms, conv_ms, _ = orc.minsum(y, 50, prec=64, want_soft=False, threads=8)
    hard, it_sp, conv_sp, _ = orc.spa(y, scale, 50, prec=64, want_soft=False, threads=8)
    assert conv_sp.sum() >= conv_ms.sum() and conv_sp.mean() > 0.9
    assert hard[conv_sp == 1].sum() == 0              # converged frames are the transmitted all-zero word
    assert it_sp.mean() < it_ms.mean()
    # a clean codeword needs no iteration, a scaled one neither
    h0, i0, c0, _ = orc.spa(-np.ones((1, H.shape[1])), 7.0, 50, prec=64)
    assert i0[0] == 0 and c0[0] == 1 and h0.sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('name,snrs', [('nearEarth', (3.0, 3.5)), ('wifi_5_6', (4.0, 6.0)), ('rand_24_48', (2.0,)),
                                       ('nearEarth_mut_w3', (3.4,)), ('nearEarth_mut_w7', (3.6,))])
@pytest.mark.parametrize('prec', [64, 32])
def test_gpu_spa_against_cpu_statement(oracle_mod, name, snrs, prec):
    """Same arithmetic type on both sides; the two differ only in the last bits of
    log1p/expm1 (device math library against the host's), which decides nothing on a
    frame that converges.  Bar: identical iteration counts, status and bits on every
    frame that converges on the CPU, and on >= 99 % of all frames.  (The 802.11n code has
    odd check degrees; with the reference's sign rule nothing converges there and the
    trajectories are chaotic, so it is run with the corrected rule.)"""
    import ldpc_amd
    qc, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    n = H.shape[1]
    dt = np.float64 if prec == 64 else np.float32
    sm = ldpc_amd.SIGN_CORRECTED if name == 'wifi_5_6' else ldpc_amd.SIGN_REFERENCE
    with ldpc_amd.Decoder(qc if qc is not None else H) as eng:
        same = total = 0
        for snr in snrs:
            F = 48
            y = make_llrs(97, snr, n, F)
            scale = 4 * 10 ** (snr / 10)
            hard, iters, conv, soft = orc.spa(y, scale, 50, prec=prec, sign_mode=sm, threads=8)
            llr = y.astype(dt) * dt(scale)
            r = eng.decode(llr, 50, prec=prec, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=sm, want_soft=True)
            ok = (r['iters'] == iters) & (r['converged'] == conv) & (r['hard'] == hard).all(axis=1)
            assert ok[conv == 1].all(), (name, snr, prec, np.flatnonzero(~ok))
            assert np.array_equal(r['ones'], r['hard'].sum(1))
            # soft values: a few ulps of the transcendental functions, amplified by the iteration
            cv = conv == 1
            if cv.any():
                assert np.allclose(r['soft'][cv], soft[cv], rtol=1e-3 if prec == 32 else 1e-9, atol=1e-3 if prec == 32 else 1e-9)
            same += int(ok.sum())
            total += F
        assert same >= 0.99 * total, (name, prec, same, total)


@pytest.mark.gpu
def test_gpu_spa_sweep_and_corrected_sign():
    import ldpc_amd
    from ldpc_amd import codes
    with ldpc_amd.Decoder(codes.near_earth()) as eng:
        c_ms, _ = eng.eval_awgn([3.0, 3.4], 512, seed=3, max_iters=50, prec=32)
        c_sp, pf = eng.eval_awgn([3.0, 3.4], 512, seed=3, max_iters=50, prec=32,
                                 algo=ldpc_amd.ALGO_SUMPRODUCT, per_frame=True)
        f = ldpc_amd.COUNTER_FIELDS
        fe = f.index('frame_errors')
        assert (c_sp[:, f.index('frames')] == 512).all()
        assert (c_sp[:, fe] <= c_ms[:, fe]).all() and c_sp[0, fe] < c_ms[0, fe]
        assert np.array_equal(c_sp[:, f.index('uncoded_errors')], c_ms[:, f.index('uncoded_errors')])   # same noise
        assert int(pf[..., 2].sum()) == int(c_sp[:, f.index('bit_errors')].sum())
        with pytest.raises(ldpc_amd.LdpcHipError):
            eng.decode(np.zeros((1, 8176), np.float32), 5, algo=ldpc_amd.ALGO_SUMPRODUCT, chained=True)
    # odd check degrees: the corrected sign rule decodes, the reference rule cannot
    with ldpc_amd.Decoder(codes.wifi_1944_r56()) as eng:
        y = make_llrs(21, 6.0, 1944, 32)
        llr = y * np.float32(4 * 10 ** 0.6)
        good = eng.decode(llr, 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=ldpc_amd.SIGN_CORRECTED)
        assert good['converged'].all() and good['hard'].sum() == 0


@pytest.mark.gpu
def test_gpu_spa_edge_list_kernel_on_near_earth(oracle_mod, monkeypatch):
    """float32 NearEarth normally takes the message-plane kernel; the edge-list kernel
    (exact log1p/expm1 of the device math library) must agree with it and with the CPU."""
    import ldpc_amd
    qc, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    y = make_llrs(31, 3.3, H.shape[1], 24)
    scale = 4 * 10 ** 0.33
    hard, iters, conv, _ = orc.spa(y, scale, 50, prec=32, want_soft=False, threads=8)
    llr = y * np.float32(scale)
    with ldpc_amd.Decoder(qc) as eng:
        fast = eng.decode(llr, 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT)
        assert eng.launch_info(algo=ldpc_amd.ALGO_SUMPRODUCT, prec=32)['lds_bytes'] == 163824
        monkeypatch.setenv('LDPC_HIP_SPA_EDGE_LIST', '1')
        slow = eng.decode(llr, 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT)
        assert eng.launch_info(algo=ldpc_amd.ALGO_SUMPRODUCT, prec=32)['lds_bytes'] == 163520
    cv = conv == 1
    assert cv.sum() >= 20
    for r in (fast, slow):
        assert np.array_equal(r['iters'][cv], iters[cv]) and np.array_equal(r['hard'][cv], hard[cv])
        assert np.array_equal(r['converged'][cv], conv[cv])
