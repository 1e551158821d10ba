"""802.11n n = 1944 codes of rates 1/2, 2/3 and 3/4.  PARITY UNPINNED by the reference: it carries the
rate-5/6 matrix only (wifiMatrices.py:6-9), so these tables are restated from the standard and nothing of
the reference's checks them.  CPU part: what can be checked without the standard -- shape, the dual-diagonal
parity part that makes the code encodable, full rank, the rate-5/6 entry against the reference's table,
short cycles (reported).  GPU part: the HIP kernels against this build's oracle on the same matrices, bit
for bit, in both precisions; the float32 path must stay in LDS (short-code kernel), not fall to the
global-memory kernel."""
import numpy as np
import pytest

from conftest import make_llrs

RATES = ('1/2', '2/3', '3/4', '5/6')
SNR = {'1/2': (0.5, 1.0, 2.0), '2/3': (1.5, 2.0, 3.0), '3/4': (2.0, 2.6, 3.5), '5/6': (3.0, 3.5, 4.5)}


def gf2_rank(H):
    """Rank over GF(2) of a 0/1 matrix, rows packed into Python integers."""
    rows = [int(''.join(map(str, r)), 2) for r in np.asarray(H, dtype=np.uint8)]
    rank = 0
    for bit in reversed(range(H.shape[1])):
        piv = next((i for i in range(rank, len(rows)) if (rows[i] >> bit) & 1), None)
        if piv is None:
            continue
        rows[rank], rows[piv] = rows[piv], rows[rank]
        for i in range(len(rows)):
            if i != rank and (rows[i] >> bit) & 1:
                rows[i] ^= rows[rank]
        rank += 1
        if rank == len(rows):
            break
    return rank


@pytest.mark.parametrize('rate', RATES)
def test_prototype_matrix_structure(rate):
    from ldpc_amd import codes
    T = np.array(codes.WIFI_1944_TABLES[rate])
    mb, nb = T.shape
    num, den = map(int, rate.split('/'))
    assert nb == 24 and mb == 24 - 24 * num // den and (T < 81).all() and (T >= -1).all()
    par = T[:, nb - mb:]
    # first parity column: shift 1 at the top and at the bottom, shift 0 in between; then the double diagonal of zeros
    first = [(r, int(v)) for r, v in enumerate(par[:, 0]) if v >= 0]
    assert len(first) == 3 and first[0] == (0, 1) and first[2] == (mb - 1, 1) and first[1][1] == 0
    for r in range(mb):
        for c in range(1, mb):
            assert par[r, c] == (0 if c in (r, r + 1) else -1)
    qc = codes.wifi_1944(rate)
    assert (qc.m, qc.n) == (81 * mb, 1944)
    if rate == '5/6':
        assert qc.key() == codes.wifi_1944_r56().key()
    cycles4 = sum((T[b1, j1] - T[b1, j2] + T[b2, j2] - T[b2, j1]) % 81 == 0
                  for b1 in range(mb) for b2 in range(b1 + 1, mb) for j1 in range(nb) for j2 in range(j1 + 1, nb)
                  if min(T[b1, j1], T[b1, j2], T[b2, j1], T[b2, j2]) >= 0)
    print(f'rate {rate}: {int((T >= 0).sum())} circulants, families of length-4 cycles: {cycles4}')
    assert cycles4 <= 1           # rate 2/3 as restated here holds one (rows 3, 5 x columns 0, 2); the others none


@pytest.mark.parametrize('rate', ('1/2', '3/4'))
def test_full_rank_and_systematic_encoding(rate):
    """The double diagonal makes H = [A | B] with B invertible: parity bits follow from the message by
    back-substitution, and every word built that way satisfies all checks."""
    from ldpc_amd import codes
    qc = codes.wifi_1944(rate)
    H = qc.to_dense().astype(np.uint8)
    assert gf2_rank(H) == qc.m
    k = qc.n - qc.m
    prng = np.random.RandomState(3)
    msg = prng.randint(0, 2, k).astype(np.uint8)
    # solve B p = A msg over GF(2) by elimination on [B | A msg]
    B = H[:, k:].copy()
    rhs = (H[:, :k].astype(np.int32) @ msg.astype(np.int32) % 2).astype(np.uint8)
    aug = np.concatenate([B, rhs[:, None]], axis=1)
    for col in range(qc.m):
        piv = col + int(np.flatnonzero(aug[col:, col])[0])
        aug[[col, piv]] = aug[[piv, col]]
        hit = np.flatnonzero(aug[:, col])
        hit = hit[hit != col]
        aug[hit] ^= aug[col]
    word = np.concatenate([msg, aug[:, -1]])
    assert not (H.astype(np.int32) @ word.astype(np.int32) % 2).any()


@pytest.mark.gpu
@pytest.mark.parametrize('rate', RATES)
def test_gpu_bit_exact_against_oracle(oracle_mod, rate):
    import ldpc_amd
    from ldpc_amd import codes
    qc = codes.wifi_1944(rate)
    H = qc.to_dense()
    orc = oracle_mod.Oracle(H)
    with ldpc_amd.Decoder(qc) as eng:
        info = eng.launch_info(prec=32)
        assert info['lds_bytes'] > 4 * 1944 and info['block'] >= qc.m, info       # the short-code kernel, state in LDS
        spa_info = eng.launch_info(algo=ldpc_amd.ALGO_SUMPRODUCT, prec=32)
        assert spa_info['lds_bytes'] > 4 * 1944
        for snr in SNR[rate]:
            Y = make_llrs(1944 + int(snr * 10), snr, qc.n, 37)                  # ragged against any frames-per-workgroup
            Y[2, 5:50] = 0.0
            for prec, dt in ((32, np.float32), (64, np.float64)):
                for sm in (1, 0):
                    hard, iters, conv, soft = orc.minsum(Y.astype(dt), 50, prec=prec, sign_mode=sm, threads=8)
                    r = eng.decode(Y, 50, prec=prec, sign_mode=sm, want_soft=True)
                    assert np.array_equal(r['iters'], iters), (rate, snr, prec, sm, r['iters'], iters)
                    assert np.array_equal(r['converged'], conv) and np.array_equal(r['hard'], hard)
                    assert np.array_equal(r['soft'], soft)
            # the corrected sign rule decodes these codes where the SNR allows
            hard, iters, conv, _ = orc.minsum(Y, 50, prec=32, sign_mode=1, want_soft=False, threads=8)
            if snr == SNR[rate][-1]:
                assert conv.mean() > 0.9 and hard[conv == 1].sum() == 0, (rate, snr, conv.mean())
            # sum-product on the same frames against the CPU statement, converged frames
            scale = np.float32(4 * 10 ** (snr / 10))
            hs, is_, cs, _ = orc.spa(Y, float(scale), 50, prec=32, sign_mode=1, want_soft=False, threads=8)
            r = eng.decode(Y * scale, 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=1)
            cv = cs == 1
            assert np.array_equal(r['iters'][cv], is_[cv]) and np.array_equal(r['hard'][cv], hs[cv])
        # sweep driver on the same kernels
        cnt, _ = eng.eval_awgn([SNR[rate][-1]], 2000, seed=5, sign_mode=1)
        assert cnt[0, 0] == 2000 and cnt[0, 1] < 200


@pytest.mark.gpu
def test_interleaved_frames_kernel(oracle_mod, monkeypatch):
    """Rates 1/2 ... 3/4 keep one or two frames per workgroup on the short-code kernel with fixed-size column
    records; the kernel with records of the columns' own length, frames interleaved in groups of 2 or 4 (one
    LDS access per edge for the whole group), holds four.  It is selected where it at least doubles the frames
    per workgroup, gives the results of the other kernel and of the oracle bit for bit -- min-sum with either
    sign rule and normalised/offset scaling, soft output, float64 input, frame counts that leave groups and
    blocks partly empty, frames that never converge beside ones that stop at once -- and so do both group
    widths on codes of other shapes (a frame length that is no multiple of 8 takes the byte-wise output path)."""
    import ldpc_amd
    from ldpc_amd import codes
    for rate, expect in (('1/2', 4), ('2/3', 4), ('3/4', 4), ('5/6', 3)):
        qc = codes.wifi_1944(rate)
        with ldpc_amd.Decoder(qc) as eng:
            monkeypatch.delenv('LDPC_HIP_NO_PACKED', raising=False)
            info = eng.launch_info(prec=32)
            assert info['frames_per_wg'] == expect, (rate, info)
            if rate == '5/6':
                continue
            orc = oracle_mod.Oracle(qc.to_dense())
            for F, snr in ((1, 2.0), (3, 1.0), (4, 2.5), (5, 0.0), (37, 1.5), (1031, 2.0)):
                Y = make_llrs(7 * F + 1, snr, qc.n, F)
                if F >= 5:
                    Y[1] = np.float32(-1.0)                  # a codeword at once, beside frames that run long
                    Y[4, ::3] = 0.0
                for sm in (1, 0):
                    monkeypatch.delenv('LDPC_HIP_NO_PACKED', raising=False)
                    a = eng.decode(Y, 20, prec=32, sign_mode=sm, want_soft=True)
                    monkeypatch.setenv('LDPC_HIP_NO_PACKED', '1')
                    assert eng.launch_info(prec=32)['frames_per_wg'] < expect
                    b = eng.decode(Y, 20, prec=32, sign_mode=sm, want_soft=True)
                    for key in ('hard', 'iters', 'converged', 'ones', 'soft'):
                        assert np.array_equal(a[key], b[key]), (rate, F, sm, key)
                    if F <= 37:
                        hard, iters, conv, soft = orc.minsum(Y, 20, prec=32, sign_mode=sm, threads=8)
                        assert np.array_equal(a['iters'], iters) and np.array_equal(a['hard'], hard)
                        assert np.array_equal(a['converged'], conv) and np.array_equal(a['soft'], soft)
            monkeypatch.delenv('LDPC_HIP_NO_PACKED', raising=False)
            # float64 channel values into the float32 kernel, and normalised min-sum
            Y = make_llrs(99, 1.5, qc.n, 9)
            a = eng.decode(Y.astype(np.float64), 20, prec=32, sign_mode=1)
            b = eng.decode(Y, 20, prec=32, sign_mode=1)
            assert np.array_equal(a['iters'], b['iters']) and np.array_equal(a['hard'], b['hard'])
            eng.set_minsum_scaling(0.8125, 0.0)
            a = eng.decode(Y, 20, prec=32, sign_mode=1)
            hard, iters, conv, _ = orc.minsum(Y, 20, prec=32, sign_mode=1, alpha=0.8125, beta=0.0, want_soft=False, threads=8)
            assert np.array_equal(a['iters'], iters) and np.array_equal(a['hard'], hard)
    # other shapes: random sparse matrices, both group widths forced
    prng = np.random.RandomState(77)
    for (m, n, dv) in ((40, 100, 3), (97, 203, 4), (250, 504, 3)):
        H = np.zeros((m, n), np.int32)
        for c in range(n):
            H[prng.choice(m, dv, replace=False), c] = 1
        for r in range(m):                                   # every row at least 2 ones
            while H[r].sum() < 2:
                H[r, prng.randint(n)] = 1
        if H.sum(1).max() > 24 or H.sum(0).max() > 12:
            continue
        orc = oracle_mod.Oracle(H)
        Y = make_llrs(m, 3.0, n, 77)
        hard, iters, conv, soft = orc.minsum(Y, 15, prec=32, sign_mode=1, threads=8)
        with ldpc_amd.Decoder(H) as eng:
            for rpt in ('2', '4'):
                monkeypatch.setenv('LDPC_HIP_PACKED_RPT', rpt)
                info = eng.launch_info(prec=32)
                assert info['frames_per_wg'] % int(rpt) == 0, info
                a = eng.decode(Y, 15, prec=32, sign_mode=1, want_soft=True)
                assert np.array_equal(a['iters'], iters) and np.array_equal(a['hard'], hard), (m, n, rpt)
                assert np.array_equal(a['converged'], conv) and np.array_equal(a['soft'], soft)
                assert np.array_equal(a['ones'], hard.sum(1))
            monkeypatch.delenv('LDPC_HIP_PACKED_RPT')
