"""Randomly drawn quasi-cyclic and unstructured codes through every kernel the library can
select (message planes, short-code slots, per-block regular, table-driven, edge list),
bit-exact against the CPU oracle in both precisions and both sign rules."""
import numpy as np
import pytest

from conftest import make_llrs

pytestmark = pytest.mark.gpu


def random_qc(prng):
    from ldpc_amd import codes
    while True:
        mb = int(prng.randint(1, 5))
        nb = int(prng.randint(mb + 1, 13))
        Z = int(prng.choice([5, 7, 16, 27, 31, 64, 65, 81, 96, 127, 128, 130]))
        wmax = int(prng.randint(1, 4))
        blocks = [[sorted(prng.choice(Z, int(prng.randint(0, wmax + 1)), replace=False).tolist())
                   for _ in range(nb)] for _ in range(mb)]
        qc = codes.QCCode(Z, blocks)
        rd = qc.row_degrees()
        if min(rd) >= 2 and max(rd) <= 60 and min(qc.col_degrees()) >= 1:
            return qc


def random_dense(prng):
    m = int(prng.randint(6, 40))
    n = int(prng.randint(m + 2, 3 * m))
    H = (prng.rand(m, n) < min(0.5, 4.0 / m)).astype(np.int32)
    for r in range(m):                                   # at least two ones per row, one per column
        while H[r].sum() < 2:
            H[r, prng.randint(n)] = 1
    for c in range(n):
        if H[:, c].sum() == 0:
            H[prng.randint(m), c] = 1
    return H


@pytest.mark.parametrize('seed', range(24))
def test_random_codes_bit_exact(oracle_mod, seed):
    import ldpc_amd
    prng = np.random.RandomState(1000 + seed)
    if seed % 4 == 3:
        qc, H = None, random_dense(prng)
    else:
        qc = random_qc(prng)
        H = qc.to_dense()
    orc = oracle_mod.Oracle(H)
    n = H.shape[1]
    F = int(prng.randint(1, 70))
    Y = make_llrs(seed + 5, float(prng.uniform(0.0, 6.0)), n, F)
    if F > 2:
        Y[1, : n // 3] = 0.0                             # erasures: exact zeros
        Y[2] = -np.abs(Y[2])                             # already the all-zero codeword: 0 iterations
    with ldpc_amd.Decoder(qc if qc is not None else H) as eng:
        for prec in (32, 64):
            for sm in (0, 1):
                hard, iters, conv, soft = orc.minsum(Y, 20, prec=prec, sign_mode=sm, threads=8)
                r = eng.decode(Y, 20, prec=prec, sign_mode=sm, want_soft=True)
                tag = (seed, H.shape, prec, sm, eng.launch_info(prec=prec))
                assert np.array_equal(r['iters'], iters), tag
                assert np.array_equal(r['converged'], conv), tag
                assert np.array_equal(r['hard'], hard), tag
                nz = iters > 0                               # 0 iterations: the soft output is the input
                assert np.array_equal(r['soft'][nz], soft[nz]), tag
                assert np.array_equal(r['ones'], hard.sum(1)), tag
