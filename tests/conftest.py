import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# the library honours its kernel-forcing / schedule switches (LDPC_HIP_NO_PLANES, ..._PRECHECK_T, ...)
# only when this is set at load time; the parity tests compare those paths with each other
os.environ.setdefault('LDPC_HIP_TEST_HOOKS', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (run with -m gpu on the MI355X box)')


def mutation_rows(kind):
    """Hot positions of the gym-style mutated codes of the golden set (same recipe as
    oracle/gen_golden.py: RandomState(20211226))."""
    prng = np.random.RandomState(20211226)
    if kind == 'w3':
        return [(0, 5, sorted(prng.choice(511, 3, replace=False).tolist()))]
    if kind == 'w7':
        prng.choice(511, 3, replace=False)
        return [(1, 9, sorted(prng.choice(511, 7, replace=False).tolist())),
                (0, 12, sorted(prng.choice(511, 1, replace=False).tolist()))]
    if kind == 'many':
        # the state after a long episode: 24 actions, each a circulant of 1...7 hot bits somewhere (the
        # environment keeps what earlier actions installed, ldpc_env.py:293-317); check degrees reach 70 and more
        prng = np.random.RandomState(20211227)
        rows = []
        for _ in range(24):
            b, j, w = int(prng.randint(2)), int(prng.randint(16)), int(prng.randint(1, 8))
            rows.append((b, j, sorted(prng.choice(511, w, replace=False).tolist())))
        for j in range(0, 16, 2):                      # and a block-row filled with heavy circulants
            rows.append((0, j, sorted(prng.choice(511, 7, replace=False).tolist())))
        rows.append((1, 5, sorted(prng.choice(511, 5, replace=False).tolist())))
        return rows
    raise ValueError(kind)


def build_code(name):
    """Code of a golden file as (QCCode or None, dense H)."""
    from ldpc_amd import codes
    if name == 'nearEarth':
        qc = codes.near_earth()
    elif name.startswith('nearEarth_zero_'):
        _, _, i, j = name.split('_')
        qc = codes.near_earth_zeroed(int(i), int(j))
    elif name.startswith('nearEarth_mut_'):
        qc = codes.near_earth()
        for (b, j, hot) in mutation_rows(name.split('_')[2]):
            qc = qc.with_block(b, j, hot)
    elif name == 'wifi_5_6':
        qc = codes.wifi_1944_r56()
    elif name == 'rand_24_48':
        return None, np.load(os.path.join(GOLDEN, name + '.npz'))['H'].astype(np.int32)
    else:
        raise ValueError(name)
    return qc, qc.to_dense()


def make_llrs(seed, snr_db, n, n_frames):
    """Fixture rule of the golden set (see oracle/gen_golden.py)."""
    prng = np.random.RandomState(seed)
    sigma = np.sqrt(0.5 / (10 ** (snr_db / 10)))
    out = np.empty((n_frames, n), dtype=np.float32)
    for k in range(n_frames):
        out[k] = np.float32(-1) + np.float32(prng.normal(0, sigma, n))
    return out


def golden_points(name):
    """Yield dicts(seed, snr, special, llr float32 [F,n], iters, conv, hard (packed little), sha, head)."""
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    n = int(g['shape'][1])
    for p in range(int(g['n_points'])):
        special = str(g[f'p{p}_special'])
        iters = g[f'p{p}_iters']
        F = len(iters)
        if special == 'clean':
            Y = np.full((1, n), -1.0, np.float32)
        elif special == 'flip0':
            Y = np.full((1, n), -1.0, np.float32)
            Y[0, 0] = 1.0
        elif special == 'zeros':
            Y = np.zeros((1, n), np.float32)
        else:
            Y = make_llrs(int(g[f'p{p}_seed']), float(g[f'p{p}_snr']), n, F)
        assert np.array_equal(Y[0], g[f'p{p}_llr0']), 'host RNG path drifted from the fixture'
        yield dict(seed=int(g[f'p{p}_seed']), snr=float(g[f'p{p}_snr']), special=special, llr=Y,
                   iters=iters, conv=g[f'p{p}_conv'], hard=g[f'p{p}_hard'],
                   sha=g[f'p{p}_soft_sha256'], head=g[f'p{p}_soft_head'],
                   max_iters=int(g['max_iters']))


ALL_CODES = ['nearEarth', 'nearEarth_zero_0_0', 'nearEarth_zero_0_7', 'nearEarth_zero_1_3',
             'nearEarth_zero_1_15', 'nearEarth_mut_w3', 'nearEarth_mut_w7', 'nearEarth_mut_many', 'wifi_5_6', 'rand_24_48']


@pytest.fixture(scope='session')
def oracle_mod():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle as om
    om.build()
    return om
