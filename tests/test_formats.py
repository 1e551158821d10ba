"""Data formats either side of the decode path, pinned by tests/golden/formats.npz (generated
from the reference's own common.py by oracle/gen_golden_formats.py)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(GOLDEN, 'formats.npz'))


def test_uncompress_matches_reference(fx):
    from ldpc_amd import codes, common
    H = common.uncompress(fx['observation'])
    assert str(H.dtype) == str(fx['dtype']) and H.shape == (1022, 8176)
    assert int(H.sum()) == int(fx['nnz'])
    assert np.array_equal(H[0], fx['row0']) and np.array_equal(H[511], fx['row511'])
    assert hashlib.sha256(np.packbits(H.astype(np.uint8)).tobytes()).digest() == bytes(fx['sha'])
    qc = codes.observation_to_qc(fx['observation'])
    assert qc.nnz == int(fx['nnz'])
    ptr, hot = fx['block_ptr'], fx['block_hot']
    for b in range(2):
        for j in range(16):
            k = b * 16 + j
            assert sorted(qc.blocks[b][j]) == sorted(hot[ptr[k]:ptr[k + 1]].tolist())
    assert np.array_equal(qc.to_dense(), H)
    assert codes.detect_qc(H).key() == qc.key()


def test_compress_is_the_environments_view_of_rows_0_and_511(fx):
    from ldpc_amd import codes, common
    H = codes.near_earth().to_dense()
    obs = common.compress(H)
    assert obs.dtype == np.uint8 and obs.size == 2048
    a, b = common.observation_to_first_columns(obs)
    assert np.array_equal(a, H[0]) and np.array_equal(b, H[511])
    # the reference's pair is not an inverse pair (first row in, first column out): the round
    # trip transposes every circulant
    T = common.uncompress(obs)
    Z = 511
    for blk in ((0, 0), (1, 7)):
        sub = H[blk[0] * Z:(blk[0] + 1) * Z, blk[1] * Z:(blk[1] + 1) * Z]
        assert np.array_equal(T[blk[0] * Z:(blk[0] + 1) * Z, blk[1] * Z:(blk[1] + 1) * Z], sub.T)


def test_recursive_linear_fit_matches_reference(fx):
    from ldpc_amd import common
    snr, ber, p, trendP, itr = common.recursiveLinearFit(fx['fit_x'], fx['fit_y'])
    assert np.array_equal(np.asarray(p, np.float64), fx['fit_p'])
    assert len(snr) == int(fx['fit_kept']) and itr == int(fx['fit_itr'])


@pytest.mark.gpu
def test_gym_evaluation_step(fx):
    """One environment step on the engine: the observation held by the reference's own test
    (a mutated code with 34,237 edges) and NearEarth itself; the reward follows from the
    per-frame scatter exactly as in LdpcEnv.calcReward."""
    from ldpc_amd import codes, common, gym_eval
    ev = gym_eval.CodeEvaluator(seed=11, numberOfTransmissions=12)
    reward, stats = ev.step(fx['observation'])
    sc_snr, sc_ber, sc_itr, snr_axis, _, ber_data, _ = stats.getStatsV2()
    assert len(sc_snr) == 36 and list(snr_axis) == [np.float32(3.0), np.float32(3.2), np.float32(3.4)]
    _, _, p1, _, _ = common.recursiveLinearFit(sc_snr, sc_ber)
    integ = (np.poly1d([1]) - p1).integ()
    assert reward == integ(np.float32(3.4)) - integ(np.float32(3.0))
    assert 0.0 < reward <= 0.4 + 1e-6
    # same seeds, NearEarth: decodes better than the mutated code => larger area
    ev2 = gym_eval.CodeEvaluator(seed=11, numberOfTransmissions=12)
    reward_ne, _ = ev2.step(codes.near_earth().to_dense())
    assert reward_ne >= reward - 1e-3
    # an evaluator on two logical devices splits the frames and keeps the row count
    ev3 = gym_eval.CodeEvaluator(seed=11, numberOfTransmissions=12, numberOfDevices=2)
    _, stats3 = ev3.step(codes.near_earth())
    assert len(stats3.getStatsV2()[0]) == 36
