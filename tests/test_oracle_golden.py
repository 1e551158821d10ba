"""CPU suite: the oracle (C restatement of ldpc.py's decoder) against the golden
vectors produced by the reference's own source (tests/golden, oracle/gen_golden.py)."""
import hashlib

import numpy as np
import pytest

from conftest import ALL_CODES, build_code, golden_points


@pytest.mark.parametrize('name', ALL_CODES)
def test_oracle_f64_matches_reference_bit_for_bit(oracle_mod, name):
    _, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    n_frames = 0
    for pt in golden_points(name):
        hard, iters, conv, soft = orc.minsum(pt['llr'].astype(np.float64), pt['max_iters'], prec=64, threads=8)
        assert np.array_equal(iters, pt['iters']), (name, pt['seed'])
        assert np.array_equal(conv, pt['conv'])
        assert np.array_equal(np.packbits(hard, axis=1, bitorder='little'), pt['hard'])
        for k in range(len(iters)):
            assert hashlib.sha256(soft[k].tobytes()).digest() == bytes(pt['sha'][k]), (name, pt['seed'], k)
            assert np.array_equal(soft[k, :8], pt['head'][k])
        n_frames += len(iters)
    assert n_frames > 0


def test_oracle_reused_decoder_matches_reference(oracle_mod):
    """Frames decoded in a row by ONE reference decoder object (stale check-node
    messages, ldpc.py:270-274): tests/golden/chained_decoder.npz."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'chained_decoder.npz'))
    for name in ('nearEarth', 'wifi_5_6', 'rand_24_48'):
        _, H = build_code(name)
        orc = oracle_mod.Oracle(H)
        state = orc.new_state(64)
        Y = g[f'{name}_llr'].astype(np.float64)
        half = len(Y) // 2
        res = [orc.minsum(Y[:half], 50, prec=64, state=state), orc.minsum(Y[half:], 50, prec=64, state=state)]
        iters = np.concatenate([r[1] for r in res])
        hard = np.concatenate([r[0] for r in res])
        soft = np.concatenate([r[3] for r in res])
        assert np.array_equal(iters, g[f'{name}_iters']), name
        assert np.array_equal(np.packbits(hard, axis=1, bitorder='little'), g[f'{name}_hard'])
        for k in range(len(Y)):
            assert hashlib.sha256(soft[k].tobytes()).digest() == bytes(g[f'{name}_soft_sha256'][k])
        # and it differs from independent decoding (otherwise the fixture pins nothing)
        fresh = orc.minsum(Y, 50, prec=64)
        assert name == 'wifi_5_6' or not np.array_equal(fresh[1], iters) or not np.array_equal(fresh[3], soft)


def test_numpy_restatement_agrees_with_c_oracle(oracle_mod):
    _, H = build_code('rand_24_48')
    orc = oracle_mod.Oracle(H)
    for pt in golden_points('rand_24_48'):
        Y = pt['llr'][:8]
        for dtype, prec in ((np.float64, 64), (np.float32, 32)):
            hard, iters, conv, soft = orc.minsum(Y.astype(dtype), 50, prec=prec)
            for k in range(len(Y)):
                h2, it2, ok2, s2 = oracle_mod.minsum_numpy(H, Y[k], 50, dtype=dtype)
                assert it2 == iters[k] and bool(ok2) == bool(conv[k])
                assert np.array_equal(h2, hard[k])
                assert np.array_equal(s2, soft[k])


def test_numpy_restatement_on_one_wifi_frame(oracle_mod):
    _, H = build_code('wifi_5_6')
    pts = list(golden_points('wifi_5_6'))
    pt = [p for p in pts if p['snr'] == 8.0][0]
    k = int(np.argmax(pt['iters']))
    h2, it2, ok2, s2 = oracle_mod.minsum_numpy(H, pt['llr'][k].astype(np.float64), 50)
    assert it2 == pt['iters'][k] and bool(ok2) == bool(pt['conv'][k])
    assert hashlib.sha256(np.ascontiguousarray(s2).tobytes()).digest() == bytes(pt['sha'][k])


def test_oracle_f32_is_deterministic_and_close_to_f64(oracle_mod):
    _, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    pt = [p for p in golden_points('nearEarth') if p['snr'] == 4.5 and not p['special']][0]
    h32, it32, c32, s32 = orc.minsum(pt['llr'], 50, prec=32, threads=8)
    h32b, it32b, _, s32b = orc.minsum(pt['llr'], 50, prec=32, threads=3)
    assert np.array_equal(s32, s32b) and np.array_equal(it32, it32b)
    # converged frames at high SNR: float32 and float64 arithmetic agree on decisions
    assert np.array_equal(np.packbits(h32, axis=1, bitorder='little'), pt['hard'])
    assert np.array_equal(it32, pt['iters'])


def test_oracle_edge_cases(oracle_mod):
    _, H = build_code('rand_24_48')
    orc = oracle_mod.Oracle(H)
    n = H.shape[1]
    # zero iteration budget: status of the channel hard decision, no update
    y = np.full((1, n), -1.0)
    y[0, 3] = 0.5
    hard, iters, conv, soft = orc.minsum(y, 0, prec=64)
    assert iters[0] == 0 and conv[0] == 0 and np.array_equal(soft[0], y[0]) and hard[0, 3] == 1
    # exact zeros decide 0 (x > 0 is False): all-zero input is a codeword after 0 iterations
    hard, iters, conv, soft = orc.minsum(np.zeros((1, n)), 50, prec=64)
    assert iters[0] == 0 and conv[0] == 1 and hard.sum() == 0
    # empty batch
    hard, iters, conv, soft = orc.minsum(np.zeros((0, n)), 50, prec=64)
    assert hard.shape == (0, n) and iters.shape == (0,)


def test_oracle_on_the_reference_single_transmission_rows_for_wifi(oracle_mod):
    """evaluateCodeAtSingleTransmission of the reference on its 802.11n matrix (ldpc.py:391-430; fixture from
    oracle/gen_golden.py --only-wifi-rows): the codeword with its first symbol flipped stays undecoded for all 50
    iterations under the reference's sign rule on rows of odd degree.  The oracle says the same."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'berstats_single_transmission_wifi.npz'))
    qc, H = build_code('wifi_5_6')
    y = -np.ones(1944)
    y[0] = 1.0
    hard, iters, conv, _ = oracle_mod.Oracle(H).minsum(y, 50, prec=64)
    for row, status in zip(g['rows'], g['status']):
        assert row[4] == 1 and row[5] == hard.sum() and row[6] == iters[0] and row[7] == 50
        assert str(status) == ('Codeword' if conv[0] else 'Not a codeword')


def test_oracle_on_the_reference_evaluateCode_rows(oracle_mod):
    """ldpc.evaluateCode of the reference itself (ldpc.py:350-388; fixture from oracle/gen_golden.py
    --only-evaluate-code-rows): nine frames on ONE decoder object, the first three failing and leaving their
    messages behind.  The oracle with carried state reproduces error counts, iteration counts and status."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'berstats_evaluate_code.npz'))
    qc, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    prng = np.random.RandomState(7134066)            # the module's generator, ldpc.py:24
    state = orc.new_state(64)
    for row, status in zip(g['rows'], g['status']):
        sigma = np.sqrt(0.5 / 10 ** (row[0] / 10))
        noise = prng.normal(0, sigma, 8176)
        assert row[2] == sigma and row[3] == np.sqrt(np.sum(noise ** 2) / 8176)
        hard, iters, conv, _ = orc.minsum(-1.0 + noise, 50, prec=64, state=state)
        assert (row[5], row[6], row[7]) == (hard.sum(), iters[0], 50)
        assert str(status) == ('Codeword' if conv[0] else 'Not a codeword')
    assert g['rows'][:3, 5].min() > 0 and (g['rows'][3:, 5] == 0).all()
