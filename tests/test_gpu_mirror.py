"""GPU suite (-m gpu): the Python mirror of the reference's entry points."""
import hashlib
import importlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, build_code, golden_points

pytestmark = pytest.mark.gpu


def test_ldpcDecoder_decoderMainLoop_matches_reference():
    from ldpc_amd import ldpc, fileHandler
    H = fileHandler.nearEarthParity()                       # dense, as the reference passes it
    pts = [p for p in golden_points('nearEarth')]
    for pt in pts:
        if pt['snr'] not in (0.0, 3.4, 4.5):
            continue
        for k in range(min(3, len(pt['iters']))):
            y = pt['llr'][k].astype(np.float64)
            dec = ldpc.ldpcDecoder(H)                       # the golden frames used a fresh decoder each
            status, binv, soft, it = dec.decoderMainLoop(y, 50)
            assert it == pt['iters'][k]
            assert status == ('Codeword' if pt['conv'][k] else 'Not a codeword')
            assert binv.dtype == np.int64 and soft.dtype == np.float64
            assert np.array_equal(np.packbits(binv.astype(np.uint8), bitorder='little'), pt['hard'][k])
            assert hashlib.sha256(np.ascontiguousarray(soft).tobytes()).digest() == bytes(pt['sha'][k])
            st2, b2 = dec.isCodeword(soft)
            assert st2 == status and np.array_equal(b2, binv)


def test_reused_decoder_object_matches_reference():
    """One decoder object, several frames: check-node messages carry over exactly as
    in the reference (tests/golden/chained_decoder.npz)."""
    from ldpc_amd import ldpc
    g = np.load(os.path.join(GOLDEN, 'chained_decoder.npz'))
    for name in ('nearEarth', 'wifi_5_6', 'rand_24_48'):
        qc, H = build_code(name)
        dec = ldpc.ldpcDecoder(H)
        Y = g[f'{name}_llr']
        for k in range(len(Y)):
            status, binv, soft, it = dec.decoderMainLoop(Y[k].astype(np.float64), 50)
            assert it == g[f'{name}_iters'][k], (name, k)
            assert (status == 'Codeword') == bool(g[f'{name}_conv'][k])
            assert np.array_equal(np.packbits(binv.astype(np.uint8), bitorder='little'), g[f'{name}_hard'][k])
            assert hashlib.sha256(np.ascontiguousarray(soft).tobytes()).digest() == bytes(g[f'{name}_soft_sha256'][k])
        # the same frames as one chained batch on a new object
        dec2 = ldpc.ldpcDecoder(qc if qc is not None else H)
        r = dec2.decodeBatch(Y.astype(np.float64), 50, wantSoft=True)
        assert np.array_equal(r['iters'], g[f'{name}_iters'])
        assert np.array_equal(r['hard_packed'], g[f'{name}_hard'])


def test_evaluateCodeAtSingleTransmission_rows_match_reference():
    """Same call as oracle/gen_golden.py made on the reference (fresh module state)."""
    from ldpc_amd import ldpc, fileHandler
    importlib.reload(ldpc)                                   # module PRNG back to its seed (ldpc.py:24)
    g = np.load(os.path.join(GOLDEN, 'berstats_single_transmission.npz'))
    bs = ldpc.evaluateCodeAtSingleTransmission(460101, [3.0, 3.2], 7154, 8176, 50, fileHandler.nearEarthParity())
    rows = bs.getRawStats()
    got = np.array([[float(x) for x in r[:8]] for r in rows])
    assert np.array_equal(got, g['rows'])
    assert [r[8] for r in rows] == [str(s) for s in g['status']]
    with pytest.raises(AssertionError):
        ldpc.evaluateCodeAtSingleTransmission(0, [3.0], 7154, 8176, 50, fileHandler.nearEarthParity())
    with pytest.raises(AssertionError):
        ldpc.evaluateCodeAtSingleTransmission(5, 3.0, 7154, 8176, 50, fileHandler.nearEarthParity())


def test_evaluateCode_and_multiprocessing_mirror(oracle_mod):
    from ldpc_amd import ldpc, codes
    importlib.reload(ldpc)
    H = codes.near_earth().to_dense()
    orc = oracle_mod.Oracle(H)
    bs = ldpc.evaluateCode(3, 17, [3.4, 4.0], 7154, 8176, 50, H)
    rows = bs.getRawStats()
    assert len(rows) == 6
    # regenerate the frames the way the reference draws them (module PRNG, float64 noise)
    prng = np.random.RandomState(7134066)
    k = 0
    state = orc.new_state(64)                # evaluateCode re-uses one decoder (ldpc.py:356, 377)
    for snr in (3.4, 4.0):
        sigma = np.sqrt(0.5 / 10 ** (snr / 10))
        for _ in range(3):
            noise = prng.normal(0, sigma, 8176)
            y = -1.0 + noise
            hard, iters, conv, _ = orc.minsum(y, 50, prec=64, state=state)
            r = rows[k]
            assert r[0] == snr and r[2] == sigma and r[3] == np.sqrt(np.sum(noise ** 2) / 8176)
            assert r[4] == 0 and r[5] == hard.sum() and r[6] == iters[0]
            assert r[8] == ('Codeword' if conv[0] else 'Not a codeword')
            k += 1
    # the same call the reference itself answered (oracle/gen_golden.py --only-evaluate-code-rows): nine frames on one
    # decoder, the first three fail and leave their messages behind -- all nine columns of every row
    importlib.reload(ldpc)
    g = np.load(os.path.join(GOLDEN, 'berstats_evaluate_code.npz'))
    rows9 = ldpc.evaluateCode(3, 17, [2.6, 3.4, 4.0], 7154, 8176, 50, H).getRawStats()
    assert np.array_equal(np.array([[float(x) for x in r[:8]] for r in rows9]), g['rows'])
    assert [r[8] for r in rows9] == [str(x) for x in g['status']]
    bs2 = ldpc.testCodeUsingMultiprocessing(460101, [3.0, 3.2], 7154, 8176, 50, 4, H)
    rows2 = bs2.getRawStats()
    assert len(rows2) == 8 and rows2 == sorted(rows2)
    assert all(r[4] == 1 and r[5] == 0 and r[8] == 'Codeword' for r in rows2)   # flipped first symbol is corrected
    # testWifi (ldpc.py:504-525, with the evaluation call its NearEarth twin makes): 4 points x 3 frames
    rows3 = ldpc.testWifi(3).getRawStats()
    assert len(rows3) == 12 and rows3 == sorted(rows3)
    assert [r[0] for r in rows3[::3]] == [3.0, 3.2, 3.4, 3.6]
    # every row decodes the same input, the codeword with its first symbol flipped (ldpc.py:417-418); under the
    # reference's rule on rows of odd degree (ldpc.py:135-141) it is NOT corrected: the fixture holds what the
    # reference itself returns for this matrix (oracle/gen_golden.py --only-wifi-rows)
    g = np.load(os.path.join(GOLDEN, 'berstats_single_transmission_wifi.npz'))
    assert [r[0] for r in rows3[::3]] == g['rows'][:, 0].tolist()
    for r in rows3:
        assert [float(x) for x in r[4:8]] == g['rows'][0, 4:8].tolist() == [1.0, 1.0, 50.0, 50.0]
        assert r[8] == str(g['status'][0]) == 'Not a codeword'
    # and the single-transmission call itself, all columns (module PRNG back to its seed, ldpc.py:24)
    importlib.reload(ldpc)
    from ldpc_amd import wifiMatrices
    bs = ldpc.evaluateCodeAtSingleTransmission(460101, [3.0, 3.2, 3.4, 3.6], 1620, 1944, 50,
                                               wifiMatrices.getWifiParityMatrix())
    rows4 = bs.getRawStats()
    assert np.array_equal(np.array([[float(x) for x in r[:8]] for r in rows4]), g['rows'])
    assert [r[8] for r in rows4] == [str(x) for x in g['status']]


def test_evaluateCodeCuda_matches_oracle_on_host_seeded_frames(oracle_mod):
    from ldpc_amd import ldpcHIP, codes
    H = codes.near_earth().to_dense().astype(np.int32)
    orc = oracle_mod.Oracle(H)
    snrs = [3.2, 3.6]
    nTx = 6
    bs = ldpcHIP.evaluateCodeCuda(460101, snrs, 50, H, nTx)
    rows = bs.getRawStats()
    assert len(rows) == 12
    prng = np.random.RandomState(460101)
    k = 0
    for snr in snrs:
        sigma = np.sqrt(0.5 / 10 ** (snr / 10))
        for _ in range(nTx):
            noise = np.float32(prng.normal(0, sigma, 8176))
            y = np.float32(-1) + noise
            hard, iters, conv, _ = orc.minsum(y, 50, prec=32)
            r = rows[k]
            assert r[0] == snr and r[2] == sigma
            assert r[4] == np.count_nonzero(y > 0) and r[5] == hard.sum() and r[6] == iters[0] and r[7] == 50
            k += 1
    # wrapper: nTx // devices frames per device, device d seeded with seeds[d]
    seeds = [460101, 7]
    bw = ldpcHIP.evaluateCodeCudaWrapper(seeds, snrs, 50, H, 12, 'None', 2)
    rw = bw.getRawStats()
    assert len(rw) == 2 * 2 * 6
    assert [tuple(r[:8]) for r in rw[:12]] == [tuple(r[:8]) for r in rows]
    st, status = ldpcHIP.testNearEarth(8)
    assert status == 'OK'
    # evaluateMatrixAndEpsilon (ldpcCUDA.py:677-828): one frame, the caller's noise
    eps = np.float32(np.random.RandomState(99).normal(0, np.sqrt(0.5 / 10 ** 0.32), 8176))
    y = np.float32(-1) + eps
    hard, iters, conv, _ = orc.minsum(y, 50, prec=32)
    unc, dec, it, seconds = ldpcHIP.evaluateMatrixAndEpsilon(H, eps)
    assert (unc, dec, it) == (np.count_nonzero(y > 0), hard.sum(), iters[0]) and unc > 0 and seconds > 0
    # ldpcCUDA.testConcurrentFutures (ldpcCUDA.py:871-888): 15 frames per point and device, seeds from RandomState(7134066)
    two = ldpcHIP.testConcurrentFutures(2).getRawStats()
    dseeds = np.random.RandomState(7134066).randint(0, 2 ** 31 - 1, 2, dtype=np.int64)
    roi = [3.0, 3.2, 3.4, 3.6]
    assert len(two) == 2 * 4 * 15
    for d in range(2):
        one = ldpcHIP.evaluateCodeCuda(dseeds[d], roi, 50, H, 15).getRawStats()
        assert [tuple(r[:8]) for r in two[60 * d:60 * (d + 1)]] == [tuple(r[:8]) for r in one]


def test_wifiCUDA_mirror(oracle_mod):
    """wifiCUDA.evaluateCodeCuda / testWifi (wifiCUDA.py:496-682) on the 802.11n matrix: rows equal to the
    float32 oracle's on the frames the reference would draw."""
    from ldpc_amd import wifiHIP, wifiMatrices
    H = wifiMatrices.getWifiParityMatrix()
    orc = oracle_mod.Oracle(np.asarray(H))
    rows = wifiHIP.evaluateCodeCuda(460101, [1.0, 3.6], 50, H, 5).getRawStats()        # 1 dB: frames that use the whole budget
    prng = np.random.RandomState(460101)
    k = 0
    for snr in (1.0, 3.6):
        sigma = np.sqrt(0.5 / 10 ** (snr / 10))
        for _ in range(5):
            y = np.float32(-1) + np.float32(prng.normal(0, sigma, 1944))
            hard, iters, conv, _ = orc.minsum(y, 40, prec=32)        # 40 whatever the argument says, wifiCUDA.py:535
            r = rows[k]
            assert (r[0], r[4], r[5], r[6], r[7]) == (snr, np.count_nonzero(y > 0), hard.sum(), iters[0], 50)
            k += 1
    assert max(r[6] for r in rows) == 40
    st, status = wifiHIP.testWifi(6)
    assert len(st.getRawStats()) == 24 and status in ('OK', 'WIFI problem')
    a, b, c, d = st.getStats(1944)
    assert status == ('OK' if c[2] == 0 and c[3] == 0 else 'WIFI problem')


def test_device_rng_evaluation_statistics():
    from ldpc_amd import ldpcHIP, codes
    bs = ldpcHIP.evaluateCodeCuda(5, [3.0, 3.4, 3.8], 50, codes.near_earth(), 200, deviceRng=True)
    scSnr, scBer, scItr, snrAxis, avgSnr, ber, _ = bs.getStatsV2()
    assert len(scSnr) == 600 and np.allclose(avgSnr, snrAxis, atol=0.02)
    # anchors embedded in the reference (common.py:113-114; rewardInvestigation.ipynb:64):
    # BER ~0.02 at 3.0 dB, dropping to 0 by 3.4-3.6 dB
    assert 0.012 < ber[0] < 0.03 and ber[1] < 0.004 and ber[2] == 0


def test_reward_constants_of_the_reference():
    """postProcessing.py:18-19 keeps the reward of NearEarth as constants: 0.3965108 over 3.0 ... 3.4 dB and
    0.7958452 over 3.0 ... 3.8 dB -- the area between 1 and the line fitted through the per-frame (SNR, BER)
    scatter (the fit before common.recursiveLinearFit replaced it, ldpc_env.py:326-339).  On 3000 frames per point of
    device noise the whole path -- channel, decoder, statistics, fit -- lands within 5e-4 of both."""
    from ldpc_amd import codes, ldpcHIP
    for roi, want in (([3.0, 3.2, 3.4], 0.3965108116285836), ([3.0, 3.2, 3.4, 3.6, 3.8], 0.7958451612664468)):
        st = ldpcHIP.evaluateCodeCuda(5, roi, 50, codes.near_earth(), 3000, deviceRng=True)
        scSnr, scBer = st.getStatsV2()[:2]
        integ = (np.poly1d([1]) - np.poly1d(np.polyfit(scSnr, scBer, 1))).integ()
        assert abs((integ(roi[-1]) - integ(roi[0])) - want) < 5e-4


def test_near_earth_quality_anchor_of_the_reference_notebook():
    """rewardInvestigation.ipynb (cell 3, lines 62-94 of the file) holds the output of
    ldpcCUDA.testNearEarth(10) -- seed 460101, 3.0/3.2/3.4 dB, 10 frames each, float32:
    berDecoded [0.02117172 0.01221869 0.], fitted line [-0.05026939 0.17176406], reward
    0.39563919692811345.  Same seed and noise here; frames that do not converge are chaotic
    (the reference's kernels add in a different order and test the syndrome every 6th
    iteration), so their error counts agree statistically, the rest exactly."""
    from ldpc_amd import codes, ldpcHIP
    roi = [3.0, 3.2, 3.4]
    st = ldpcHIP.evaluateCodeCuda(460101, roi, 50, codes.near_earth().to_dense(), 10)
    scatterSnr, scatterBer, scatterItr, snrAxis, _, berData, _ = st.getStatsV2()
    assert berData[2] == 0.0                                            # 3.4 dB: every frame decodes
    assert abs(berData[0] - 0.02117172) < 0.003 and abs(berData[1] - 0.01221869) < 0.003
    p = np.polyfit(scatterSnr, scatterBer, 1)
    assert abs(p[0] - (-0.05026939)) < 0.005 and abs(p[1] - 0.17176406) < 0.015
    integ = (np.poly1d([1]) - np.poly1d(p)).integ()
    assert abs((integ(roi[-1]) - integ(roi[0])) - 0.39563919692811345) < 3e-4


@pytest.mark.gpu
def test_device_noise_curve_within_monte_carlo_interval_of_the_oracle(oracle_mod):
    """FER / BER of the sweep driver (noise generated on the device, float32 kernels) against the CPU
    oracle decoding an INDEPENDENT sample drawn on the host the way the reference draws it
    (ldpcCUDA.py:509-517): the two estimates of every point must agree within their joint Monte-Carlo
    spread (3 sigma of the difference, frame as the sampling unit), and the oracle's estimate must
    lie inside the 99.7 % interval the confidence-interval helpers give for the device's."""
    import ldpc_amd
    from ldpc_amd import codes, common
    from conftest import make_llrs
    qc = codes.near_earth()
    orc = oracle_mod.Oracle(qc.to_dense())
    snrs, Fg, Fc = [3.0, 3.2, 3.4], 20000, 1500
    with ldpc_amd.Decoder(qc) as eng:
        cnt, pf = eng.eval_awgn(snrs, Fg, seed=20261111, per_frame=True)
    for k, snr in enumerate(snrs):
        hard, iters, conv, _ = orc.minsum(make_llrs(77 + k, snr, qc.n, Fc), 50, prec=32, want_soft=False, threads=16)
        e_c = hard.sum(1).astype(np.float64)
        e_g = pf[k, :, 2].astype(np.float64)
        fer_g, fer_c = (e_g > 0).mean(), (e_c > 0).mean()
        sd = np.sqrt(fer_g * (1 - fer_g) / Fg + fer_c * (1 - fer_c) / Fc)
        assert abs(fer_g - fer_c) <= 3 * sd + 1e-12, (snr, fer_g, fer_c, sd)
        ber_g, ber_c = e_g.mean() / qc.n, e_c.mean() / qc.n
        sdb = np.sqrt(e_g.var(ddof=1) / Fg + e_c.var(ddof=1) / Fc) / qc.n
        assert abs(ber_g - ber_c) <= 3 * sdb + 1e-12, (snr, ber_g, ber_c, sdb)
        it_g, it_c = pf[k, :, 3].astype(np.float64), iters.astype(np.float64)
        assert abs(it_g.mean() - it_c.mean()) <= 3 * np.sqrt(it_g.var(ddof=1) / Fg + it_c.var(ddof=1) / Fc)
        lo, hi = common.wilson_interval(int((e_g > 0).sum()), Fg, z=3.0)
        assert lo - 3 * np.sqrt(fer_c * (1 - fer_c) / Fc) <= fer_c <= hi + 3 * np.sqrt(fer_c * (1 - fer_c) / Fc)
        b, blo, bhi = common.ber_interval(e_g, qc.n)
        assert blo <= b <= bhi and b == ber_g
    assert (pf[0, :, 2] > 0).mean() > 0.5 > (pf[2, :, 2] > 0).mean()       # the points straddle the waterfall


@pytest.mark.gpu
def test_library_communicator_single_rank():
    """ldpc_comm_* / ldpc_allreduce_counters over RCCL with one rank: the id round trip, the collective
    (a sum over one rank leaves the counters as they are, whatever their number) and teardown.  More ranks need
    more cards than this test's box has; RCCL refuses two ranks on one GPU."""
    import ldpc_amd
    from ldpc_amd import codes
    ident = ldpc_amd.comm_unique_id()
    assert len(ident) == 128 and any(ident)
    with ldpc_amd.Comm(0, 0, 1, ident) as comm, ldpc_amd.Decoder(codes.near_earth()) as eng:
        cnt, _ = eng.eval_awgn([3.0, 3.5], 200, seed=4)
        total = comm.allreduce(cnt)
        assert total.shape == cnt.shape and np.array_equal(total, cnt)
        big = np.arange(1000, dtype=np.uint64) * np.uint64(1 << 40)
        assert np.array_equal(comm.allreduce(big), big)
    with pytest.raises(ldpc_amd.LdpcHipError):
        ldpc_amd.Comm(0, 1, 1, ident)


@pytest.mark.gpu
def test_library_communicator_two_ranks():
    """Two processes, one GPU each, sum their shares of a sweep through ldpc_allreduce_counters
    (tools/comm_two_ranks.py): the sum equals one rank decoding all the frames.  Needs two cards."""
    import subprocess
    import sys
    import ldpc_amd
    if ldpc_amd.device_count() < 2:
        pytest.skip('one rank per GPU: needs two cards')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'comm_two_ranks.py'), '2'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'OK' in r.stdout, r.stdout + r.stderr
