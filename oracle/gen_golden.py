#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz by executing the
reference's own decoder source (read-only, from /root/reference) in the build
container.  Never runs on the GPU box; only its OUTPUT files are committed.

How the reference is made runnable here (nothing is copied into this repo):
  * `numba` is not installed, so a stand-in module turns @jit / @jitclass into
    identity decorators; the node classes' bodies are plain NumPy, so the
    reference's float64 arithmetic runs verbatim.
  * /root/reference/ldpc.py:438 has a run of blanks inside an identifier (a
    SyntaxError); the source text is read, that one identifier is repaired in
    memory, and the result is exec'd into a fresh module.

Fixture rule (so any machine can regenerate the INPUTS without the reference):
    prng  = numpy.random.RandomState(seed)          # legacy, frozen stream
    sigma = sqrt(0.5 / 10**(snr_db/10))             # ldpc.py:53-55
    frame k (k = 0,1,...) : noise = float32(prng.normal(0, sigma, n))
                            y32   = float32(-1) + noise          # ldpcCUDA.py:514-516
    the reference decodes y32.astype(float64) with a FRESH ldpcDecoder(H), 50 iterations max.
Stored per frame: iterations, converged flag, hard decisions (packbits,
bitorder='little'), sha256 of the float64 soft vector bytes, first 8 soft values.

usage: python oracle/gen_golden.py [--quick] [--jobs N]
"""
import argparse
import hashlib
import multiprocessing as mp
import os
import re
import sys
import time
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
MAX_ITERS = 50


def load_reference():
    os.environ.setdefault('MPLBACKEND', 'Agg')

    class _T:
        def __init__(self, dt=None):
            self.dtype = np.dtype(dt) if dt is not None else None

        def __getitem__(self, k):
            return self

        def __call__(self, *a, **k):
            return self.dtype.type(*a) if self.dtype is not None else self

    def _deco(*a, **k):
        return a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f)

    nb = types.ModuleType('numba')
    nb.jit = _deco
    nb.jitclass = lambda spec: (lambda cls: cls)
    nb.int32, nb.float32, nb.float64, nb.int64, nb.boolean = map(
        _T, (np.int32, np.float32, np.float64, np.int64, np.bool_))
    nb.types = types.SimpleNamespace(DictType=lambda *a: _T(), ListType=lambda *a: _T())
    nb.typed = types.SimpleNamespace(Dict=types.SimpleNamespace(empty=lambda *a, **k: {}))
    sys.modules['numba'] = nb
    if REF not in sys.path:
        sys.path.insert(0, REF)
    src = open(os.path.join(REF, 'ldpc.py')).read()
    src = re.sub(r'numberOf\s+Iterations', 'numberOfIterations', src)
    ref = types.ModuleType('ldpc_ref')
    ref.__file__ = os.path.join(REF, 'ldpc.py')
    exec(compile(src, ref.__file__, 'exec'), ref.__dict__)
    return ref


# ---------------------------------------------------------------- code table
def mutation_rows(kind):
    """Gym-style mutations: (block_row, block_col, hot positions of the new first row).
    The action of the gym replaces one 511x511 circulant by circulant(first row)
    (ldpc_env.py:293-317, first-row vector built in ldpc_env.py:169-264)."""
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
    """Dense H built by the REFERENCE's own constructors."""
    import fileHandler
    import wifiMatrices
    from scipy.linalg import circulant
    if name.startswith('nearEarth'):
        H = fileHandler.readMatrixFromFile(os.path.join(REF, 'codeMatrices', 'nearEarthParity.txt'),
                                           1022, 8176, 511, True, False, False)
        H = np.array(H, dtype=np.int32)
        if name.startswith('nearEarth_zero_'):
            # testMatricesGeneratorScript.py:27-31
            _, _, i, j = name.split('_')
            i, j = int(i), int(j)
            H[i * 511:(i + 1) * 511, j * 511:(j + 1) * 511] = 0
        elif name.startswith('nearEarth_mut_'):
            for (bi, bj, hot) in mutation_rows(name.split('_')[2]):
                first = np.zeros(511, dtype=np.int32)
                first[hot] = 1
                # ldpc_env.py:208 installs circulant(firstRow).T: row i has ones at (i + loc) mod Z
                H[bi * 511:(bi + 1) * 511, bj * 511:(bj + 1) * 511] = circulant(first).T
        return H
    if name == 'wifi_5_6':
        return np.array(wifiMatrices.getWifiParityMatrix(), dtype=np.int32)
    if name == 'rand_24_48':
        prng = np.random.RandomState(99)
        m, n = 24, 48
        H = np.zeros((m, n), dtype=np.int32)
        for c in range(n):
            H[prng.choice(m, 3, replace=False), c] = 1
        for r in range(m):
            while H[r].sum() < 2:
                H[r, prng.randint(n)] = 1
        return H
    raise ValueError(name)


def make_llrs(seed, snr_db, n, n_frames):
    prng = np.random.RandomState(seed)
    sigma = np.sqrt(0.5 / (10 ** (snr_db / 10)))
    out = np.empty((n_frames, n), dtype=np.float32)
    for k in range(n_frames):
        noise = np.float32(prng.normal(0, sigma, n))
        out[k] = np.float32(-1) + noise
    return out


_REF = None
_CODES = {}


def run_job(job):
    global _REF
    name, seed, snr, n_frames, special = job
    if _REF is None:
        _REF = load_reference()
    if name not in _CODES:
        _CODES[name] = build_code(name)
    H = _CODES[name]
    m, n = H.shape
    if special == 'clean':          # noise-free all-zero codeword -> 0 iterations
        Y = np.full((1, n), -1.0, dtype=np.float32)
    elif special == 'flip0':        # ldpc.py:417-418: clean codeword with bit 0 flipped
        Y = np.full((1, n), -1.0, dtype=np.float32)
        Y[0, 0] = 1.0
    elif special == 'zeros':        # all-zero LLRs: hard decision 0 everywhere (x > 0 is False)
        Y = np.zeros((1, n), dtype=np.float32)
    else:
        Y = make_llrs(seed, snr, n, n_frames)
    F = Y.shape[0]
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, np.uint8)
    hard = np.zeros((F, (n + 7) // 8), np.uint8)
    sha = np.zeros((F, 32), np.uint8)
    head = np.zeros((F, 8), np.float64)
    t0 = time.time()
    for k in range(F):
        dec = _REF.ldpcDecoder(H)                      # fresh decoder per frame
        status, binv, soft, it = dec.decoderMainLoop(Y[k].astype(np.float64), MAX_ITERS)
        soft = np.ascontiguousarray(soft, dtype=np.float64)
        iters[k] = it
        conv[k] = status == 'Codeword'
        hard[k] = np.packbits(np.asarray(binv, dtype=np.uint8), bitorder='little')
        sha[k] = np.frombuffer(hashlib.sha256(soft.tobytes()).digest(), dtype=np.uint8)
        head[k] = soft[:8]
    return dict(name=name, seed=seed, snr=snr, special=special or '', iters=iters, conv=conv,
                hard=hard, sha=sha, head=head, seconds=time.time() - t0,
                llr0=Y[0].copy())


def gen_chained():
    """ONE reference decoder object decoding several frames in a row, as
    ldpc.evaluateCode does (ldpc.py:356, 377): the check nodes keep the messages
    of the previous frame (decoderSet resets only variable nodes, ldpc.py:270-274)."""
    ref = load_reference()
    out = {}
    for name, seed, snr, nfr in (('nearEarth', 460201, 3.4, 6), ('wifi_5_6', 7101, 7.0, 8),
                                 ('rand_24_48', 8101, 4.0, 16)):
        H = build_code(name)
        Y = make_llrs(seed, snr, H.shape[1], nfr)
        if name == 'nearEarth':
            # the frame evaluateCodeAtSingleTransmission decodes at every SNR point (ldpc.py:417-418)
            Y[1, :] = -1.0
            Y[1, 0] = 1.0
            Y[2] = Y[1]
        dec = ref.ldpcDecoder(H)
        iters, conv, hard, sha = [], [], [], []
        for k in range(nfr):
            status, binv, soft, it = dec.decoderMainLoop(Y[k].astype(np.float64), MAX_ITERS)
            soft = np.ascontiguousarray(soft, dtype=np.float64)
            iters.append(it)
            conv.append(status == 'Codeword')
            hard.append(np.packbits(np.asarray(binv, dtype=np.uint8), bitorder='little'))
            sha.append(np.frombuffer(hashlib.sha256(soft.tobytes()).digest(), dtype=np.uint8))
        out[f'{name}_llr'] = Y
        out[f'{name}_iters'] = np.array(iters, np.int32)
        out[f'{name}_conv'] = np.array(conv, np.uint8)
        out[f'{name}_hard'] = np.stack(hard)
        out[f'{name}_soft_sha256'] = np.stack(sha)
        print('chained', name, iters, conv)
    np.savez_compressed(os.path.join(OUT, 'chained_decoder.npz'), **out)


def gen_wifi_rows():
    """berStatistics rows of the reference's evaluateCodeAtSingleTransmission (ldpc.py:391-430) on its 802.11n
    matrix, the call ldpc.testWifi (ldpc.py:504-525) would make through testCodeUsingMultiprocessing: the codeword
    with its first symbol flipped (ldpc.py:417-418), decoded under the reference's sign rule on rows of odd degree."""
    ref = load_reference()
    import wifiMatrices
    H = np.array(wifiMatrices.getWifiParityMatrix(), dtype=np.int32)
    bs = ref.evaluateCodeAtSingleTransmission(460101, [3.0, 3.2, 3.4, 3.6], 1620, 1944, 50, H)
    rows = bs.getRawStats()
    num = np.array([[float(x) for x in row[:8]] for row in rows], dtype=np.float64)
    status = np.array([row[8] for row in rows])
    np.savez_compressed(os.path.join(OUT, 'berstats_single_transmission_wifi.npz'), rows=num, status=status)
    print('berStatistics rows (802.11n):', num.tolist(), status.tolist())


def gen_evaluate_code_rows():
    """berStatistics rows of the reference's evaluateCode (ldpc.py:350-388) on a freshly loaded module: 3 frames at
    each of three SNR points (the first low enough that frames fail and leave their messages behind for the next
    frame), noise from the module's own generator (ldpc.py:24, 372), ONE decoder object for all nine frames
    (ldpc.py:356, 377)."""
    ref = load_reference()
    import fileHandler
    H = np.array(fileHandler.readMatrixFromFile(os.path.join(REF, 'codeMatrices', 'nearEarthParity.txt'),
                                                1022, 8176, 511, True, False, False), dtype=np.int32)
    bs = ref.evaluateCode(3, 17, [2.6, 3.4, 4.0], 7154, 8176, 50, H)
    rows = bs.getRawStats()
    num = np.array([[float(x) for x in row[:8]] for row in rows], dtype=np.float64)
    status = np.array([row[8] for row in rows])
    np.savez_compressed(os.path.join(OUT, 'berstats_evaluate_code.npz'), rows=num, status=status)
    print('evaluateCode rows:', num.tolist(), status.tolist())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true', help='2 frames per point (plumbing check)')
    ap.add_argument('--jobs', type=int, default=8)
    ap.add_argument('--only-chained', action='store_true', help='only (re)generate chained_decoder.npz')
    ap.add_argument('--only', default=None, help='only (re)generate the file of this code, e.g. nearEarth_mut_many')
    ap.add_argument('--only-evaluate-code-rows', action='store_true', help='only (re)generate berstats_evaluate_code.npz')
    ap.add_argument('--only-wifi-rows', action='store_true', help='only (re)generate berstats_single_transmission_wifi.npz')
    args = ap.parse_args()
    nf = 2 if args.quick else None
    os.makedirs(OUT, exist_ok=True)
    if args.only_wifi_rows:
        gen_wifi_rows()
        return
    if args.only_evaluate_code_rows:
        gen_evaluate_code_rows()
        return
    if not args.only:
        gen_chained()
    if args.only_chained:
        return

    plan = []   # (code, seed, snr, frames, special)
    for i, snr in enumerate([3.0, 3.2, 3.4, 3.5, 3.6, 4.5]):
        # seeds: 460101 is the one ldpc.testNearEarth uses (ldpc.py:495)
        plan.append(('nearEarth', 460101 + i, snr, nf or 32, None))
    plan.append(('nearEarth', 7134066, 3.5, nf or 8, None))          # ldpc.py:24
    plan.append(('nearEarth', 61017449, 3.3, nf or 8, None))         # postProcessing.py:21 (42+61017406+1)
    for special in ('clean', 'flip0', 'zeros'):
        plan.append(('nearEarth', 1, 0.0, 1, special))
    for (i, j) in [(0, 0), (0, 7), (1, 3), (1, 15)]:
        plan.append((f'nearEarth_zero_{i}_{j}', 5000 + 16 * i + j, 3.4, nf or 8, None))
    # odd check degrees (33; 31 and 37): pins the reference's sign rule for odd degree
    plan.append(('nearEarth_mut_w3', 6001, 3.4, nf or 8, None))
    plan.append(('nearEarth_mut_w3', 6003, 6.0, nf or 8, None))
    plan.append(('nearEarth_mut_w7', 6002, 3.4, nf or 8, None))
    plan.append(('nearEarth_mut_w7', 6004, 6.0, nf or 8, None))
    # check degrees above 64 (the state of a long gym episode)
    plan.append(('nearEarth_mut_many', 6100, 3.0, nf or 6, None))
    plan.append(('nearEarth_mut_many', 6103, 4.0, nf or 6, None))
    plan.append(('nearEarth_mut_many', 6101, 5.0, nf or 4, None))
    plan.append(('nearEarth_mut_many', 6102, 8.0, nf or 6, None))
    plan.append(('nearEarth_mut_many', 1, 0.0, 1, 'flip0'))
    plan.append(('wifi_5_6', 7001, 4.0, nf or 32, None))
    plan.append(('wifi_5_6', 7003, 6.0, nf or 32, None))
    plan.append(('wifi_5_6', 7002, 8.0, nf or 32, None))
    plan.append(('wifi_5_6', 1, 0.0, 1, 'flip0'))
    plan.append(('rand_24_48', 8001, 2.0, nf or 32, None))
    plan.append(('rand_24_48', 8003, 4.0, nf or 64, None))
    plan.append(('rand_24_48', 8002, 6.0, nf or 32, None))

    if args.only:
        plan = [j for j in plan if j[0] == args.only]
    # split long jobs so the pool stays busy: one job per (point, chunk of frames)
    # is not possible (one RandomState stream per point), so just sort by cost.
    plan.sort(key=lambda j: -(j[3] * (60 - 10 * min(j[2], 5.0))))
    t0 = time.time()
    with mp.Pool(args.jobs) as pool:
        results = pool.map(run_job, plan, chunksize=1)
    print('reference decode wall %.1f s' % (time.time() - t0))

    os.makedirs(OUT, exist_ok=True)
    by_code = {}
    for r in results:
        by_code.setdefault(r['name'], []).append(r)
    ref = load_reference()
    for name, rs in by_code.items():
        H = build_code(name)
        rs.sort(key=lambda r: (r['special'], r['seed'], r['snr']))
        arrays = dict(
            shape=np.array(H.shape, np.int64),
            H_sha256=np.frombuffer(hashlib.sha256(np.packbits(H.astype(np.uint8)).tobytes()).digest(),
                                   dtype=np.uint8),
            row_degrees=np.unique(H.sum(1)), col_degrees=np.unique(H.sum(0)),
            nnz=np.array(int(H.sum()), np.int64),
            n_points=np.array(len(rs), np.int64),
            max_iters=np.array(MAX_ITERS, np.int64),
        )
        if H.size <= 4096:
            arrays['H'] = H.astype(np.uint8)
        for p, r in enumerate(rs):
            arrays[f'p{p}_seed'] = np.array(r['seed'], np.int64)
            arrays[f'p{p}_snr'] = np.array(r['snr'], np.float64)
            arrays[f'p{p}_special'] = np.array(r['special'])
            arrays[f'p{p}_iters'] = r['iters']
            arrays[f'p{p}_conv'] = r['conv']
            arrays[f'p{p}_hard'] = r['hard']
            arrays[f'p{p}_soft_sha256'] = r['sha']
            arrays[f'p{p}_soft_head'] = r['head']
            arrays[f'p{p}_llr0'] = r['llr0']       # raw fp32 LLRs of frame 0: guards RNG-path drift
            print(f"{name:22s} seed {r['seed']:9d} snr {r['snr']:4.1f} {r['special']:6s} "
                  f"frames {len(r['iters']):3d} mean it {r['iters'].mean():5.1f} conv {int(r['conv'].sum()):3d} "
                  f"({r['seconds']:.0f} s)")
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)

    if args.only:
        return
    # berStatistics rows of the reference's evaluateCodeAtSingleTransmission (ldpc.py:391-430)
    # on a freshly loaded module (module-global PRNG at its initial state, ldpc.py:24).
    import fileHandler
    H = np.array(fileHandler.readMatrixFromFile(os.path.join(REF, 'codeMatrices', 'nearEarthParity.txt'),
                                                1022, 8176, 511, True, False, False), dtype=np.int32)
    ref = load_reference()
    bs = ref.evaluateCodeAtSingleTransmission(460101, [3.0, 3.2], 7154, 8176, 50, H)
    rows = bs.getRawStats()
    num = np.array([[float(x) for x in row[:8]] for row in rows], dtype=np.float64)
    status = np.array([row[8] for row in rows])
    v2 = bs.getStatsV2()
    np.savez_compressed(os.path.join(OUT, 'berstats_single_transmission.npz'),
                        rows=num, status=status,
                        **{f'v2_{i}': np.asarray(a) for i, a in enumerate(v2)})
    print('berStatistics rows:', num.tolist(), status.tolist())
    gen_wifi_rows()
    gen_evaluate_code_rows()


if __name__ == '__main__':
    main()
