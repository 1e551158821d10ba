"""GPU suite (-m gpu): the HIP path, called through the C ABI, against
  * the golden vectors of the reference (float64 arithmetic: bit-exact hard decisions,
    iteration counts, status AND soft vectors),
  * the CPU oracle in float32 (bit-exact as well),
  * size-independent properties at bench scale.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import ALL_CODES, build_code, golden_points, make_llrs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engines():
    import ldpc_amd
    cache = {}

    def get(name, generic=False):
        """generic=True: force the table-driven QC kernel even where the unrolled
        kernel for regular codes applies (the library reads the variable at creation)."""
        key = (name, generic)
        if key not in cache:
            qc, H = build_code(name)
            os.environ['LDPC_HIP_FORCE_GENERIC'] = '1' if generic else '0'
            try:
                cache[key] = ldpc_amd.Decoder(qc if qc is not None else H)
            finally:
                os.environ.pop('LDPC_HIP_FORCE_GENERIC', None)
        return cache[key]
    yield get
    for e in cache.values():
        e.close()


@pytest.mark.parametrize('name', ALL_CODES)
def test_f64_bit_exact_against_reference_golden(engines, name):
    eng = engines(name)
    for pt in golden_points(name):
        r = eng.decode(pt['llr'], pt['max_iters'], prec=64, want_soft=True)
        assert np.array_equal(r['iters'], pt['iters']), (name, pt['seed'], r['iters'], pt['iters'])
        assert np.array_equal(r['converged'], pt['conv'])
        assert np.array_equal(r['hard_packed'], pt['hard'])
        assert np.array_equal(r['ones'], r['hard'].sum(1))
        for k in range(len(pt['iters'])):
            if pt['iters'][k] == 0:
                continue    # the host mirror returns the input itself for 0 iterations
            assert hashlib.sha256(r['soft'][k].tobytes()).digest() == bytes(pt['sha'][k]), (name, pt['seed'], k)


@pytest.mark.parametrize('name', ['nearEarth', 'nearEarth_zero_1_3', 'wifi_5_6'])
def test_table_driven_kernel_on_regular_codes(engines, oracle_mod, name):
    """Codes the unrolled kernel takes by default, pushed through the table-driven one."""
    eng = engines(name, generic=True)
    _, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    for pt in golden_points(name):
        Y = pt['llr'][:8]
        r = eng.decode(Y, pt['max_iters'], prec=64, want_soft=True)
        assert np.array_equal(r['iters'], pt['iters'][:8]) and np.array_equal(r['hard_packed'], pt['hard'][:8])
        hard, iters, conv, soft = orc.minsum(Y, 50, prec=32, threads=8)
        r = eng.decode(Y, 50, prec=32, want_soft=True)
        assert np.array_equal(r['iters'], iters) and np.array_equal(r['soft'], soft)


@pytest.mark.parametrize('name', ALL_CODES)
def test_f32_bit_exact_against_oracle(engines, oracle_mod, name):
    eng = engines(name)
    _, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    for pt in golden_points(name):
        Y = pt['llr'][:16]
        hard, iters, conv, soft = orc.minsum(Y, 50, prec=32, threads=8)
        r = eng.decode(Y, 50, prec=32, want_soft=True)
        assert np.array_equal(r['iters'], iters), (name, pt['seed'])
        assert np.array_equal(r['converged'], conv)
        assert np.array_equal(r['hard'], hard)
        assert np.array_equal(r['soft'], soft)


@pytest.mark.parametrize('name', ['nearEarth_mut_w3', 'nearEarth_mut_w7', 'nearEarth_mut_many', 'wifi_5_6', 'rand_24_48'])
def test_corrected_sign_mode_against_oracle(engines, oracle_mod, name):
    """Odd check degrees: opt-in sign rule (-1)^degree, both arithmetic widths."""
    eng = engines(name)
    _, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    pt = [p for p in golden_points(name) if not p['special']][0]
    Y = pt['llr'][:8]
    for prec, dt in ((64, np.float64), (32, np.float32)):
        hard, iters, conv, soft = orc.minsum(Y.astype(dt), 50, prec=prec, sign_mode=1, threads=8)
        r = eng.decode(Y, 50, prec=prec, sign_mode=1, want_soft=True)
        assert np.array_equal(r['iters'], iters) and np.array_equal(r['converged'], conv)
        assert np.array_equal(r['hard'], hard) and np.array_equal(r['soft'], soft)
    if name == 'wifi_5_6':
        # with the textbook sign rule the WiFi code does decode at 6 dB
        pt6 = [p for p in golden_points(name) if p['snr'] == 6.0][0]
        r = eng.decode(pt6['llr'], 50, prec=64, sign_mode=1)
        assert r['converged'].all() and r['ones'].sum() == 0


@pytest.mark.parametrize('prec', [64, 32])
def test_chained_decoder_against_oracle(engines, oracle_mod, prec):
    """ldpc_decode_chained: state carried across frames AND across calls; reset works."""
    import ldpc_amd
    dt = np.float64 if prec == 64 else np.float32
    for name in ('nearEarth', 'nearEarth_mut_w7', 'nearEarth_mut_many', 'wifi_5_6', 'rand_24_48'):
        qc, H = build_code(name)
        orc = oracle_mod.Oracle(H)
        pts = [p for p in golden_points(name) if not p['special']]
        Y = np.concatenate([pts[-1]['llr'][:5], pts[0]['llr'][:3]])
        state = orc.new_state(prec)
        hard, iters, conv, soft = orc.minsum(Y.astype(dt), 50, prec=prec, state=state)
        with ldpc_amd.Decoder(qc if qc is not None else H) as eng:
            r1 = eng.decode(Y[:3], 50, prec=prec, want_soft=True, chained=True)
            r2 = eng.decode(Y[3:], 50, prec=prec, want_soft=True, chained=True)
            got_it = np.concatenate([r1['iters'], r2['iters']])
            got_soft = np.concatenate([r1['soft'], r2['soft']])
            assert np.array_equal(got_it, iters), (name, got_it, iters)
            assert np.array_equal(got_soft, soft)
            assert np.array_equal(np.concatenate([r1['hard'], r2['hard']]), hard)
            eng.reset_state()
            r3 = eng.decode(Y[:3], 50, prec=prec, want_soft=True, chained=True)
            assert np.array_equal(r3['iters'], iters[:3]) and np.array_equal(r3['soft'], soft[:3])
            # the unchained entry point is unaffected by the state
            f = orc.minsum(Y.astype(dt), 50, prec=prec)
            r4 = eng.decode(Y, 50, prec=prec, want_soft=True)
            assert np.array_equal(r4['iters'], f[1]) and np.array_equal(r4['soft'], f[3])


def test_float64_channel_values(engines, oracle_mod):
    """float64 inputs that are not float32-representable (what ldpc.addAWGN produces)."""
    eng = engines('nearEarth')
    _, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    prng = np.random.RandomState(7134066)
    sigma = np.sqrt(0.5 / 10 ** 0.34)
    Y = -1.0 + prng.normal(0, sigma, (6, 8176))
    hard, iters, conv, soft = orc.minsum(Y, 50, prec=64, threads=6)
    r = eng.decode(Y, 50, prec=64, want_soft=True)
    assert np.array_equal(r['iters'], iters) and np.array_equal(r['hard'], hard)
    assert np.array_equal(r['soft'], soft)


def test_edge_cases(engines, oracle_mod):
    eng = engines('nearEarth')
    n = eng.n
    # empty batch
    r = eng.decode(np.zeros((0, n), np.float32), 50, prec=32)
    assert r['iters'].shape == (0,) and r['hard'].shape == (0, n)
    # iteration budget 0 and 1
    pt = [p for p in golden_points('nearEarth') if p['snr'] == 3.4 and not p['special']][0]
    Y = pt['llr'][:5]
    _, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    for budget in (0, 1, 3):
        hard, iters, conv, soft = orc.minsum(Y.astype(np.float64), budget, prec=64)
        r = eng.decode(Y, budget, prec=64, want_soft=True)
        assert np.array_equal(r['iters'], iters) and np.array_equal(r['converged'], conv)
        assert np.array_equal(r['hard'], hard) and np.array_equal(r['soft'], soft)
    # negative zero and exact zero inputs decide 0
    y = np.full((1, n), -1.0, np.float32)
    y[0, :4] = [-0.0, 0.0, -0.0, 0.0]
    hard, iters, conv, _ = orc.minsum(y.astype(np.float64), 50, prec=64)
    r = eng.decode(y, 50, prec=64)
    assert np.array_equal(r['hard'], hard) and r['iters'][0] == iters[0] == 0
    # erasure burst (zeros) in a noisy frame: the decoder has to fill it in
    y = pt['llr'][:2].copy()
    y[:, 100:140] = 0.0
    hard, iters, conv, soft = orc.minsum(y.astype(np.float64), 50, prec=64)
    r = eng.decode(y, 50, prec=64, want_soft=True)
    assert np.array_equal(r['iters'], iters) and np.array_equal(r['soft'], soft)


@pytest.mark.parametrize('name', ['nearEarth', 'nearEarth_zero_1_3', 'nearEarth_mut_w3', 'wifi_5_6'])
def test_small_iteration_budgets_float32(engines, oracle_mod, name):
    """The float32 kernels (message planes for the NearEarth shapes, the short-code kernel for 802.11n)
    with budgets that end a frame before it converges: iterations, status, bits and soft values
    against the oracle, at an SNR where frames need 4 to 50 iterations."""
    eng = engines(name)
    _, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    sm = 0 if name in ('nearEarth', 'nearEarth_zero_1_3') else 1
    Y = make_llrs(5150, 3.4 if name.startswith('nearEarth') else 3.5, H.shape[1], 40)
    seen = set()
    for budget in (0, 1, 2, 3, 7):
        hard, iters, conv, soft = orc.minsum(Y, budget, prec=32, sign_mode=sm, threads=8)
        r = eng.decode(Y, budget, prec=32, sign_mode=sm, want_soft=True)
        assert np.array_equal(r['iters'], iters), (name, budget, r['iters'], iters)
        assert np.array_equal(r['converged'], conv) and np.array_equal(r['hard'], hard)
        nz = iters > 0          # 0 iterations: the posterior is the input
        assert np.array_equal(r['soft'][nz], soft[nz])
        assert np.array_equal(r['soft'][~nz], Y[~nz])
        seen.add((int(iters.max()), bool(conv.all())))
    assert (0, False) in seen and (3, False) in seen     # budgets really cut frames short


def test_ragged_batches_on_packed_frames(engines, oracle_mod):
    """WiFi frames are packed several to a workgroup; batch sizes that do not fill the last one."""
    eng = engines('wifi_5_6')
    _, H = build_code('wifi_5_6')
    orc = oracle_mod.Oracle(H)
    info = eng.launch_info(prec=32)
    assert info['frames_per_wg'] >= 2
    Y = np.concatenate([make_llrs(11, 8.0, 1944, 9), make_llrs(12, 5.0, 1944, 4)])
    hard, iters, conv, soft = orc.minsum(Y, 50, prec=32, sign_mode=1, threads=8)
    for F in (1, 2, 3, 5, 13):
        r = eng.decode(Y[:F], 50, prec=32, sign_mode=1, want_soft=True)
        assert np.array_equal(r['iters'], iters[:F]) and np.array_equal(r['hard'], hard[:F])
        assert np.array_equal(r['soft'], soft[:F])
    assert len(set(iters.tolist())) > 2      # frames of one workgroup stop at different iterations


def test_device_entry_points_and_channel(engines):
    import torch
    eng = engines('nearEarth')
    n, F = eng.n, 512
    dev = torch.device('cuda:0')
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    sig = torch.empty(F, dtype=torch.float32, device=dev)
    unc = torch.empty(F, dtype=torch.int32, device=dev)
    eng.channel_awgn_device(llr, F, 3.5, seed=1234, first_frame=0, d_sigma_hat=sig, d_uncoded=unc)
    # same frames again, shifted window: counter-based noise does not depend on the batch
    llr2 = torch.empty((F // 2, n), dtype=torch.float32, device=dev)
    eng.channel_awgn_device(llr2, F // 2, 3.5, seed=1234, first_frame=F // 2)
    torch.cuda.synchronize()
    assert torch.equal(llr[F // 2:], llr2)
    y = llr.cpu().numpy().astype(np.float64)
    sigma = np.sqrt(0.5 / 10 ** 0.35)
    noise = y + 1.0
    assert abs(noise.mean()) < 4 * sigma / np.sqrt(noise.size)
    assert abs(noise.std() / sigma - 1) < 5e-3
    z = noise / sigma
    assert abs((z ** 4).mean() - 3.0) < 0.02 and abs((z ** 3).mean()) < 0.01
    assert np.allclose(sig.cpu().numpy(), np.sqrt((noise ** 2).mean(1)), rtol=1e-5)
    assert np.array_equal(unc.cpu().numpy(), (y > 0).sum(1))
    # decode on the device, compare with the host entry point on the same values
    hard = torch.zeros((F, eng.nbytes), dtype=torch.uint8, device=dev)
    iters = torch.zeros(F, dtype=torch.int32, device=dev)
    conv = torch.zeros(F, dtype=torch.uint8, device=dev)
    ones = torch.zeros(F, dtype=torch.int32, device=dev)
    eng.decode_device(llr, F, 50, prec=32, d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones)
    torch.cuda.synchronize()
    eng.set_stream(None)
    r = eng.decode(llr.cpu().numpy(), 50, prec=32)
    assert np.array_equal(r['iters'], iters.cpu().numpy())
    assert np.array_equal(r['hard_packed'], hard.cpu().numpy())
    assert np.array_equal(r['ones'], ones.cpu().numpy())
    assert conv.cpu().numpy().mean() > 0.95          # 3.5 dB: essentially every frame decodes


def test_sweep_driver_counts(engines):
    eng = engines('nearEarth')
    snr = [3.0, 3.5, 4.5]
    F = 256
    cnt, pf = eng.eval_awgn(snr, F, seed=99, max_iters=50, prec=32, per_frame=True)
    assert cnt.shape == (3, 6) and pf.shape == (3, F, 4)
    assert np.all(cnt[:, 0] == F)
    assert np.array_equal(cnt[:, 2], pf[:, :, 2].sum(1).astype(np.uint64))
    assert np.array_equal(cnt[:, 3], pf[:, :, 3].sum(1).astype(np.uint64))
    assert np.array_equal(cnt[:, 4], pf[:, :, 1].sum(1).astype(np.uint64))
    assert np.array_equal(cnt[:, 1], (pf[:, :, 2] > 0).sum(1).astype(np.uint64))
    mean_it = cnt[:, 3] / F
    assert mean_it[0] > 30 and 4 < mean_it[1] < 12 and mean_it[2] < 4      # cf. reference runs: 47.8 / 7.6 / 2.6
    assert cnt[1, 1] <= 2 and cnt[2, 1] == 0
    # uncoded BER ~ Q(1/sigma)
    from math import erfc, sqrt
    for k, s in enumerate(snr):
        sigma = sqrt(0.5 / 10 ** (s / 10))
        q = 0.5 * erfc(1 / sigma / sqrt(2))
        assert abs(cnt[k, 4] / (F * 8176) / q - 1) < 0.03
    # splitting the frame range reproduces the same totals (counter-based noise)
    c1, _ = eng.eval_awgn(snr, F // 2, seed=99, first_frame=0, prec=32)
    c2, _ = eng.eval_awgn(snr, F // 2, seed=99, first_frame=F // 2, prec=32)
    assert np.array_equal(c1 + c2, cnt)


def test_bench_scale_properties(engines):
    """At bench sizes there is no oracle run; check invariants of the outputs instead."""
    import torch
    eng = engines('nearEarth')
    n, F = eng.n, 20000
    dev = torch.device('cuda:0')
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.channel_awgn_device(llr, F, 3.5, seed=7, first_frame=0)
    hard = torch.zeros((F, eng.nbytes), dtype=torch.uint8, device=dev)
    iters = torch.zeros(F, dtype=torch.int32, device=dev)
    conv = torch.zeros(F, dtype=torch.uint8, device=dev)
    ones = torch.zeros(F, dtype=torch.int32, device=dev)
    eng.decode_device(llr, F, 50, prec=32, d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones)
    torch.cuda.synchronize()
    it, cv, on = iters.cpu().numpy(), conv.cpu().numpy(), ones.cpu().numpy()
    hb = np.unpackbits(hard.cpu().numpy(), axis=1, bitorder='little')[:, :n]
    assert np.array_equal(on, hb.sum(1))
    H = eng.qc.to_dense().astype(np.uint8)
    # status is exactly "syndrome of the returned hard decision is zero"
    sub = np.concatenate([np.flatnonzero(on)[:50], np.arange(200)])
    synd = (hb[sub].astype(np.int32) @ H.T.astype(np.int32)) % 2
    assert np.array_equal(synd.any(1) == 0, cv[sub] == 1)
    assert np.all(it[cv == 0] == 50) and np.all(it[cv == 1] <= 50)
    assert cv.mean() > 0.995 and 5 < it.mean() < 10
    # idempotence: decoding again gives the same answer; order of frames does not matter
    iters2 = torch.zeros(F, dtype=torch.int32, device=dev)
    perm = torch.randperm(F, device=dev)
    eng.decode_device(llr[perm].contiguous(), F, 50, prec=32, d_iters=iters2)
    torch.cuda.synchronize()
    eng.set_stream(None)
    assert torch.equal(iters2, iters[perm])


def test_short_code_kernel_against_the_other_paths(engines, oracle_mod, monkeypatch):
    """float32 decoding of short codes runs several frames per workgroup on the edge list
    (802.11n: 3 frames); the older per-block kernels must give the same bits."""
    for name in ('wifi_5_6', 'rand_24_48'):
        eng = engines(name)
        _, H = build_code(name)
        orc = oracle_mod.Oracle(H)
        Y = make_llrs(77, 5.0 if name == 'wifi_5_6' else 2.0, H.shape[1], 67)      # ragged: 67 = 22*3 + 1
        hard, iters, conv, soft = orc.minsum(Y, 30, prec=32, threads=8)
        info = eng.launch_info(prec=32)
        assert info['frames_per_wg'] >= 3
        fast = eng.decode(Y, 30, prec=32, want_soft=True)
        monkeypatch.setenv('LDPC_HIP_NO_SLOTS', '1')
        assert eng.launch_info(prec=32) != info
        slow = eng.decode(Y, 30, prec=32, want_soft=True)
        monkeypatch.delenv('LDPC_HIP_NO_SLOTS')
        for r in (fast, slow):
            assert np.array_equal(r['iters'], iters) and np.array_equal(r['converged'], conv)
            assert np.array_equal(r['hard'], hard) and np.array_equal(r['soft'], soft)
            assert np.array_equal(r['ones'], hard.sum(1))
        # corrected sign rule (odd check degrees decode) through the same kernel
        h2, i2, c2, _ = orc.minsum(Y, 30, prec=32, sign_mode=1, want_soft=False, threads=8)
        r2 = eng.decode(Y, 30, prec=32, sign_mode=1)
        assert np.array_equal(r2['iters'], i2) and np.array_equal(r2['hard'], h2)


def test_full_size_batch_properties(engines):
    """The benchmark's own batch (10^6 NearEarth frames at 3.5 dB, float32): no oracle run at
    this size, so size-independent properties -- a checksum of checksums (per-frame ones
    counts against the packed bits), status vs iteration budget, run-to-run determinism,
    independence of how the batch is split, and agreement with a float64 decode of a slice
    through the golden-pinned kernel wherever both converge."""
    import torch
    eng = engines('nearEarth')
    n, F = eng.n, 1_000_000
    dev = torch.device('cuda:0')
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    eng.channel_awgn_device(llr, F, 3.5, seed=460101, first_frame=0)
    out = []
    for rep in range(2):
        hard = torch.zeros((F, eng.nbytes), dtype=torch.uint8, device=dev)
        iters = torch.full((F,), -1, dtype=torch.int32, device=dev)
        conv = torch.full((F,), 7, dtype=torch.uint8, device=dev)
        ones = torch.full((F,), -1, dtype=torch.int32, device=dev)
        eng.decode_device(llr, F, 50, prec=32, d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones)
        torch.cuda.synchronize()
        out.append((hard, iters, conv, ones))
    (hard, iters, conv, ones), (hard2, iters2, conv2, ones2) = out
    assert torch.equal(hard, hard2) and torch.equal(iters, iters2) and torch.equal(conv, conv2) and torch.equal(ones, ones2)
    del hard2, iters2, conv2, ones2, out
    lut = torch.tensor([bin(v).count('1') for v in range(256)], dtype=torch.int32, device=dev)
    pop = torch.zeros(F, dtype=torch.int32, device=dev)
    for f0 in range(0, F, 100000):
        pop[f0:f0 + 100000] = lut[hard[f0:f0 + 100000].long()].sum(1, dtype=torch.int32)
    assert torch.equal(pop, ones)                                     # checksum of checksums
    assert int(((conv != 0) & (conv != 1)).sum()) == 0
    assert int(((iters < 0) | (iters > 50)).sum()) == 0
    assert int(((conv == 0) & (iters != 50)).sum()) == 0             # gave up <=> budget spent
    assert int(((conv == 1) & (ones != 0)).sum()) <= 2                # a converged frame is the sent word (or, rarely, another codeword)
    fer = float((ones != 0).float().mean())
    assert 1e-3 < fer < 5e-3 and 6.5 < float(iters.float().mean()) < 7.5
    # splitting the batch differently changes nothing
    a, b = 333_333, 777_001
    it_part = torch.empty(b - a, dtype=torch.int32, device=dev)
    hd_part = torch.empty((b - a, eng.nbytes), dtype=torch.uint8, device=dev)
    eng.decode_device(llr[a:b], b - a, 50, prec=32, d_hard=hd_part, d_iters=it_part)
    torch.cuda.synchronize()
    assert torch.equal(it_part, iters[a:b]) and torch.equal(hd_part, hard[a:b])
    # float64 decode (the arithmetic pinned by the reference's golden vectors) of a slice
    S = 50_000
    it64 = torch.empty(S, dtype=torch.int32, device=dev)
    cv64 = torch.empty(S, dtype=torch.uint8, device=dev)
    hd64 = torch.empty((S, eng.nbytes), dtype=torch.uint8, device=dev)
    eng.decode_device(llr[:S], S, 50, prec=64, d_hard=hd64, d_iters=it64, d_conv=cv64)
    torch.cuda.synchronize()
    eng.set_stream(None)
    both = (cv64 == 1) & (conv[:S] == 1)
    assert float(both.float().mean()) > 0.99
    assert torch.equal(hd64[both], hard[:S][both])
    assert float((it64[both] == iters[:S][both]).float().mean()) > 0.999   # rounding may move a frame by one iteration


def test_light_circulants_on_the_message_plane_kernel(oracle_mod):
    """Circulants of weight 0, 1 and 2 mixed (what a gym action with at most two hot bits, or a
    swap of two circulants, produces): float32 runs on the message-plane kernel with the
    missing edges parked on a dummy record -- odd and even numbers of them per row, in both
    block-rows -- and must stay bit-identical to the oracle, for both sign rules."""
    import ldpc_amd
    from ldpc_amd import codes
    base = codes.near_earth()
    variants = [
        base.with_block(1, 9, [42]),                                            # one edge missing in block-row 1
        base.with_block(0, 0, [7]).with_block(0, 15, []).with_block(1, 3, [500]).with_block(1, 4, [0, 510]),
        base.with_block(0, 5, base.blocks[1][6]).with_block(1, 6, base.blocks[0][5]),   # a swap: still all weight 2
    ]
    for qc in variants:
        H = qc.to_dense()
        orc = oracle_mod.Oracle(H)
        Y = make_llrs(1234, 3.3, qc.n, 20)
        Y[3, :40] = 0.0                                                          # erased symbols: exact zeros
        with ldpc_amd.Decoder(qc) as eng:
            assert eng.launch_info(prec=32)['lds_bytes'] == 163824              # the message-plane kernel
            for sm in (0, 1):
                hard, iters, conv, soft = orc.minsum(Y, 50, prec=32, sign_mode=sm, threads=8)
                r = eng.decode(Y, 50, prec=32, sign_mode=sm, want_soft=True)
                assert np.array_equal(r['iters'], iters), (qc.row_degrees(), sm, r['iters'], iters)
                assert np.array_equal(r['converged'], conv) and np.array_equal(r['hard'], hard)
                assert np.array_equal(r['soft'], soft)
            # float64 of the same code goes through the table-driven kernel
            h64, i64, c64, s64 = orc.minsum(Y, 50, prec=64, threads=8)
            r = eng.decode(Y, 50, prec=64, want_soft=True)
            assert np.array_equal(r['iters'], i64) and np.array_equal(r['soft'], s64)
            # sum-product on the same planes (corrected sign rule: some rows have odd degree)
            scale = np.float32(4 * 10 ** 0.33)
            hs, is_, cs, _ = orc.spa(Y, float(scale), 50, prec=32, sign_mode=1, want_soft=False, threads=8)
            r = eng.decode(Y * scale, 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=1)
            cv = cs == 1
            assert cv.sum() >= 15
            assert np.array_equal(r['iters'][cv], is_[cv]) and np.array_equal(r['hard'][cv], hs[cv])


@pytest.mark.parametrize('name', ['nearEarth', 'wifi_5_6', 'nearEarth_mut_w3', 'rand_24_48'])
def test_normalised_and_offset_minsum(engines, oracle_mod, name):
    """Optional decoder variant (no counterpart in the reference, so parity is against this
    build's oracle only): |message| = max(alpha*min - beta, 0).  Bit-exact in both precisions on
    every kernel family, and (1, 0) restores the plain rule."""
    eng = engines(name)
    _, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    pt = next(iter(golden_points(name)))
    Y = pt['llr'][:12]
    sm = 1 if name in ('wifi_5_6', 'nearEarth_mut_w3') else 0
    try:
        for alpha, beta in ((0.8125, 0.0), (1.0, 0.15), (0.9, 0.05)):
            eng.set_minsum_scaling(alpha, beta)
            for prec in (32, 64):
                hard, iters, conv, soft = orc.minsum(Y, 50, prec=prec, sign_mode=sm, threads=8, alpha=alpha, beta=beta)
                r = eng.decode(Y, 50, prec=prec, sign_mode=sm, want_soft=True)
                assert np.array_equal(r['iters'], iters), (name, alpha, beta, prec, r['iters'], iters)
                assert np.array_equal(r['hard'], hard) and np.array_equal(r['soft'], soft)
    finally:
        eng.set_minsum_scaling(1.0, 0.0)
    hard, iters, conv, soft = orc.minsum(Y, 50, prec=32, sign_mode=sm, threads=8)
    r = eng.decode(Y, 50, prec=32, sign_mode=sm, want_soft=True)
    assert np.array_equal(r['iters'], iters) and np.array_equal(r['soft'], soft)
    with pytest.raises(Exception):
        eng.set_minsum_scaling(0.0, 0.0)


@pytest.mark.parametrize('name,prec', [('nearEarth', 32), ('nearEarth', 64), ('wifi_5_6', 32), ('wifi_5_6', 64),
                                      ('nearEarth_mut_w3', 32), ('nearEarth_mut_w7', 32), ('nearEarth_zero_1_3', 32),
                                      ('rand_24_48', 32)])
def test_erasures_and_exact_zeros(engines, oracle_mod, name, prec):
    """Channel values that are exactly +0.0 / -0.0 (erased or punctured positions) scattered over
    noisy frames: the slicer decides 0 for them (bit = x > 0, ldpc.py:47), their posteriors can stay
    exactly zero for several iterations, and every kernel has to agree with the oracle bit for bit
    (hard decisions, iteration counts, status, soft values)."""
    qc, H = build_code(name)
    n = H.shape[1]
    prng = np.random.RandomState(97)
    Y = make_llrs(4242, 3.6 if name.startswith('nearEarth') else 5.0, n, 24)
    for f in range(Y.shape[0]):
        k = min([0, 1, 8, 40, 200, n // 8][f % 6], n // 2)
        pos = prng.choice(n, k, replace=False)
        Y[f, pos] = np.where(prng.rand(k) < 0.5, 0.0, -0.0).astype(np.float32)
    Y[3, :] = 0.0                                     # nothing received at all: syndrome 0, stops at once
    dt = np.float64 if prec == 64 else np.float32
    orc = oracle_mod.Oracle(H)
    sign_mode = 0 if name.startswith('nearEarth') and 'mut' not in name else 1
    hard, iters, conv, soft = orc.minsum(Y.astype(dt), 50, prec=prec, sign_mode=sign_mode, threads=8)
    r = engines(name).decode(Y, 50, prec=prec, sign_mode=sign_mode, want_soft=True)
    assert np.array_equal(r['iters'], iters), (r['iters'], iters)
    assert np.array_equal(r['converged'], conv)
    assert np.array_equal(r['hard'], hard)
    assert np.array_equal(r['soft'], soft)
    assert iters[3] == 0 and conv[3] == 1 and iters.max() >= 1


@pytest.mark.parametrize('name,prec,algo', [('nearEarth', 32, 0), ('nearEarth', 64, 0), ('nearEarth', 32, 1),
                                           ('wifi_5_6', 32, 1), ('nearEarth_zero_1_3', 64, 0)])
def test_syndrome_only_pass_changes_nothing(engines, monkeypatch, name, prec, algo):
    """Kernels start an iteration with a syndrome-only pass when the pass before left few rows
    unsatisfied (a frame's last check pass is otherwise thrown away).  Never, by default and before
    EVERY iteration (LDPC_HIP_PRECHECK_T = 0 / unset / huge): the same bits, iteration counts,
    status and soft values, frame for frame, at an SNR where frames stop early, late and not at all."""
    qc, H = build_code(name)
    n = H.shape[1]
    snr = 3.3 if name.startswith('nearEarth') else 3.0
    Y = make_llrs(777, snr, n, 600)
    if algo == 1:
        Y = (Y * np.float32(4.0 * 10 ** (snr / 10))).astype(np.float32)      # sum-product takes 2y/sigma^2
    sign_mode = 0 if name.startswith('nearEarth') else 1
    eng = engines(name)
    out = {}
    for label, value in (('never', '0'), ('default', None), ('always', '500000')):
        if value is None:
            monkeypatch.delenv('LDPC_HIP_PRECHECK_T', raising=False)
        else:
            monkeypatch.setenv('LDPC_HIP_PRECHECK_T', value)
        out[label] = eng.decode(Y, 50, prec=prec, algo=algo, sign_mode=sign_mode, want_soft=True)
    ref = out['never']
    assert 0 < ref['iters'].min() < ref['iters'].max() and ref['converged'].mean() > 0.2
    if algo == 0:
        assert ref['iters'].max() == 50 and ref['converged'].mean() < 1.0
    for label in ('default', 'always'):
        for k in ('hard_packed', 'iters', 'converged', 'ones', 'soft'):
            assert np.array_equal(out[label][k], ref[k]), (label, k)


@pytest.mark.parametrize('name,prec', [('nearEarth', 32), ('nearEarth', 64), ('wifi_5_6', 32)])
def test_claimed_frames_equal_fixed_stride(engines, monkeypatch, name, prec):
    """Workgroups claim their next frames from a device counter (frames differ in iteration count);
    LDPC_HIP_STATIC_FRAMES=1 deals them out in a fixed stride instead.  Which workgroup decodes a
    frame must not show in any result: a batch several times the grid, both ways."""
    qc, H = build_code(name)
    n = H.shape[1]
    Y = make_llrs(31337, 3.4 if name.startswith('nearEarth') else 3.2, n, 1500 if n > 4000 else 4000)
    sign_mode = 0 if name.startswith('nearEarth') else 1
    eng = engines(name)
    monkeypatch.delenv('LDPC_HIP_STATIC_FRAMES', raising=False)
    claimed = eng.decode(Y, 50, prec=prec, sign_mode=sign_mode, want_soft=True)
    monkeypatch.setenv('LDPC_HIP_STATIC_FRAMES', '1')
    fixed = eng.decode(Y, 50, prec=prec, sign_mode=sign_mode, want_soft=True)
    for k in ('hard_packed', 'iters', 'converged', 'ones', 'soft'):
        assert np.array_equal(claimed[k], fixed[k]), k
    assert claimed['iters'].min() < claimed['iters'].max()


@pytest.mark.parametrize('name', ['nearEarth', 'nearEarth_zero_1_3'])
def test_fused_channel_sweep_equals_unfused(engines, monkeypatch, name):
    """The sweep driver's decode kernel makes its own channel values where it can (message-plane
    kernels); LDPC_HIP_NO_FUSED_CHANNEL=1 runs the channel kernel into a buffer first.  Same noise
    stream, same arithmetic: counters and every per-frame column (sigma_hat, uncoded, decoded,
    iterations) must be identical, for min-sum and sum-product, whatever the frame window."""
    eng = engines(name)
    for algo in (0, 1):
        for first, F in ((0, 1500), (123457, 700)):
            monkeypatch.delenv('LDPC_HIP_NO_FUSED_CHANNEL', raising=False)
            c1, p1 = eng.eval_awgn([3.0, 3.5], F, seed=5, first_frame=first, per_frame=True, algo=algo)
            monkeypatch.setenv('LDPC_HIP_NO_FUSED_CHANNEL', '1')
            c2, p2 = eng.eval_awgn([3.0, 3.5], F, seed=5, first_frame=first, per_frame=True, algo=algo)
            monkeypatch.delenv('LDPC_HIP_NO_FUSED_CHANNEL')
            assert np.array_equal(c1, c2), (algo, c1, c2)
            assert np.array_equal(p1, p2)
            assert c1[0, 3] > c1[1, 3] > 0 and (c1[:, 0] == F).all()
    # sigma_hat is the sample RMS of the noise (ldpc.py:58): close to sigma, frame by frame
    sigma = np.sqrt(0.5 / 10 ** 0.35)
    assert np.allclose(p1[1, :, 0], sigma, rtol=0.05) and p1[1, :, 0].std() > 0


def test_small_sweep_in_one_launch_equals_one_launch_per_point(monkeypatch):
    """A sweep of at most 8 points and 8192 frames in all goes out as one launch (the points' frames side by
    side); LDPC_HIP_NO_POINT_BATCH=1 launches point after point.  Same noise per (point, frame): counters
    and per-frame columns identical, for min-sum and sum-product, on the regular and on a heavy code."""
    import ldpc_amd
    from ldpc_amd import codes
    base = codes.near_earth()
    many, _ = build_code('nearEarth_mut_many')       # table-driven kernel: reads its channel values, batched all the same
    for code in (base, base.with_block(1, 4, [3, 77, 200, 401, 500]), many, codes.wifi_1944('1/2')):
        with ldpc_amd.Decoder(code) as eng:
            for algo in (0, 1):
                for snrs, F, first in (([3.0, 3.2, 3.4], 10, 0), ([2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5, 6.0], 1024, 999),
                                       ([3.1, 3.6], 1, 7)):
                    monkeypatch.delenv('LDPC_HIP_NO_POINT_BATCH', raising=False)
                    c1, p1 = eng.eval_awgn(snrs, F, seed=11, first_frame=first, per_frame=True, algo=algo, sign_mode=1)
                    monkeypatch.setenv('LDPC_HIP_NO_POINT_BATCH', '1')
                    c2, p2 = eng.eval_awgn(snrs, F, seed=11, first_frame=first, per_frame=True, algo=algo, sign_mode=1)
                    monkeypatch.delenv('LDPC_HIP_NO_POINT_BATCH')
                    assert np.array_equal(c1, c2), (algo, snrs, c1, c2)
                    assert np.array_equal(p1, p2)
                    assert (c1[:, 0] == F).all()
            # without per-frame output, and a sweep too large to batch
            c1, _ = eng.eval_awgn([3.0, 3.4], 300, seed=2)
            assert c1[1, 0] == 300 and c1[0, 4] > c1[1, 4]
            c4, _ = eng.eval_awgn([3.0, 3.4], 5000, seed=2)
            assert (c4[:, 0] == 5000).all()


def test_edge_tables_are_built_on_first_need(oracle_mod, monkeypatch):
    """A code the message-plane kernels take gets its edge list only when a kernel that walks it is chosen
    (float64 sum-product here): the results are those of a handle that built it at creation
    (LDPC_HIP_EAGER_EDGES=1), and the allocation-free entry point says what to do instead of allocating."""
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    code = codes.near_earth().with_block(0, 6, [5, 90, 300])
    prng = np.random.RandomState(8)
    Y = (-1 + 0.47 * prng.normal(size=(24, code.n))).astype(np.float32)
    outs = []
    for eager in ('1', None):
        if eager:
            monkeypatch.setenv('LDPC_HIP_EAGER_EDGES', eager)
        else:
            monkeypatch.delenv('LDPC_HIP_EAGER_EDGES', raising=False)
        with ldpc_amd.Decoder(code) as eng:
            a = eng.decode(Y, 20, prec=32, sign_mode=1)                       # message planes: no edge list needed
            if not eager:
                d = torch.from_numpy(Y).cuda()
                hard = torch.empty((24, eng.nbytes), dtype=torch.uint8, device='cuda')
                with pytest.raises(ldpc_amd.LdpcHipError, match='edge tables'):
                    eng.decode_device(d, 24, 20, prec=64, algo=ldpc_amd.ALGO_SUMPRODUCT, llr_scale=8.0, d_hard=hard, noalloc=True)
            b = eng.decode(Y * 8.0, 20, prec=64, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=1)
            c = eng.decode(Y, 20, prec=32, sign_mode=1)
            outs.append((a, b, c))
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x['hard'], y['hard']) and np.array_equal(x['iters'], y['iters']) and np.array_equal(x['converged'], y['converged'])
    assert np.array_equal(outs[1][0]['hard'], outs[1][2]['hard'])
    assert outs[1][1]['iters'].max() > 1


def test_check_degrees_above_64(oracle_mod):
    """The state of a long gym episode (every action leaves a circulant of up to 7 hot bits, ldpc_env.py:293-317)
    has checks of 84 and 55 edges.  Such a code decodes from the edge list with one explicit message per
    edge, whatever the degree: float64 against the reference's own results is in the golden tests; here both
    sign rules and precisions against the oracle, sum-product, the dense-matrix and CSR routes, the sweep
    driver, and an unstructured matrix with a check of 100 edges."""
    import ldpc_amd
    qc, H = build_code('nearEarth_mut_many')
    assert max(qc.row_degrees()) == 84
    orc = oracle_mod.Oracle(H)
    Y = make_llrs(31, 6.0, qc.n, 12)
    Y[3] = np.float32(-1.0)
    Y[5, :40] = 0.0
    with ldpc_amd.Decoder(qc) as eng, ldpc_amd.Decoder(H) as dense_eng:
        for e in (eng, dense_eng):                        # up to 128 edges per check: the table-driven kernel, state in LDS
            assert e.launch_info(prec=32)['block'] == 1024 and e.launch_info(prec=64)['lds_bytes'] > 8 * qc.n
        for prec, dt in ((32, np.float32), (64, np.float64)):
            for sm in (1, 0):
                hard, iters, conv, soft = orc.minsum(Y.astype(dt), 30, prec=prec, sign_mode=sm, threads=8)
                for e in (eng, dense_eng):
                    r = e.decode(Y, 30, prec=prec, sign_mode=sm, want_soft=True)
                    assert np.array_equal(r['iters'], iters) and np.array_equal(r['converged'], conv), (prec, sm)
                    assert np.array_equal(r['hard'], hard) and np.array_equal(r['soft'], soft)
                if sm == 1:
                    assert conv.sum() >= 8 and hard[conv == 1].sum() == 0
            hs, is_, cs, _ = orc.spa(Y.astype(dt), 8.0, 30, prec=prec, sign_mode=1, want_soft=False, threads=8)
            r = eng.decode((Y * np.float32(8.0)).astype(dt), 30, prec=prec, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=1)
            cv = cs == 1
            assert cv.sum() >= 8
            assert np.array_equal(r['iters'][cv], is_[cv]) and np.array_equal(r['hard'][cv], hs[cv])
        # re-used decoder (check-node messages carried from frame to frame)
        st = orc.new_state(32)
        hard, iters, conv, _ = orc.minsum(Y, 30, prec=32, sign_mode=1, want_soft=False, state=st)
        r = eng.decode(Y, 30, prec=32, sign_mode=1, chained=True)
        assert np.array_equal(r['iters'], iters) and np.array_equal(r['hard'], hard)
        cnt, pf = eng.eval_awgn([5.0, 7.0], 300, seed=3, sign_mode=1, per_frame=True)
        assert (cnt[:, 0] == 300).all() and cnt[1, 1] <= cnt[0, 1] and pf.shape == (2, 300, 4)
    # beyond 128 edges per check: block-circulant or not, from the edge list with one message per edge
    prng = np.random.RandomState(6)
    big = qc
    for j in range(16):
        big = big.with_block(0, j, sorted(prng.choice(511, 9 if j < 4 else 8, replace=False).tolist()))
    assert max(big.row_degrees()) == 132
    orc = oracle_mod.Oracle(big.to_dense())
    Y = make_llrs(32, 7.0, big.n, 6)
    with ldpc_amd.Decoder(big) as eng:
        assert eng.launch_info(prec=32)['block'] == 256 and eng.launch_info(prec=32)['lds_bytes'] == 0
        for prec, dt in ((32, np.float32), (64, np.float64)):
            hard, iters, conv, soft = orc.minsum(Y.astype(dt), 12, prec=prec, sign_mode=1, threads=8)
            r = eng.decode(Y, 12, prec=prec, sign_mode=1, want_soft=True)
            assert np.array_equal(r['iters'], iters) and np.array_equal(r['hard'], hard) and np.array_equal(r['soft'], soft)
    prng = np.random.RandomState(5)
    m, n = 60, 240
    Hr = np.zeros((m, n), np.int32)
    for c in range(n):
        Hr[prng.choice(m, 3, replace=False), c] = 1
    Hr[7, prng.choice(n, 100, replace=False)] = 1
    assert Hr.sum(1).max() >= 100
    orc = oracle_mod.Oracle(Hr)
    Y = make_llrs(9, 4.0, n, 33)
    with ldpc_amd.Decoder(Hr) as eng:
        for prec, dt in ((32, np.float32), (64, np.float64)):
            hard, iters, conv, soft = orc.minsum(Y.astype(dt), 25, prec=prec, sign_mode=1, threads=4)
            r = eng.decode(Y, 25, prec=prec, sign_mode=1, want_soft=True)
            assert np.array_equal(r['iters'], iters) and np.array_equal(r['hard'], hard) and np.array_equal(r['soft'], soft)


def test_heavy_circulants_on_the_message_plane_kernel(oracle_mod, monkeypatch):
    """Circulants of weight 3..8 (what a gym action with up to 7 hot bits installs, ldpc_env.py:293-317):
    float32 stays on the message-plane kernel, the shifts beyond a block's two lowest as extra edges
    through a global scratch row.  Bit-identical to the oracle for both sign rules, with heavy blocks
    in either block-row, two in one block-column, beside zeroed and light ones, at the limits (weight
    8, 8 extra edges in a block-row) -- and identical to the table-driven kernel, which takes over
    beyond those limits."""
    import ldpc_amd
    from ldpc_amd import codes
    base = codes.near_earth()
    prng = np.random.RandomState(511)
    hot = lambda w: sorted(prng.choice(511, w, replace=False).tolist())
    variants = [
        base.with_block(0, 5, hot(3)),
        base.with_block(0, 2, hot(4)).with_block(1, 13, hot(3)),                      # the limits of the kernel variant for lightly mutated codes
        base.with_block(1, 9, hot(7)).with_block(0, 12, hot(1)),
        base.with_block(0, 3, hot(4)).with_block(1, 3, hot(5)),                      # both blocks of one block-column
        base.with_block(0, 0, hot(8)).with_block(0, 15, []).with_block(1, 8, hot(3)).with_block(1, 2, [0]),
        base.with_block(1, 14, hot(6)).with_block(1, 1, hot(6)),                      # 8 extra edges in block-row 1
        base.with_block(0, 7, [0, 1, 510]).with_block(1, 7, [509, 510, 0, 1]),        # shifts that wrap for almost every row
    ]
    for qc in variants:
        H = qc.to_dense()
        orc = oracle_mod.Oracle(H)
        Y = make_llrs(4321, 3.4, qc.n, 24)
        Y[5, :64] = 0.0
        with ldpc_amd.Decoder(qc) as eng:
            assert eng.launch_info(prec=32)['lds_bytes'] == 163824, qc.row_degrees()      # the message-plane kernel
            for sm in (0, 1):
                for budget in (50, 2):
                    hard, iters, conv, soft = orc.minsum(Y, budget, prec=32, sign_mode=sm, threads=8)
                    r = eng.decode(Y, budget, prec=32, sign_mode=sm, want_soft=True)
                    assert np.array_equal(r['iters'], iters), (qc.row_degrees(), sm, budget, r['iters'], iters)
                    assert np.array_equal(r['converged'], conv) and np.array_equal(r['hard'], hard)
                    assert np.array_equal(r['soft'], soft)
            # sum-product on the same kernel family (parity unpinned by the reference: against this build's CPU statement,
            # on the frames that converge there)
            assert eng.launch_info(prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT)['lds_bytes'] == 163824
            scale = np.float32(4 * 10 ** 0.34)
            hs, is_, cs, _ = orc.spa(Y, float(scale), 50, prec=32, sign_mode=1, want_soft=False, threads=8)
            r = eng.decode(Y * scale, 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=1)
            cv = cs == 1
            assert cv.sum() >= 20, (qc.row_degrees(), int(cv.sum()))
            assert np.array_equal(r['iters'][cv], is_[cv]) and np.array_equal(r['hard'][cv], hs[cv]) and r['converged'][cv].all()
            fast = eng.decode(Y, 50, prec=32, sign_mode=1, want_soft=True)
            monkeypatch.setenv('LDPC_HIP_NO_PLANES', '1')
            assert eng.launch_info(prec=32)['lds_bytes'] != 163824
            slow = eng.decode(Y, 50, prec=32, sign_mode=1, want_soft=True)
            monkeypatch.delenv('LDPC_HIP_NO_PLANES')
            for k in ('iters', 'hard_packed', 'soft', 'converged', 'ones'):
                assert np.array_equal(fast[k], slow[k]), k
            # the sweep driver on the same kernel, channel values made in the kernel or read from a buffer
            c1, p1 = eng.eval_awgn([3.2, 3.6], 600, seed=9, per_frame=True, sign_mode=1)
            monkeypatch.setenv('LDPC_HIP_NO_FUSED_CHANNEL', '1')
            c2, p2 = eng.eval_awgn([3.2, 3.6], 600, seed=9, per_frame=True, sign_mode=1)
            monkeypatch.delenv('LDPC_HIP_NO_FUSED_CHANNEL')
            assert np.array_equal(c1, c2) and np.array_equal(p1, p2)
    # beyond the limits: 9 extra edges in one block-row, or weight 9 -> the table-driven kernel, same results
    for qc in (base.with_block(1, 14, hot(7)).with_block(1, 1, hot(6)), base.with_block(0, 4, hot(9))):
        orc = oracle_mod.Oracle(qc.to_dense())
        Y = make_llrs(99, 3.5, qc.n, 8)
        with ldpc_amd.Decoder(qc) as eng:
            assert eng.launch_info(prec=32)['lds_bytes'] != 163824
            hard, iters, conv, soft = orc.minsum(Y, 50, prec=32, sign_mode=1, threads=8)
            r = eng.decode(Y, 50, prec=32, sign_mode=1, want_soft=True)
            assert np.array_equal(r['iters'], iters) and np.array_equal(r['soft'], soft)


def test_release_pool_gives_back_what_closed_handles_left(oracle_mod):
    """A closed handle leaves its stream and allocations for the next one (the gym makes a decoder per step,
    ldpc_env.py:313, 370); ldpc_release_pool frees them, and decoding goes on as before afterwards."""
    import ldpc_amd
    qc, H = build_code('nearEarth_mut_w3')
    Y = make_llrs(12, 3.5, qc.n, 4)
    ref = oracle_mod.Oracle(H).minsum(Y, 50, prec=32, threads=4)
    for _ in range(3):
        with ldpc_amd.Decoder(qc) as eng:
            r = eng.decode(Y, 50, prec=32, want_soft=True)
            assert np.array_equal(r['iters'], ref[1]) and np.array_equal(r['soft'], ref[3])
    assert ldpc_amd.release_pool() >= 1
    assert ldpc_amd.release_pool() == 0
    with ldpc_amd.Decoder(qc) as eng:
        r = eng.decode(Y, 50, prec=32, want_soft=True)
        assert np.array_equal(r['iters'], ref[1]) and np.array_equal(r['soft'], ref[3])
