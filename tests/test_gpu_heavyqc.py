"""GPU suite (-m gpu): the heavy-circulant kernel (qc_heavy.hip) -- float32 min-sum on the states a gym
episode reaches (every action leaves a circulant of up to 7 hot bits, ldpc_env.py:293-317), called
through the C ABI, bit for bit against the CPU oracle and against the table-driven kernel it replaces."""
import numpy as np
import pytest

from conftest import build_code, golden_points, make_llrs

pytestmark = pytest.mark.gpu


def heavy_lds_bytes(qc):
    """LDS bytes of the heavy-circulant kernel for this code (the layout of HeavyGeom, qc_heavy.h):
    how the tests tell which kernel a handle launches."""
    Z, mb, nb = qc.Z, qc.mb, qc.nb
    Zp = (Z + 63) // 64 * 64
    dpad = (max(qc.row_degrees()) + 3) // 4 * 4
    nwu = (dpad + 31) // 32
    split = nwu > 2
    rec_off = (nb * 2 * Z * 4 + Zp * 4 + 15) & ~15
    sw_off = rec_off + (1 if split else nwu) * mb * 2 * Z * 16
    flag_off = sw_off + (nwu * mb * 2 * Z * 4 if split else 0)
    return flag_off + 16 + 2 * mb * 32 * (2 if nwu <= 2 else 4) * 4


def episode(n_actions, seed=20211227):
    from ldpc_amd import codes
    return codes.gym_episode_state(codes.near_earth(), n_actions, seed=seed)


def check_against_oracle(eng, orc, Y, budgets=(50,), sign_modes=(0, 1)):
    for sm in sign_modes:
        for budget in budgets:
            hard, iters, conv, soft = orc.minsum(Y, budget, prec=32, sign_mode=sm, threads=8)
            r = eng.decode(Y, budget, prec=32, sign_mode=sm, want_soft=True)
            assert np.array_equal(r['iters'], iters), (sm, budget, r['iters'], iters)
            assert np.array_equal(r['converged'], conv), (sm, budget)
            assert np.array_equal(r['hard'], hard), (sm, budget)
            assert np.array_equal(r['ones'], hard.sum(1)), (sm, budget)
            assert np.array_equal(r['soft'], soft), (sm, budget)


@pytest.mark.parametrize('n_actions', [6, 12, 24, 40, 64])
def test_episode_states_against_oracle(oracle_mod, n_actions):
    """States after 6 ... 64 actions: checks of 37 ... 70 edges (one or two sign words ... three), every
    circulant weight from 0 to 7; frames that converge, frames that do not, a clean one, one with erasures."""
    import ldpc_amd
    qc = episode(n_actions)
    orc = oracle_mod.Oracle(qc.to_dense())
    Y = make_llrs(1000 + n_actions, 5.0, qc.n, 20)
    Y[3] = np.float32(-1.0)
    Y[5, :64] = 0.0
    Y[6] = make_llrs(7, 1.0, qc.n, 1)[0]
    with ldpc_amd.Decoder(qc) as eng:
        info = eng.launch_info(prec=32)
        if info['lds_bytes'] == 163824:
            pytest.skip('still within the message-plane kernels limits: %s' % (qc.row_degrees(),))
        assert info['lds_bytes'] == heavy_lds_bytes(qc) and info['block'] == 1024, (info, qc.row_degrees())
        check_against_oracle(eng, orc, Y, budgets=(50, 0, 1, 2, 7))


def test_golden_long_episode_state(oracle_mod, monkeypatch):
    """The golden long-episode state (checks of 84 and 55 edges: four sign words, the layout with the sign
    words apart): the float32 oracle, the table-driven kernel (LDPC_HIP_NO_HEAVYQC=1 at creation) and the
    allocation-free entry point."""
    import torch
    import ldpc_amd
    qc, H = build_code('nearEarth_mut_many')
    orc = oracle_mod.Oracle(H)
    pts = [p for p in golden_points('nearEarth_mut_many') if not p['special']]
    Y = np.concatenate([p['llr'][:6] for p in pts] + [make_llrs(77, 6.0, qc.n, 10)])
    with ldpc_amd.Decoder(qc) as eng:
        assert eng.launch_info(prec=32)['lds_bytes'] == heavy_lds_bytes(qc)
        check_against_oracle(eng, orc, Y, budgets=(50, 3))
        fast = eng.decode(Y, 50, prec=32, sign_mode=1, want_soft=True)
        monkeypatch.setenv('LDPC_HIP_NO_HEAVYQC', '1')
        with ldpc_amd.Decoder(qc) as old:
            assert old.launch_info(prec=32)['lds_bytes'] != heavy_lds_bytes(qc)
            slow = old.decode(Y, 50, prec=32, sign_mode=1, want_soft=True)
        monkeypatch.delenv('LDPC_HIP_NO_HEAVYQC')
        for k in ('iters', 'hard_packed', 'soft', 'converged', 'ones'):
            assert np.array_equal(fast[k], slow[k]), k
        # device entry points, with and without allocation, float64 channel values
        F = len(Y)
        d = torch.from_numpy(Y.astype(np.float64)).cuda()
        for noalloc in (False, True):
            hard = torch.zeros((F, eng.nbytes), dtype=torch.uint8, device='cuda')
            iters = torch.zeros(F, dtype=torch.int32, device='cuda')
            conv = torch.zeros(F, dtype=torch.uint8, device='cuda')
            ones = torch.zeros(F, dtype=torch.int32, device='cuda')
            eng.decode_device(d, F, 50, prec=32, sign_mode=1, llr_bits=64, d_hard=hard, d_iters=iters, d_conv=conv,
                              d_ones=ones, noalloc=noalloc)
            eng.sync()
            assert np.array_equal(hard.cpu().numpy(), fast['hard_packed']) and np.array_equal(iters.cpu().numpy(), fast['iters'])
            assert np.array_equal(conv.cpu().numpy(), fast['converged']) and np.array_equal(ones.cpu().numpy(), fast['ones'])


@pytest.mark.parametrize('name', ['nearEarth', 'nearEarth_zero_1_3', 'nearEarth_mut_w3', 'nearEarth_mut_w7'])
def test_forced_on_light_codes(oracle_mod, monkeypatch, name):
    """Codes the faster kernels take by default, pushed through this one (LDPC_HIP_FORCE_HEAVYQC=1): weights 0, 1, 2
    and a single heavy block; one sign word or two; rows with padding edges."""
    import ldpc_amd
    qc, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    monkeypatch.setenv('LDPC_HIP_FORCE_HEAVYQC', '1')
    with ldpc_amd.Decoder(qc) as eng:
        assert eng.launch_info(prec=32)['lds_bytes'] == heavy_lds_bytes(qc)
        for pt in golden_points(name):
            check_against_oracle(eng, orc, pt['llr'][:12], budgets=(50,))
        # normalised / offset min-sum
        Y = make_llrs(5, 3.3, qc.n, 8)
        for alpha, beta in ((0.75, 0.0), (1.0, 0.15)):
            eng.set_minsum_scaling(alpha, beta)
            hard, iters, conv, soft = orc.minsum(Y, 50, prec=32, sign_mode=1, alpha=alpha, beta=beta, threads=8)
            r = eng.decode(Y, 50, prec=32, sign_mode=1, want_soft=True)
            assert np.array_equal(r['iters'], iters) and np.array_equal(r['soft'], soft), (alpha, beta)
        eng.set_minsum_scaling(1.0, 0.0)


def test_small_codes_and_other_geometries(oracle_mod, monkeypatch):
    """Forced onto codes of other shapes: four and three block-rows, circulant sizes that are no multiple of 64,
    thread groups with different numbers of block-columns, ragged batches larger than the grid."""
    import ldpc_amd
    from ldpc_amd import codes
    prng = np.random.RandomState(99)
    monkeypatch.setenv('LDPC_HIP_FORCE_HEAVYQC', '1')
    for (Z, mb, nb, wmax) in ((81, 4, 24, 3), (37, 3, 10, 5), (128, 2, 9, 7), (200, 4, 13, 4)):
        blocks = []
        for b in range(mb):
            row = []
            for j in range(nb):
                w = int(prng.randint(0, wmax + 1))
                row.append(sorted(prng.choice(Z, w, replace=False).tolist()))
            if sum(len(x) for x in row) < 2:
                row[0] = [0, 1]
            blocks.append(row)
        qc = codes.QCCode(Z, blocks)
        orc = oracle_mod.Oracle(qc.to_dense())
        for F in (1, 5, 700):
            Y = make_llrs(int(prng.randint(1 << 20)), 4.0, qc.n, F)
            with ldpc_amd.Decoder(qc) as eng:
                assert eng.launch_info(prec=32)['lds_bytes'] == heavy_lds_bytes(qc), (Z, mb, nb)
                check_against_oracle(eng, orc, Y, budgets=(20,))


def test_sweep_driver_on_an_episode_state(monkeypatch):
    """ldpc_eval_awgn on a state the heavy-circulant kernel takes: counters and per-frame columns equal to
    the table-driven kernel's on the same noise; one launch for all points equal to one launch per point."""
    import ldpc_amd
    qc = episode(40)
    with ldpc_amd.Decoder(qc) as eng:
        assert eng.launch_info(prec=32)['lds_bytes'] == heavy_lds_bytes(qc)
        c1, p1 = eng.eval_awgn([3.0, 3.2, 3.4], 10, seed=11, per_frame=True)
        monkeypatch.setenv('LDPC_HIP_NO_POINT_BATCH', '1')
        c2, p2 = eng.eval_awgn([3.0, 3.2, 3.4], 10, seed=11, per_frame=True)
        monkeypatch.delenv('LDPC_HIP_NO_POINT_BATCH')
        c3, p3 = eng.eval_awgn([5.5, 6.5], 2000, seed=12, per_frame=True, sign_mode=1)
    monkeypatch.setenv('LDPC_HIP_NO_HEAVYQC', '1')
    with ldpc_amd.Decoder(qc) as old:
        assert old.launch_info(prec=32)['lds_bytes'] != heavy_lds_bytes(qc)
        c4, p4 = old.eval_awgn([3.0, 3.2, 3.4], 10, seed=11, per_frame=True)
        c5, p5 = old.eval_awgn([5.5, 6.5], 2000, seed=12, per_frame=True, sign_mode=1)
    assert np.array_equal(c1, c2) and np.array_equal(p1, p2)
    assert np.array_equal(c1, c4) and np.array_equal(p1, p4)
    assert np.array_equal(c3, c5) and np.array_equal(p3, p5)
    assert (c3[:, 0] == 2000).all()
