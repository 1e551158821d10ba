"""Parity at scale (-m gpu): 10^5 NearEarth frames at 3.3 dB (frames that stop early, late and not at
all) through the C ABI against the CPU oracle, float32 and float64, frame for frame; the 10^6 frames at
3.5 dB of the headline figure in float32; and the
sum-product kernels against the CPU statement of the same precision on 2*10^4 frames per code.
Channel values come from the device generator (any values do: both sides decode the same array)."""
import numpy as np
import pytest

from conftest import build_code

pytestmark = pytest.mark.gpu

THREADS = 16


def _device_frames(eng, n_frames, snr_db, seed):
    import torch
    llr = torch.empty((n_frames, eng.n), dtype=torch.float32, device='cuda:0')
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.channel_awgn_device(llr, n_frames, snr_db, seed=seed)
    torch.cuda.synchronize()
    eng.set_stream(None)
    return llr.cpu().numpy()


def test_minsum_1e5_frames_bit_exact(oracle_mod):
    import ldpc_amd
    qc, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    F = 100_000
    with ldpc_amd.Decoder(qc) as eng:
        Y = _device_frames(eng, F, 3.3, 20261111)
        for prec, dt in ((32, np.float32), (64, np.float64)):
            hard, iters, conv, _ = orc.minsum(Y.astype(dt), 50, prec=prec, threads=THREADS, want_soft=False)
            r = eng.decode(Y, 50, prec=prec)
            assert np.array_equal(r['iters'], iters), (prec, int((r['iters'] != iters).sum()))
            assert np.array_equal(r['converged'], conv)
            assert np.array_equal(r['hard'], hard)
            assert np.array_equal(r['ones'], hard.sum(1, dtype=np.int64))
            # the batch holds what it is meant to hold: early, late and failed frames
            assert iters.min() <= 6 and 0.01 < (conv == 0).mean() < 0.6 and ((iters > 25) & (conv == 1)).sum() > 50
            del hard, r


def test_minsum_1e6_frames_bit_exact(oracle_mod):
    """The size the headline figure is quoted on: 10^6 NearEarth frames at 3.5 dB, <= 50 iterations, float32
    (the arithmetic of the reference's GPU path, ldpcCUDA.py:538-548), decoded where they are made -- on the
    device, through ldpc_decode_device -- against the CPU oracle on the same values, frame for frame: hard
    decisions, iteration counts, status, error counts.  Slices of 10^5 frames; the oracle (16 host threads, about
    40 s in all) is most of the test's time."""
    import torch
    import ldpc_amd
    qc, H = build_code('nearEarth')
    orc = oracle_mod.Oracle(H)
    F, S = 1_000_000, 100_000
    tot = dict(frames=0, iters=0, failed=0)
    with ldpc_amd.Decoder(qc) as eng:
        assert eng.launch_info(prec=32)['lds_bytes'] == 163824          # the message-plane kernel of the headline
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        llr = torch.empty((S, eng.n), dtype=torch.float32, device='cuda:0')
        d_hard = torch.empty((S, eng.nbytes), dtype=torch.uint8, device='cuda:0')
        d_iters = torch.empty(S, dtype=torch.int32, device='cuda:0')
        d_conv = torch.empty(S, dtype=torch.uint8, device='cuda:0')
        d_ones = torch.empty(S, dtype=torch.int32, device='cuda:0')
        for first in range(0, F, S):
            eng.channel_awgn_device(llr, S, 3.5, seed=460101, first_frame=first)
            eng.decode_device(llr, S, 50, prec=32, d_hard=d_hard, d_iters=d_iters, d_conv=d_conv, d_ones=d_ones)
            torch.cuda.synchronize()
            hard, iters, conv, _ = orc.minsum(llr.cpu().numpy(), 50, prec=32, threads=THREADS, want_soft=False)
            g_hard = np.unpackbits(d_hard.cpu().numpy(), axis=1, bitorder='little')[:, :eng.n]
            bad = np.flatnonzero((d_iters.cpu().numpy() != iters) | (d_conv.cpu().numpy() != conv) |
                                 (g_hard != hard).any(axis=1) | (d_ones.cpu().numpy() != hard.sum(1, dtype=np.int64)))
            assert bad.size == 0, (first, (first + bad[:10]).tolist())
            tot['frames'] += S
            tot['iters'] += int(iters.sum())
            tot['failed'] += int((conv == 0).sum())
            del hard, g_hard
        eng.set_stream(None)
    print(f"{tot['frames']} frames identical; mean iterations {tot['iters'] / F:.3f}, {tot['failed']} frames do not converge")
    assert tot['frames'] == F and 6.5 < tot['iters'] / F < 7.5 and 1000 < tot['failed'] < 5000


SPA_BAR = 0.9999          # identical (iterations, status, bits) on this fraction of ALL frames


@pytest.mark.parametrize('name,snr,sign_mode', [
    ('nearEarth', 3.3, 0), ('wifi_5_6', 4.0, 1), ('wifi_5_6', 3.0, 1),
    pytest.param('nearEarth', 3.1, 0, marks=pytest.mark.xfail(
        strict=False, reason='0.86 % of the frames never converge at 3.1 dB; on some of those the two 50-iteration trajectories part '
                             '(hardware exp2 / log2 / rcp on the device, libm on the host): measured 99.985 % identical, under the '
                             '99.99 % bar -- every frame that converges on the CPU is identical'))])
def test_spa_2e4_frames_against_cpu_statement(oracle_mod, name, snr, sign_mode):
    """PARITY UNPINNED by the reference (it has no sum-product rule): the bar is the stated one against
    this build's own CPU statement in the same precision -- identical (iterations, status, bits) on
    >= 99.99 % of ALL frames and on every frame that converges on the CPU.  The lower SNR points add frames
    that never converge, where the last bits of the transcendental functions can send the two 50-iteration
    trajectories apart; a point that misses the bar is marked as an expected failure with the measured figure,
    not given a lower bar.  The printed line also says what the agreement is when the CPU statement computes
    phi in the kernel's own three-branch form (oracle phi_form=1): the form is not what parts them."""
    import ldpc_amd
    qc, H = build_code(name)
    orc = oracle_mod.Oracle(H)
    F = 20_000
    scale = 4 * 10 ** (snr / 10)
    with ldpc_amd.Decoder(qc) as eng:
        Y = _device_frames(eng, F, snr, 777001)
        hard, iters, conv, _ = orc.spa(Y, scale, 50, prec=32, sign_mode=sign_mode, want_soft=False, threads=THREADS)
        r = eng.decode(Y * np.float32(scale), 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=sign_mode)
        h2, i2, c2, _ = orc.spa(Y, scale, 50, prec=32, sign_mode=sign_mode, want_soft=False, threads=THREADS, phi_form=1)
    ok = (r['iters'] == iters) & (r['converged'] == conv) & (r['hard'] == hard).all(axis=1)
    ok2 = (r['iters'] == i2) & (r['converged'] == c2) & (r['hard'] == h2).all(axis=1)
    cv = conv == 1
    print(f'{name} {snr} dB: {F} frames, CPU converged {int(cv.sum())}, identical {int(ok.sum())} '
          f'({ok.mean() * 100:.4f} %), identical among CPU-converged {ok[cv].mean() * 100:.4f} %, '
          f'mismatching frames {np.flatnonzero(~ok)[:20].tolist()}; against the CPU statement with phi in the '
          f'kernel\'s form {ok2.mean() * 100:.4f} %')
    assert cv.mean() > 0.5
    assert ok[cv].all(), (name, np.flatnonzero(~ok & cv)[:20])
    assert ok.mean() >= SPA_BAR, (name, snr, 'identical: %.4f %%' % (ok.mean() * 100), int((~ok).sum()), F)
