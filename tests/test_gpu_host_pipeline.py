"""The host-buffer entry points switch to a two-slot pipeline (page-locked staging, uploads on
their own stream) from 128 MB of channel values on.  Its results must be those of the plain
single-buffer path, which the parity tests pin against the oracle: same frames through both."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(n, count, seed, snr_db=3.3):
    prng = np.random.RandomState(seed)
    sigma = np.sqrt(0.5 / 10 ** (snr_db / 10))
    return (-1.0 + prng.normal(0.0, sigma, (count, n))).astype(np.float32)


def _same(a, b, keys=('hard_packed', 'iters', 'converged', 'ones', 'soft')):
    for k in keys:
        assert np.array_equal(a[k], b[k]), k


def test_pipelined_host_decode_equals_plain_path(monkeypatch):
    import ldpc_amd
    from ldpc_amd import codes
    qc = codes.near_earth()
    F = 9001                                     # 294 MB of float32: five chunks, the last one ragged
    y = _frames(qc.n, F, 17)
    with ldpc_amd.Decoder(qc) as eng:
        piped = eng.decode(y, 50, prec=32, want_soft=True)
        piped2 = eng.decode(y[:6000], 50, prec=32)             # slots re-used by a second, shorter call
        monkeypatch.setenv('LDPC_HIP_NO_PIPELINE', '1')
        plain = eng.decode(y, 50, prec=32, want_soft=True)
    _same(piped, plain)
    for k in ('hard_packed', 'iters', 'converged', 'ones'):
        assert np.array_equal(piped2[k], plain[k][:6000]), k
    assert 0 < plain['iters'].mean() < 50 and plain['converged'].mean() > 0.5


def test_pipelined_chained_float64_equals_plain_path(monkeypatch):
    """A re-used decoder's check-node state passes from chunk to chunk (ldpc.py:356,377)."""
    import ldpc_amd
    from ldpc_amd import codes
    qc = codes.near_earth()
    F = 2300                                     # 150 MB of float64: two full chunks and a short one
    y = _frames(qc.n, F, 23, snr_db=3.0).astype(np.float64)
    with ldpc_amd.Decoder(qc) as eng:
        piped = eng.decode(y, 20, prec=64, chained=True, want_soft=True)
    with ldpc_amd.Decoder(qc) as eng:
        monkeypatch.setenv('LDPC_HIP_NO_PIPELINE', '1')
        plain = eng.decode(y, 20, prec=64, chained=True, want_soft=True)
    _same(piped, plain)
