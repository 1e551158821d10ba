"""ldpc_decode_device_noalloc (scratch in a static device array, no hipMalloc) against
ldpc_decode_device, through the C ABI: identical outputs, two calls in a row, the second with a
smaller input; every kernel family, including the edge-list kernels whose per-workgroup
workspace lives in the array too."""
import ctypes

import numpy as np
import pytest

from conftest import build_code, make_llrs

pytestmark = pytest.mark.gpu


def _run(L, h, fn, algo, prec, Y, scale=1.0, budget=50, sign_mode=0):
    import torch
    dev = torch.device('cuda:0')
    F, n = Y.shape
    nbytes = (n + 7) // 8
    llr = torch.from_numpy(Y).to(dev)
    hard = torch.full((F, nbytes), 0xAA, dtype=torch.uint8, device=dev)
    iters = torch.full((F,), -7, dtype=torch.int32, device=dev)
    conv = torch.full((F,), 9, dtype=torch.uint8, device=dev)
    ones = torch.full((F,), -7, dtype=torch.int32, device=dev)
    soft = torch.full((F, n), float('nan'), dtype=torch.float64 if prec == 64 else torch.float32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = fn(h, algo, prec, sign_mode, budget, F, p(llr), 32, ctypes.c_float(scale), p(hard), p(iters), p(conv), p(ones), p(soft))
    assert rc == 0, L.ldpc_last_error().decode()
    assert L.ldpc_sync(h) == 0
    return [t.cpu().numpy() for t in (hard, iters, conv, ones, soft)]


@pytest.mark.parametrize('name,prec,algo', [('nearEarth', 32, 0), ('nearEarth', 64, 0), ('wifi_5_6', 32, 0),
                                           ('nearEarth_mut_w7', 32, 0), ('rand_24_48', 64, 0), ('rand_24_48', 64, 1),
                                           ('nearEarth', 32, 1)])
def test_noalloc_equals_decode_device(name, prec, algo):
    import ldpc_amd
    from ldpc_amd import _lib
    L = _lib.load()
    qc, H = build_code(name)
    n = H.shape[1]
    snr = 3.4 if name.startswith('nearEarth') else 3.0
    scale = float(4 * 10 ** (snr / 10)) if algo == 1 else 1.0
    big = make_llrs(2024, snr, n, 700 if n > 4000 else 1500)
    small = make_llrs(2025, snr, n, 37)
    sm = 0 if name in ('nearEarth',) else 1           # odd check degrees decode only with the corrected sign rule
    with ldpc_amd.Decoder(qc if qc is not None else H) as eng:
        h = eng._h
        # two allocation-free calls in a row (the second one smaller), then the allocating entry point
        got_big = _run(L, h, L.ldpc_decode_device_noalloc, algo, prec, big, scale, sign_mode=sm)
        got_small = _run(L, h, L.ldpc_decode_device_noalloc, algo, prec, small, scale, sign_mode=sm)
        ref_big = _run(L, h, L.ldpc_decode_device, algo, prec, big, scale, sign_mode=sm)
        ref_small = _run(L, h, L.ldpc_decode_device, algo, prec, small, scale, sign_mode=sm)
        for got, ref in ((got_big, ref_big), (got_small, ref_small)):
            for g, r in zip(got, ref):
                assert np.array_equal(g, r, equal_nan=True)
        assert ref_big[1].min() < ref_big[1].max()          # frames stop at different iterations
        # and once more after the allocating call: the array's old contents do not matter
        again = _run(L, h, L.ldpc_decode_device_noalloc, algo, prec, small, scale, sign_mode=sm)
        for g, r in zip(again, ref_small):
            assert np.array_equal(g, r, equal_nan=True)


def test_noalloc_rejects_what_does_not_fit():
    """A code whose edge-list workspace for a single workgroup exceeds the static array."""
    import ldpc_amd
    from ldpc_amd import _lib
    L = _lib.load()
    # 100000 x 200000, row degree 4 (not block-circulant): one workgroup needs 6.4 MB in float64
    m, n = 100000, 200000
    row_ptr = (np.arange(m + 1) * 4).astype(np.int32)
    col_idx = np.sort((np.arange(m)[:, None] * 2 + np.array([0, 1, 7, 12345])) % n, axis=1).astype(np.int32).ravel()
    h = ctypes.c_void_p()
    assert L.ldpc_create_csr(0, m, n, row_ptr.ctypes.data, col_idx.ctypes.data, ctypes.byref(h)) == 0
    try:
        import torch
        llr = torch.zeros((1, n), dtype=torch.float32, device='cuda:0')
        it = torch.zeros(1, dtype=torch.int32, device='cuda:0')
        rc = L.ldpc_decode_device_noalloc(h, 0, 64, 0, 5, 1, ctypes.c_void_p(llr.data_ptr()), 32, ctypes.c_float(1.0),
                                          None, ctypes.c_void_p(it.data_ptr()), None, None, None)
        assert rc == -3 and b'workspace' in L.ldpc_last_error()
        rc = L.ldpc_decode_device(h, 0, 64, 0, 5, 1, ctypes.c_void_p(llr.data_ptr()), 32, ctypes.c_float(1.0),
                                  None, ctypes.c_void_p(it.data_ptr()), None, None, None)
        assert rc == 0 and L.ldpc_sync(h) == 0
    finally:
        L.ldpc_destroy(h)
