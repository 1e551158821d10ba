"""The raw ctypes binding a maintainer of the reference would write against include/ldpc_hip.h
(the one shown in the integration notes), exercised without the ldpc_amd engine wrapper."""
import ctypes

import numpy as np
import pytest

from conftest import build_code, golden_points

pytestmark = pytest.mark.gpu


def test_raw_ctypes_binding_matches_reference_golden():
    import ldpc_amd
    from ldpc_amd import _lib
    _lib.load()        # in a process that also imports torch, the HIP runtime must be the one torch ships
    L = ctypes.CDLL(ldpc_amd.library_path())
    L.ldpc_last_error.restype = ctypes.c_char_p
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64

    def check(rc):
        if rc:
            raise RuntimeError(L.ldpc_last_error().decode())

    _, H = build_code('nearEarth')
    rows, cols = np.nonzero(H)
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=H.shape[0]))]).astype(np.int32)
    col_idx = cols.astype(np.int32)
    h = vp()
    check(L.ldpc_create_csr(i32(0), i32(H.shape[0]), i32(H.shape[1]), vp(row_ptr.ctypes.data),
                            vp(col_idx.ctypes.data), ctypes.byref(h)))
    n = H.shape[1]
    pt = next(iter(golden_points('nearEarth')))
    for k in range(3):
        y = np.ascontiguousarray(pt['llr'][k], np.float64).reshape(1, n)
        hard = np.zeros((1, (n + 7) // 8), np.uint8)
        soft = np.zeros((1, n), np.float64)
        iters = np.zeros(1, np.int32)
        conv = np.zeros(1, np.uint8)
        ones = np.zeros(1, np.int32)
        check(L.ldpc_decode(h, i32(0), i32(64), i32(0), i32(int(pt['max_iters'])), i64(1), vp(y.ctypes.data), i32(64),
                            vp(hard.ctypes.data), vp(iters.ctypes.data), vp(conv.ctypes.data),
                            vp(ones.ctypes.data), vp(soft.ctypes.data)))
        assert iters[0] == pt['iters'][k] and conv[0] == pt['conv'][k]
        assert np.array_equal(hard[0], pt['hard'][k])
    snr = np.array([3.0, 3.6], np.float64)
    cnt = np.zeros((2, 6), np.uint64)
    check(L.ldpc_eval_awgn(h, i32(0), i32(32), i32(0), i32(50), i32(2), vp(snr.ctypes.data), i64(64),
                           ctypes.c_uint64(5), ctypes.c_uint64(0), vp(cnt.ctypes.data), None))
    assert cnt[:, 0].tolist() == [64, 64] and cnt[1, 1] <= cnt[0, 1]
    assert L.ldpc_decode(h, i32(7), i32(64), i32(0), i32(5), i64(1), vp(y.ctypes.data), i32(64),
                         None, vp(iters.ctypes.data), vp(conv.ctypes.data), None, None) == -1
    assert b'algo' in L.ldpc_last_error()
    L.ldpc_destroy.argtypes = [vp]
    L.ldpc_destroy(h)
