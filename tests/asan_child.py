#!/usr/bin/env python3
"""Child of tests/test_sanitizers.py: runs with the AddressSanitizer / UndefinedBehaviorSanitizer run-times
preloaded and LDPC_ORACLE_SO pointing at the sanitizer build of the oracle; exercises the oracle on golden
vectors (both precisions, both rules, the re-used decoder, several threads) and the library's host-only
helper ldpc_dense_to_qc on matrices that are block-circulant, almost, and not at all.  Any sanitizer report
aborts the process."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import numpy as np

from conftest import build_code, golden_points, make_llrs
import oracle as om


def oracle_part():
    for name in ('nearEarth', 'nearEarth_mut_w7', 'wifi_5_6', 'rand_24_48'):
        qc, H = build_code(name)
        orc = om.Oracle(H)
        for pt in list(golden_points(name))[:3]:
            Y = pt['llr'][:4]
            hard, iters, conv, soft = orc.minsum(Y.astype(np.float64), pt['max_iters'], prec=64, threads=4)
            assert np.array_equal(iters, pt['iters'][:4]) and np.array_equal(np.packbits(hard, axis=1, bitorder='little'), pt['hard'][:4])
            orc.minsum(Y, 7, prec=32, sign_mode=1, threads=3, alpha=0.8, beta=0.05)
            st = orc.new_state(32)
            orc.minsum(Y, 5, prec=32, state=st)
        y = make_llrs(3, 4.0, H.shape[1], 3)
        orc.spa(y, 10.0, 6, prec=64, threads=2)
        orc.spa(y, 10.0, 6, prec=32, sign_mode=1)
        orc.minsum(np.zeros((0, H.shape[1]), np.float32), 5, prec=32)       # empty batch
    # degenerate rows (degree 1 and 0 are not decodable by the reference; the oracle must not read out of bounds)
    Hs = np.zeros((3, 6), np.int32)
    Hs[0, :4] = 1
    Hs[1, 2:] = 1
    Hs[2, [0, 5]] = 1
    om.Oracle(Hs).minsum(make_llrs(1, 2.0, 6, 2), 4, prec=64)


def helper_part():
    L = ctypes.CDLL(os.environ['LDPC_HOST_SO'])
    L.ldpc_dense_to_qc.restype = ctypes.c_int
    L.ldpc_dense_to_qc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]

    def run(H, Z, cap):
        m, n = H.shape
        ptr = np.zeros((m // Z) * (n // Z) + 1, np.int32)
        sh = np.zeros(max(cap, 1), np.int32)
        rc = L.ldpc_dense_to_qc(H.ctypes.data, H.dtype.itemsize, m, n, Z, ptr.ctypes.data, sh.ctypes.data, cap)
        return rc, ptr, sh
    qc, H = build_code('nearEarth_mut_w7')
    for dt in (np.int8, np.int32, np.int64, np.float64):
        M = np.ascontiguousarray(H.astype(dt))
        rc, ptr, sh = run(M, 511, 64 * 32)
        assert rc == 0 and ptr[-1] == sum(len(b) for row in qc.blocks for b in row)
        flat = [tuple(int(x) for x in sh[ptr[k]:ptr[k + 1]]) for k in range(32)]
        assert flat == [tuple(sorted(b)) for row in qc.blocks for b in row]
    M = np.ascontiguousarray(H.astype(np.int32))
    assert run(M, 511, 3)[0] < 0                       # too small a shifts array: refused, nothing written past it
    M[700, 4000] ^= 1
    assert run(M, 511, 64 * 32)[0] == 1                # one entry off: not block-circulant
    assert run(np.ones((14, 21), np.int32), 7, 7 * 6)[0] == 0           # all-ones blocks are circulants of weight Z
    assert run(np.eye(8, dtype=np.int8), 4, 16)[0] == 0
    assert run(np.zeros((4, 4), np.int8), 4, 0)[0] == 0
    _, Hr = build_code('rand_24_48')
    assert run(np.ascontiguousarray(Hr), 12, 4096)[0] == 1


if __name__ == '__main__':
    oracle_part()
    helper_part()
    print('SANITIZED OK')
