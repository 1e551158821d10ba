#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/formats.npz from the reference's own
common.py (imported read-only from /root/reference in the build container; never runs on the
GPU box).  Pins the data formats either side of the decode path:
  * the gym environment's 2048-byte observation -> parity matrix (common.uncompress,
    common.py:349-365) on the observation the reference's own test holds (common.py:368-377);
  * the reward fit (common.recursiveLinearFit, common.py:293-303) on synthetic scatter data.
Stored: the observation bytes, row 0 and row 511 of the resulting matrix, the hot columns of
row 0 of every 511x511 block, nnz, sha256 of packbits(H); the fit's inputs and coefficients.
"""
import hashlib
import inspect
import os
import re
import sys

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.path.insert(0, REF)
    import common as ref_common
    H = ref_common.test_uncompress()
    # the observation is a literal inside the reference's test function: recover the DATA from it
    text = inspect.getsource(ref_common.test_uncompress)
    obs = np.array(re.search(r"'\[(.*?)\]'", text, re.S).group(1).split(), dtype=np.uint8)
    assert obs.size == 2048 and np.array_equal(ref_common.uncompress(obs), H)
    Z = 511
    hot, ptr = [], [0]
    for b in range(2):
        for j in range(16):
            cols = np.flatnonzero(H[b * Z, j * Z:(j + 1) * Z])
            hot.extend(cols.tolist())
            ptr.append(len(hot))
    prng = np.random.RandomState(5)
    xs = np.repeat(np.array([3.0, 3.2, 3.4, 3.6]), 12).astype(np.float32)
    ys = np.maximum(0, 0.02 * (3.5 - xs) + prng.normal(0, 0.002, xs.size)).astype(np.float32)
    ys[xs > 3.5] = 0
    snr, ber, p, trendP, itr = ref_common.recursiveLinearFit(xs, ys)
    np.savez_compressed(os.path.join(OUT, 'formats.npz'),
                        observation=obs, row0=H[0].astype(np.uint8), row511=H[511].astype(np.uint8),
                        block_hot=np.array(hot, np.int32), block_ptr=np.array(ptr, np.int32), nnz=int(H.sum()),
                        sha=np.frombuffer(hashlib.sha256(np.packbits(H.astype(np.uint8)).tobytes()).digest(), np.uint8),
                        dtype=str(H.dtype), fit_x=xs, fit_y=ys, fit_p=np.asarray(p, np.float64),
                        fit_kept=np.int32(len(snr)), fit_itr=np.int32(itr))
    print('formats.npz: nnz', int(H.sum()), 'dtype', H.dtype, 'fit', p, 'kept', len(snr))


if __name__ == '__main__':
    main()
