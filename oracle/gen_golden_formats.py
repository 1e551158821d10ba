#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/formats.npz from the reference's own
common.py (imported read-only from /root/reference in the build container; never runs on the
GPU box).  Pins the data formats either side of the decode path:
  * the gym environment's 2048-byte observation -> parity matrix (common.uncompress,
    common.py:349-365) on the observation the reference's own test holds (common.py:368-377);
  * the reward fit (common.recursiveLinearFit, common.py:293-303) on synthetic scatter data.
  * the hexadecimal naming and the .mat workspace of a saved code (fileHandler.py:36-135,
    183-231) on the NearEarth matrix the reference reads from its own parity file.
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


def code_instance_vectors():
    """Names, hex strings and workspace layout the reference's fileHandler produces."""
    import contextlib
    import io
    import tempfile
    import scipy.io
    import fileHandler as ref_fh
    import common as ref_common
    H = ref_fh.readMatrixFromFile(REF + '/codeMatrices/nearEarthParity.txt', 1022, 8176, 511, True, False, False)
    hex_name = ref_fh.binaryMatrixToHexString(H, 511)
    bits = np.random.RandomState(11).randint(0, 2, 96)
    digits, text = ref_fh.binaryArraytoHex(bits)
    back = ref_fh.hexStringToBinaryArray(text + 'g\n')
    circ = ref_fh.hexToCirculant(hex_name[:128], 511)
    stats = ref_common.berStatistics()
    for k in range(6):
        stats.addEntry(3.0 + 0.2 * (k % 3), 0.5, 0.49, 0.02, 0.001 * k, 5 + k, 50, 'test')
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        returned = ref_fh.saveCodeInstance(H, 511, 8176, evaluationData=stats, path=tmp + '/', evaluationTime=1.5,
                                           numberOfNonZero=int(H.sum()))
        files = sorted(os.listdir(tmp))
        ws = scipy.io.loadmat(os.path.join(tmp, files[0]))
    keys = sorted(k for k in ws if not k.startswith('__'))
    return dict(inst_hex_sha256=hashlib.sha256(hex_name.encode()).hexdigest(), inst_hex_head=hex_name[:256],
                inst_returned_is_hex=bool(returned == hex_name), inst_file=files[0],
                inst_keys=np.array(keys), inst_shapes=np.array([str(ws[k].shape) for k in keys]),
                inst_dtypes=np.array([str(ws[k].dtype) for k in keys]),
                inst_ber=ws['berData'], inst_snr=ws['snrData'], inst_itr=ws['itrData'], inst_avg_snr=ws['averageSnrAxis'],
                hex_bits=bits.astype(np.uint8), hex_text=text, hex_digits=digits, hex_back=back.astype(np.uint8),
                circ_row0=circ[0].astype(np.uint8), circ_col0=circ[:, 0].astype(np.uint8))


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
    inst = code_instance_vectors()
    np.savez_compressed(os.path.join(OUT, 'formats.npz'), **inst,
                        observation=obs, row0=H[0].astype(np.uint8), row511=H[511].astype(np.uint8),
                        block_hot=np.array(hot, np.int32), block_ptr=np.array(ptr, np.int32), nnz=int(H.sum()),
                        sha=np.frombuffer(hashlib.sha256(np.packbits(H.astype(np.uint8)).tobytes()).digest(), np.uint8),
                        dtype=str(H.dtype), fit_x=xs, fit_y=ys, fit_p=np.asarray(p, np.float64),
                        fit_kept=np.int32(len(snr)), fit_itr=np.int32(itr))
    print('formats.npz: nnz', int(H.sum()), 'dtype', H.dtype, 'fit', p, 'kept', len(snr))


if __name__ == '__main__':
    main()
