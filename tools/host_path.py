#!/usr/bin/env python3
"""Rate of the host-buffer entry point (ldpc_decode: pageable NumPy arrays in, results out),
i.e. the PCIe-inclusive figure; the device-resident figure is bench.py's.  Times the C call
itself (no unpacking of the hard decisions on the Python side)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=200000)
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--threads', default='6,4,8,12')
    args = ap.parse_args()
    import numpy as np
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    from ldpc_amd.engine import _check
    eng = ldpc_amd.Decoder(codes.near_earth())
    F, n = args.frames, eng.n
    llr = torch.empty((F, n), dtype=torch.float32, device='cuda:0')
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.channel_awgn_device(llr, F, args.snr, seed=5)
    torch.cuda.synchronize()
    host = llr.cpu().numpy()
    del llr
    hard = np.zeros((F, eng.nbytes), np.uint8)
    iters = np.zeros(F, np.int32)
    conv = np.zeros(F, np.uint8)
    ones = np.zeros(F, np.int32)

    def run(label):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            _check(eng._L.ldpc_decode(eng._h, 0, 32, 0, 50, F, host.ctypes.data, 32, hard.ctypes.data,
                                      iters.ctypes.data, conv.ctypes.data, ones.ctypes.data, None))
            best = min(best, time.perf_counter() - t0)
        print(f'{label}: {F} frames in {best * 1e3:.1f} ms = {7154 * F / best / 1e9:.2f} Gbit/s info, '
              f'{F * n * 4 / best / 1e9:.1f} GB/s of channel values, mean iterations {iters.mean():.2f}', flush=True)

    for t in args.threads.split(','):
        os.environ['LDPC_HIP_COPY_THREADS'] = t
        run(f'pipelined, {t} copy threads')
    os.environ['LDPC_HIP_NO_PIPELINE'] = '1'
    run('single buffer')


if __name__ == '__main__':
    main()
