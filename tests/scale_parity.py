#!/usr/bin/env python3
"""Bit-exactness at scale: `--frames` frames at `--snr` dB decoded by the HIP library (float32 and
float64) and by the CPU oracle in the same precision; hard decisions, iteration counts and status
must be identical frame for frame.  (The parity tests do this on hundreds of frames; this is the
same check on 10^5.)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repository root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=100000)
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--threads', type=int, default=16)
    args = ap.parse_args()
    import numpy as np
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    import oracle as om
    om.build()
    qc = codes.near_earth()
    orc = om.Oracle(qc.to_dense())
    with ldpc_amd.Decoder(qc) as eng:
        F, n = args.frames, eng.n
        llr = torch.empty((F, n), dtype=torch.float32, device='cuda:0')
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.channel_awgn_device(llr, F, args.snr, seed=20261111)
        torch.cuda.synchronize()
        Y = llr.cpu().numpy()
        del llr
        ok = True
        for prec, dt in ((32, np.float32), (64, np.float64)):
            hard, iters, conv, _ = orc.minsum(Y.astype(dt), 50, prec=prec, threads=args.threads, want_soft=False)
            r = eng.decode(Y, 50, prec=prec)
            same = (np.array_equal(r['iters'], iters), np.array_equal(r['converged'], conv), np.array_equal(r['hard'], hard))
            print(f'prec {prec}: frames {F}, mean iterations {iters.mean():.3f}, not converged {int((conv == 0).sum())}, '
                  f'iterations/status/bits identical: {same}', flush=True)
            ok = ok and all(same)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
