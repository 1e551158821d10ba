#!/usr/bin/env python3
"""Frame-for-frame parity at the size the headline figure is quoted on: 10^6 NearEarth frames at 3.5 dB, <= 50
iterations, float32 and float64, the GPU decode against the CPU oracle on the same channel values (made on the device
by the counter-based generator, slice after slice, so that the host never holds more than one slice).  Run by name on
the GPU box (about five minutes of 16 host threads; too long for the collected suite, whose scale test holds 10^5):

    python -u tests/full_size_parity.py --out gpurun_out/full_size_parity.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=1_000_000)
    ap.add_argument('--slice', type=int, default=100_000)
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--seed', type=int, default=460101)
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import torch
    import ldpc_amd
    from conftest import build_code
    from oracle import oracle as om
    qc, H = build_code('nearEarth')
    orc = om.Oracle(H)
    out = {'frames': args.frames, 'snr_db': args.snr, 'seed': args.seed, 'max_iters': 50, 'precisions': {}}
    with ldpc_amd.Decoder(qc) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        llr = torch.empty((args.slice, eng.n), dtype=torch.float32, device='cuda:0')
        for prec, dt in ((32, np.float32), (64, np.float64)):
            st = dict(frames=0, identical_frames=0, iterations=0, frame_errors=0, bit_errors=0, not_converged=0,
                      oracle_seconds=0.0, max_iterations_seen=0)
            for first in range(0, args.frames, args.slice):
                F = min(args.slice, args.frames - first)
                eng.channel_awgn_device(llr, F, args.snr, seed=args.seed, first_frame=first)
                torch.cuda.synchronize()
                Y = llr[:F].cpu().numpy()
                t0 = time.time()
                hard, iters, conv, _ = orc.minsum(Y.astype(dt), 50, prec=prec, threads=args.threads, want_soft=False)
                st['oracle_seconds'] += time.time() - t0
                r = eng.decode(Y, 50, prec=prec)
                same = (r['iters'] == iters) & (r['converged'] == conv) & (r['hard'] == hard).all(axis=1) & \
                       (r['ones'] == hard.sum(1, dtype=np.int64))
                st['frames'] += F
                st['identical_frames'] += int(same.sum())
                st['iterations'] += int(iters.sum())
                st['frame_errors'] += int((hard.sum(1) != 0).sum())
                st['bit_errors'] += int(hard.sum())
                st['not_converged'] += int((conv == 0).sum())
                st['max_iterations_seen'] = max(st['max_iterations_seen'], int(iters[conv == 1].max()))
                print(f'float{prec}: frames {st["frames"]:8d} identical {st["identical_frames"]:8d} '
                      f'oracle {st["oracle_seconds"]:6.1f} s', flush=True)
                if not same.all():
                    print('MISMATCH at frames', (first + np.flatnonzero(~same)[:10]).tolist(), flush=True)
            out['precisions'][f'float{prec}'] = st
    out['result'] = ('identical frame for frame' if all(s['frames'] == s['identical_frames']
                                                        for s in out['precisions'].values()) else 'MISMATCH')
    print(json.dumps(out), flush=True)
    if args.out:
        json.dump(out, open(args.out, 'w'), indent=1)
    return 0 if out['result'].startswith('identical') else 1


if __name__ == '__main__':
    sys.exit(main())
