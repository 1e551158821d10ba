#!/usr/bin/env python3
"""The reference's evaluation grid on one GPU: NearEarth, plain min-sum, <= 50 iterations,
SNR 2.5:0.25:4.5 dB, `--frames` frames per point through the sweep driver (channel on the device,
decode, counters).  Prints one JSON object: per point BER and FER with their 95 % confidence intervals
(frame as the sampling unit), mean iterations, wall time."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=1000000)
    ap.add_argument('--seed', type=int, default=7134066)
    ap.add_argument('--prec', type=int, default=32)
    args = ap.parse_args()
    import numpy as np
    import ldpc_amd
    from ldpc_amd import codes, common
    snr = [round(2.5 + 0.25 * k, 2) for k in range(9)]
    rows = []
    with ldpc_amd.Decoder(codes.near_earth()) as eng:
        eng.eval_awgn([3.5], 10000, seed=1, prec=args.prec)
        t_all = time.perf_counter()
        for s in snr:
            t0 = time.perf_counter()
            cnt, pf = eng.eval_awgn([s], args.frames, seed=args.seed, prec=args.prec, per_frame=True)
            dt = time.perf_counter() - t0
            frames, ferr, berr, itsum, unc, notcw = [int(x) for x in cnt[0]]
            _, b_lo, b_hi = common.ber_interval(pf[0, :, 2], eng.n)
            rows.append(dict(snr_db=s, frames=frames, fer=ferr / frames, fer_ci95=list(common.wilson_interval(ferr, frames)),
                             ber=berr / (frames * eng.n), ber_ci95=[b_lo, b_hi],
                             uncoded_ber=unc / (frames * eng.n), mean_iters=itsum / frames,
                             not_codeword=notcw, wall_s=dt, info_gbps=7154 * frames / dt / 1e9))
            print(f'{s:5.2f} dB  FER {rows[-1]["fer"]:.3e}  BER {rows[-1]["ber"]:.3e}  iters {rows[-1]["mean_iters"]:6.2f}  '
                  f'{dt * 1e3:8.1f} ms', file=sys.stderr, flush=True)
        total = time.perf_counter() - t_all
    print(json.dumps(dict(code='NearEarth (8176,7154)', algo='min-sum', prec=args.prec, max_iters=50,
                          frames_per_point=args.frames, seed=args.seed, total_wall_s=total, points=rows)))


if __name__ == '__main__':
    main()
