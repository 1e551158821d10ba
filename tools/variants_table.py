#!/usr/bin/env python3
"""The 32 circulant-zeroed NearEarth variants of the reference's test-matrix script
(testMatricesGeneratorScript.py:23-34: block (b, j) zeroed, b in 0..1, j in 0..15), each
evaluated at 3.0 / 3.2 / 3.4 dB on the same noise, plus the unmodified code: per variant FER, BER
and mean iterations per SNR point and the wall time of the whole table.  One process per GPU
(python -m torch.distributed.run ...) splits the variants over the ranks; a single process
evaluates all of them.  Prints one JSON object on rank 0."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=100000)
    ap.add_argument('--seed', type=int, default=7134066)
    args = ap.parse_args()
    from ldpc_amd import codes, distributed as D
    D.init()
    rank, _, ws = D.world()
    snr = [3.0, 3.2, 3.4]
    names = ['NearEarth'] + [f'zeroed ({b},{j})' for b in range(2) for j in range(16)]
    variants = [codes.near_earth()] + [codes.near_earth_zeroed(b, j) for b in range(2) for j in range(16)]
    D.evaluate_variants(variants[:ws], snr, 1000, args.seed)          # warm-up: handles, kernels
    t0 = time.perf_counter()
    cnt = D.evaluate_variants(variants, snr, args.frames, args.seed)
    dt = time.perf_counter() - t0
    if rank == 0:
        n = variants[0].n
        rows = []
        for v, name in enumerate(names):
            c = cnt[v].astype(float)
            rows.append(dict(variant=name, fer=(c[:, 1] / c[:, 0]).tolist(), ber=(c[:, 2] / (c[:, 0] * n)).tolist(),
                             mean_iters=(c[:, 3] / c[:, 0]).tolist()))
        print(json.dumps(dict(snr_db=snr, frames_per_point=args.frames, seed=args.seed, ranks=ws, variants=len(variants),
                              wall_s=dt, frames_per_s=len(variants) * len(snr) * args.frames / dt, table=rows)))


if __name__ == '__main__':
    main()
