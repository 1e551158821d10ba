#!/usr/bin/env python3
"""One sweep point through ldpc_eval_awgn (channel values made inside the decode kernel) for rocprofv3."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=1000000)
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--reps', type=int, default=1)
    args = ap.parse_args()
    import ldpc_amd
    from ldpc_amd import codes
    with ldpc_amd.Decoder(codes.near_earth()) as eng:
        for _ in range(args.reps):
            cnt, _ = eng.eval_awgn([args.snr], args.frames, seed=2)
        print('counters', cnt[0].tolist())


if __name__ == '__main__':
    main()
