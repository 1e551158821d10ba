#!/usr/bin/env python3
"""Wall time of the sweep driver (channel + decode + counters) per SNR point."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import ldpc_amd
    from ldpc_amd import codes
    with ldpc_amd.Decoder(codes.near_earth()) as eng:
        eng.eval_awgn([3.5], 10000, seed=1)
        for F in (100000, 1000000):
            for pf in (False, True):
                ts = []
                for rep in range(4):
                    t0 = time.perf_counter()
                    cnt, _ = eng.eval_awgn([3.5], F, seed=2, per_frame=pf)
                    ts.append(time.perf_counter() - t0)
                t = min(ts)
                print(f'frames {F:8d} per_frame={pf!s:5s} {t * 1e3:8.1f} ms  {7154 * F / t / 1e9:6.2f} Gbit/s info  counters {cnt[0].tolist()}')


if __name__ == '__main__':
    main()
