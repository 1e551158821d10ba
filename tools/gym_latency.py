#!/usr/bin/env python3
"""Latency of one gym evaluation step (new parity matrix every step, 3 SNR points x 10 frames)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from ldpc_amd import codes, gym_eval
    prng = np.random.RandomState(1)
    ev = gym_eval.CodeEvaluator(seed=3)
    base = codes.near_earth()
    for kind in ('dense', 'qc'):
        ts = []
        for step in range(12):
            w = int(prng.randint(1, 4))
            qc = base.with_block(int(prng.randint(2)), int(prng.randint(16)), sorted(prng.choice(511, w, replace=False).tolist()))
            H = qc.to_dense() if kind == 'dense' else qc
            t0 = time.perf_counter()
            reward, _ = ev.step(H)
            ts.append(time.perf_counter() - t0)
        print(f'{kind}: per step ms', ' '.join(f'{t * 1e3:.1f}' for t in ts), ' median', f'{np.median(ts) * 1e3:.1f}')


if __name__ == '__main__':
    main()
