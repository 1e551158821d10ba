#!/usr/bin/env python3
"""What a gym step from a late episode state is made of (NearEarth after 40 actions, 3 SNR points x 10 frames,
50 iterations; ldpc_env.py:353-377): handle creation, the sweep call, the reward, handle destruction -- medians
over 60 steps, each on a different state (one more action applied)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import ldpc_amd
    from ldpc_amd import codes, gym_eval
    base = codes.gym_episode_state(codes.near_earth(), 40, seed=20211227)
    prng = np.random.RandomState(1)
    parts = {k: [] for k in ('create', 'sweep', 'reward', 'destroy', 'total')}
    ev = gym_eval.CodeEvaluator(seed=3, deviceRng=True)
    ev.step(base)
    for k in range(60):
        qc = base.with_block(int(prng.randint(2)), int(prng.randint(16)), sorted(prng.choice(511, int(prng.randint(1, 8)), replace=False).tolist()))
        t0 = time.perf_counter()
        eng = ldpc_amd.Decoder(qc)
        t1 = time.perf_counter()
        cnt, pf = eng.eval_awgn([3.0, 3.2, 3.4], 10, seed=5 + k, per_frame=True)
        t2 = time.perf_counter()
        t3 = time.perf_counter()
        eng.close()
        t4 = time.perf_counter()
        ta = time.perf_counter()
        ev.step(qc)
        tb = time.perf_counter()
        for name, v in (('create', t1 - t0), ('sweep', t2 - t1), ('destroy', t4 - t3), ('total', tb - ta)):
            parts[name].append(v)
    for name, v in parts.items():
        if v:
            print('%-8s p50 %.3f ms  p90 %.3f ms' % (name, np.percentile(v, 50) * 1e3, np.percentile(v, 90) * 1e3))
    print('(total = CodeEvaluator.step: create + sweep + reward + destroy)')


if __name__ == '__main__':
    main()
