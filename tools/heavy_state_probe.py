#!/usr/bin/env python3
"""The state of a long gym episode (24 actions of 1...7 hot bits and a block-row of heavy circulants: checks of
84 and 55 edges, tests/conftest.py 'nearEarth_mut_many') on the table-driven block-circulant kernel: latency of
a gym step from such a state, throughput of a sweep point, and the same for a weight-3 mutation with the
message-plane kernels switched off (LDPC_HIP_TEST_HOOKS=1 LDPC_HIP_NO_PLANES=1)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import numpy as np
    import ldpc_amd
    import conftest
    from ldpc_amd import gym_eval
    qc, _ = conftest.build_code('nearEarth_mut_many')
    ev = gym_eval.CodeEvaluator(seed=3, deviceRng=True)
    ev.step(qc)
    ts = []
    for k in range(20):
        q2 = qc.with_block(1, k % 16, [1 + k, 77, 300])
        t0 = time.perf_counter()
        r, _ = ev.step(q2)
        ts.append(time.perf_counter() - t0)
    print('gym step from a state with checks of 84 / 5x edges: p50 %.2f ms, max %.2f ms' % (np.percentile(ts, 50) * 1e3, max(ts) * 1e3))
    for name in ('nearEarth_mut_many', 'nearEarth_mut_w3'):
        code, _ = conftest.build_code(name)
        with ldpc_amd.Decoder(code) as eng:
            for F, snr in ((40000, 6.0), (4000, 2.0)):
                eng.eval_awgn([snr], F, seed=1, sign_mode=1)
                t0 = time.perf_counter()
                cnt, _ = eng.eval_awgn([snr], F, seed=1, sign_mode=1)
                dt = time.perf_counter() - t0
                it = cnt[0, 3] / F
                print('%-20s %6d frames at %.1f dB: %7.1f ms, %6.2f Gbit/s info, mean iterations %4.1f, %5.2f us per frame-iteration per CU, launch %s'
                      % (name, F, snr, dt * 1e3, 7154 * F / dt / 1e9, it, dt * 1e6 * 256 / (F * max(it, 1.0)), eng.launch_info(prec=32)))


if __name__ == '__main__':
    main()
