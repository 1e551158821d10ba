#!/usr/bin/env python3
"""Latency of one gym evaluation step, new parity matrix -> reward (LdpcEnv.step's evaluation part,
ldpc_env.py:217-226, 347-371): 3 SNR points x 10 frames, 50 iterations, a different mutated matrix
every step, handed over as the dense matrix (what the environment holds), as shift lists, or as the
2048-byte observation.  Prints one JSON object: p50 / p99 / mean per kind, and where the time goes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import ldpc_amd
    from ldpc_amd import codes, common, gym_eval
    from ldpc_amd import ldpc as mirror
    prng = np.random.RandomState(1)
    base = codes.near_earth()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out = {'steps': steps, 'snr_points': [3.0, 3.2, 3.4], 'frames_per_point': 10, 'kinds': {}}
    cands = []
    for _ in range(steps):
        w = int(prng.randint(1, 8))                       # up to 7 hot bits (models.py:296, 368)
        cands.append(base.with_block(int(prng.randint(2)), int(prng.randint(16)), sorted(prng.choice(511, w, replace=False).tolist())))
    for kind in ('dense', 'qc', 'observation'):
        for rng in (False, True):
            ev = gym_eval.CodeEvaluator(seed=3, deviceRng=rng)
            ev.step(base)                                  # library loaded, kernels resident
            ts = []
            for qc in cands:
                if kind == 'dense':
                    H = qc.to_dense()
                elif kind == 'observation':
                    H = common.compress(qc.to_dense())
                else:
                    H = qc
                t0 = time.perf_counter()
                ev.step(H)
                ts.append(time.perf_counter() - t0)
            ts = np.array(ts) * 1e3
            out['kinds'][f'{kind}, {"device" if rng else "host (reference)"} noise'] = {
                'p50_ms': float(np.percentile(ts, 50)), 'p99_ms': float(np.percentile(ts, 99)), 'mean_ms': float(ts.mean())}
    # An episode as the environment runs it: every action's circulant stays (ldpc_env.py:293-317), so the state
    # grows heavier step by step -- past the message-plane kernels' limits after a few actions, to checks of
    # 60 ... 100 edges after 40.
    ev = gym_eval.CodeEvaluator(seed=3, deviceRng=True)
    ev.step(base)
    ts, degs = [], []
    for ep in range(3):
        state = base
        for k in range(40):
            w = int(prng.randint(1, 8))
            state = state.with_block(int(prng.randint(2)), int(prng.randint(16)), sorted(prng.choice(511, w, replace=False).tolist()))
            t0 = time.perf_counter()
            ev.step(state)
            ts.append(time.perf_counter() - t0)
            degs.append(max(state.row_degrees()))
    ts = np.array(ts) * 1e3
    degs = np.array(degs)
    out['episode'] = {'what': '3 episodes of 40 actions from NearEarth, shift lists, device noise; the state keeps every action',
                      'p50_ms': float(np.percentile(ts, 50)), 'p99_ms': float(np.percentile(ts, 99)), 'mean_ms': float(ts.mean()),
                      'mean_ms_first_5_actions': float(ts.reshape(3, 40)[:, :5].mean()),
                      'mean_ms_last_10_actions': float(ts.reshape(3, 40)[:, 30:].mean()),
                      'largest_check_degree_at_the_end': int(degs.reshape(3, 40)[:, -1].max())}
    # where the time goes for a dense matrix
    H = cands[0].to_dense()
    t = {}
    t0 = time.perf_counter(); qc = codes.detect_qc(H); t['detect block-circulant structure'] = time.perf_counter() - t0
    t0 = time.perf_counter(); eng = ldpc_amd.Decoder(qc); t['create handle (tables to the device)'] = time.perf_counter() - t0
    Y = np.zeros((30, eng.n), np.float32) - 1
    eng.decode(Y, 50, prec=32, want_hard=False)
    t0 = time.perf_counter(); eng.decode(Y, 50, prec=32, want_hard=False); t['decode 30 frames (host entry point)'] = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.eval_awgn([3.0, 3.2, 3.4], 10, seed=1, per_frame=True); t['sweep 3 x 10 frames (device noise)'] = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.close(); t['destroy handle'] = time.perf_counter() - t0
    out['breakdown_ms'] = {k: v * 1e3 for k, v in t.items()}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
