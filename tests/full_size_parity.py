#!/usr/bin/env python3
"""Frame-for-frame parity at the size the headline figure is quoted on: 10^6 NearEarth frames at 3.5 dB, <= 50
iterations, float32 and float64, the GPU decode against the CPU oracle on the same channel values (made on the device
by the counter-based generator, slice after slice, so that the host never holds more than one slice).  Run by name on
the GPU box (about five minutes of 16 host threads; too long for the collected suite, whose scale test holds 10^5):

    python -u tests/full_size_parity.py --out gpurun_out/full_size_parity.json
    python -u tests/full_size_parity.py --codes nearEarth_mut_w3:3.5,wifi_1/2:2.0 --frames64 200000 --out ...

--codes: name:snr pairs; names are those of the golden set (tests/conftest.py build_code), wifi_<rate> for the four
802.11n n = 1944 tables and episode<k> for NearEarth after k gym actions (ldpc_env.py:293-317; the states of
tools/bench_episode.py, which the heavy-circulant kernel decodes).
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
    ap.add_argument('--frames64', type=int, default=None, help='frames of the float64 pass (default: as float32)')
    ap.add_argument('--codes', default=None, help='name:snr[,name:snr...] instead of NearEarth at --snr')
    ap.add_argument('--slice', type=int, default=100_000)
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--seed', type=int, default=460101)
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--sign-mode', type=int, default=0, help='0: the sign rule of the reference (ldpc.py:135-141), 1: corrected '
                                                             '(rows of odd degree; the rule the 802.11n figures use)')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import torch
    import ldpc_amd
    from conftest import build_code
    from oracle import oracle as om
    from ldpc_amd import codes as codes_mod
    todo = [('nearEarth', args.snr)] if not args.codes else [(c.rsplit(':', 1)[0], float(c.rsplit(':', 1)[1]))
                                                            for c in args.codes.split(',')]
    out = {'frames': args.frames, 'frames_float64': args.frames64 or args.frames, 'seed': args.seed, 'max_iters': 50,
           'sign_mode': args.sign_mode, 'codes': {}}
    for name, snr in todo:
        if name.startswith('wifi_') and '/' in name:
            qc = codes_mod.wifi_1944(name.split('_', 1)[1])
            H = qc.to_dense()
        elif name.startswith('episode'):
            qc = codes_mod.gym_episode_state(codes_mod.near_earth(), int(name[len('episode'):]), seed=20211227)
            H = qc.to_dense()
        else:
            qc, H = build_code(name)
        orc = om.Oracle(H)
        res = {'snr_db': snr, 'precisions': {}}
        with ldpc_amd.Decoder(qc) as eng:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            res['kernel_float32'] = eng.launch_info(prec=32)
            llr = torch.empty((args.slice, eng.n), dtype=torch.float32, device='cuda:0')
            for prec, dt, total in ((32, np.float32, args.frames), (64, np.float64, args.frames64 or args.frames)):
                st = dict(frames=0, identical_frames=0, iterations=0, frame_errors=0, bit_errors=0, not_converged=0,
                          oracle_seconds=0.0)
                for first in range(0, total, args.slice):
                    F = min(args.slice, total - first)
                    eng.channel_awgn_device(llr, F, snr, seed=args.seed, first_frame=first)
                    torch.cuda.synchronize()
                    Y = llr[:F].cpu().numpy()
                    t0 = time.time()
                    hard, iters, conv, _ = orc.minsum(Y.astype(dt), 50, prec=prec, sign_mode=args.sign_mode, threads=args.threads,
                                                      want_soft=False)
                    st['oracle_seconds'] += time.time() - t0
                    r = eng.decode(Y, 50, prec=prec, sign_mode=args.sign_mode)
                    same = (r['iters'] == iters) & (r['converged'] == conv) & (r['hard'] == hard).all(axis=1) & \
                           (r['ones'] == hard.sum(1, dtype=np.int64))
                    st['frames'] += F
                    st['identical_frames'] += int(same.sum())
                    st['iterations'] += int(iters.sum())
                    st['frame_errors'] += int((hard.sum(1) != 0).sum())
                    st['bit_errors'] += int(hard.sum())
                    st['not_converged'] += int((conv == 0).sum())
                    print(f'{name} float{prec}: frames {st["frames"]:8d} identical {st["identical_frames"]:8d} '
                          f'oracle {st["oracle_seconds"]:6.1f} s', flush=True)
                    if not same.all():
                        print('MISMATCH at frames', (first + np.flatnonzero(~same)[:10]).tolist(), flush=True)
                res['precisions'][f'float{prec}'] = st
            del llr
        out['codes'][name] = res
    out['result'] = ('identical frame for frame' if all(s['frames'] == s['identical_frames'] for c in out['codes'].values()
                                                        for s in c['precisions'].values()) else 'MISMATCH')
    print(json.dumps(out), flush=True)
    if args.out:
        json.dump(out, open(args.out, 'w'), indent=1)
    return 0 if out['result'].startswith('identical') else 1


if __name__ == '__main__':
    sys.exit(main())
