#!/usr/bin/env python3
"""Decode throughput on the states a gym episode reaches (NearEarth after k actions, every action's
circulant kept, ldpc_env.py:293-317), float32 min-sum, frames resident in HBM: the kernel the library
picks, and -- with --old -- the table-driven kernel on the same frames (LDPC_HIP_TEST_HOOKS=1 and
LDPC_HIP_NO_HEAVYQC=1 at handle creation).  One JSON list on stdout.

    python tools/bench_episode.py [--actions 6 12 24 40 64] [--frames 20000] [--snr 3.5] [--old]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--actions', type=int, nargs='+', default=[6, 12, 24, 40, 64])
    ap.add_argument('--frames', type=int, default=20000)
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--sign-mode', type=int, default=1)
    ap.add_argument('--old', action='store_true', help='also time the table-driven kernel')
    args = ap.parse_args()
    if args.old:
        os.environ['LDPC_HIP_TEST_HOOKS'] = '1'
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()
    base = codes.near_earth()
    F, out = args.frames, []
    llr = torch.empty((F, base.n), dtype=torch.float32, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    ones = torch.empty(F, dtype=torch.int32, device=dev)
    for k in args.actions:
        code = codes.gym_episode_state(base, k, seed=20211227)
        for which in (('library choice', 'table-driven kernel') if args.old else ('library choice',)):
            if which != 'library choice':
                os.environ['LDPC_HIP_NO_HEAVYQC'] = '1'
            try:
                eng = ldpc_amd.Decoder(code)
            finally:
                os.environ.pop('LDPC_HIP_NO_HEAVYQC', None)
            with eng:
                eng.set_stream(stream.cuda_stream)
                eng.channel_awgn_device(llr, F, args.snr, seed=11)
                ts = []
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    eng.decode_device(llr, F, 50, prec=32, sign_mode=args.sign_mode, d_iters=iters, d_ones=ones)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                t = min(ts[1:])
                mi = float(iters.float().mean())
                edges = sum(code.row_degrees()) * code.Z
                out.append(dict(actions=k, kernel=which, row_degrees=code.row_degrees(), edges=edges, frames=F, kernel_ms=t,
                                info_gbps=7154 * F / (t * 1e-3) / 1e9, mean_iters=mi,
                                us_per_frame_iteration_per_cu=t * 1e3 * 256 / F / mi,
                                ns_per_1000_edges_per_iteration_per_cu=t * 1e6 * 256 / F / mi / (edges / 1000.0),
                                fer=float((ones != 0).float().mean()), launch=eng.launch_info(prec=32)))
                o = out[-1]
                print(f"{k:3d} actions {str(code.row_degrees()):10s} {which:20s} {o['info_gbps']:7.2f} Gbit/s  {mi:5.2f} iters  "
                      f"{o['us_per_frame_iteration_per_cu']:.2f} us/iteration/CU  lds {o['launch']['lds_bytes']}", file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
