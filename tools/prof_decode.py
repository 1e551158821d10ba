#!/usr/bin/env python3
"""One decode launch (after one channel-fill launch) for rocprofv3 counter passes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=100000)
    ap.add_argument('--prec', type=int, default=32)
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--budget', type=int, default=50)
    ap.add_argument('--reps', type=int, default=1)
    ap.add_argument('--code', default='nearEarth',
                    help="nearEarth | nearEarth_w3 | nearEarth_w7 | nearEarth_episode40 | wifi_<rate> (1/2, 2/3, 3/4, 5/6)")
    ap.add_argument('--algo', choices=('minsum', 'spa'), default='minsum')
    ap.add_argument('--sign-mode', type=int, default=0)
    args = ap.parse_args()
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    if args.code == 'nearEarth':
        code = codes.near_earth()
    elif args.code == 'nearEarth_w3':
        code = codes.near_earth().with_block(0, 5, [17, 140, 400])
    elif args.code == 'nearEarth_w7':
        code = codes.near_earth().with_block(1, 9, [3, 50, 121, 200, 317, 404, 499])
    elif args.code == 'nearEarth_episode40':
        code = codes.gym_episode_state(codes.near_earth(), 40, 20211227)       # the state bench.py's secondary line times
    else:
        code = codes.wifi_1944(args.code.split('_', 1)[1])
    eng = ldpc_amd.Decoder(code)
    algo = ldpc_amd.ALGO_SUMPRODUCT if args.algo == 'spa' else ldpc_amd.ALGO_MINSUM
    scale = 4.0 * 10 ** (args.snr / 10) if args.algo == 'spa' else 1.0
    dev = torch.device('cuda:0')
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    F, n = args.frames, eng.n
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    hard = torch.empty((F, eng.nbytes), dtype=torch.uint8, device=dev)
    conv = torch.empty(F, dtype=torch.uint8, device=dev)
    ones = torch.empty(F, dtype=torch.int32, device=dev)
    eng.channel_awgn_device(llr, F, args.snr, seed=1)
    for _ in range(args.reps):
        eng.decode_device(llr, F, args.budget, prec=args.prec, algo=algo, sign_mode=args.sign_mode, llr_scale=scale,
                          d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones)
    torch.cuda.synchronize()
    print('mean iters', float(iters.float().mean()), 'frames', F, 'launch', eng.launch_info(prec=args.prec, algo=algo))


if __name__ == '__main__':
    main()
