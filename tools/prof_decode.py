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
    args = ap.parse_args()
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    eng = ldpc_amd.Decoder(codes.near_earth())
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
        eng.decode_device(llr, F, args.budget, prec=args.prec, d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones)
    torch.cuda.synchronize()
    print('mean iters', float(iters.float().mean()), 'frames', F)


if __name__ == '__main__':
    main()
