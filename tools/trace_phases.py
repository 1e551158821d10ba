#!/usr/bin/env python3
"""Phase timestamps of the fp32 NearEarth kernel (library built with -DLDPC_TRACE: the
soft-output buffer receives ten clock64() samples per frame from thread 0 of the workgroup)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=25600)
    ap.add_argument('--budgets', default='0,4')
    ap.add_argument('--snr', type=float, default=1.0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    eng = ldpc_amd.Decoder(codes.near_earth())
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    F, n = args.frames, eng.n
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    soft = torch.zeros((F, 512), dtype=torch.float32, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    hard = torch.empty((F, eng.nbytes), dtype=torch.uint8, device=dev)
    conv = torch.empty(F, dtype=torch.uint8, device=dev)
    ones = torch.empty(F, dtype=torch.int32, device=dev)
    eng.channel_awgn_device(llr, F, args.snr, seed=1)
    names = ['lds-write+prefetch-issue', 'barrier', 'check0', 'barrier', 'iterations', 'ballot loop',
             'consume prefetch', 'barrier', 'stores', 'loop back']
    for b in [int(x) for x in args.budgets.split(',')]:
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            eng.decode_device(llr, F, b, prec=32, d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones, d_soft=soft)
            e1.record(stream)
            torch.cuda.synchronize()
        print(f'kernel {e0.elapsed_time(e1):.3f} ms = {e0.elapsed_time(e1) * 1e3 / F * 256:.2f} us per frame per CU')
        t = soft.cpu().numpy().view(np.uint64).reshape(F, 16, 16)[:, :, :10].astype(np.int64)
        wg = 100
        rows = t[wg::256][5:-1]                      # [frames, wave, stamp]
        t0 = rows[:, :1, :1].min(axis=1, keepdims=True)
        t0 = rows[:, :, 0].min(axis=1)[:, None, None]
        rel = (rows - t0).mean(axis=0)               # [wave, stamp], relative to the earliest wave's frame start
        print(f'budget {b}: mean cycle stamps relative to the first wave entering the frame (workgroup {wg})')
        print('  wave ' + ' '.join(f'{k:>7d}' for k in range(10)))
        for w in range(16):
            print(f'  {w:4d} ' + ' '.join(f'{v:7.0f}' for v in rel[w]))
        nxt = t[wg::256][6:]
        print('  frame period', (nxt[:, 0, 0] - rows[:, 0, 0]).mean())


if __name__ == '__main__':
    main()
