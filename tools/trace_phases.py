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
    ap.add_argument('--heavy', action='store_true', help='NearEarth with block (0,5) of weight 3 (heavy-circulant kernel)')
    args = ap.parse_args()
    import numpy as np
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    code = codes.near_earth().with_block(0, 5, [7, 100, 333]) if args.heavy else codes.near_earth()
    eng = ldpc_amd.Decoder(code)
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    F, n = args.frames, eng.n
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    soft = torch.zeros((256 * 16 * 16 * 2,), dtype=torch.float32, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    hard = torch.empty((F, eng.nbytes), dtype=torch.uint8, device=dev)
    conv = torch.empty(F, dtype=torch.uint8, device=dev)
    ones = torch.empty(F, dtype=torch.int32, device=dev)
    eng.channel_awgn_device(llr, F, args.snr, seed=1)
    names = ['y->LDS', 'barrier', 'prefetch+flush issue', 'check pass 0', 'barrier', 'iterations', 'ballot',
             'consume prefetch', 'barrier', 'bookkeeping', 'loop back']
    for b in [int(x) for x in args.budgets.split(',')]:
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            eng.decode_device(llr, F, b, prec=32, d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones, d_soft=soft)
            e1.record(stream)
            torch.cuda.synchronize()
        print(f'budget {b}: kernel {e0.elapsed_time(e1):.3f} ms = {e0.elapsed_time(e1) * 1e3 / F * 256:.2f} us per frame per CU')
        raw = soft.cpu().numpy().view(np.uint64)
        t = raw[:256 * 16 * 12].reshape(256, 16, 12).astype(np.float64)
        inner = raw[256 * 16 * 12:].reshape(256, 16, 4).astype(np.float64)
        iters_total = float(iters.sum().item())
        per = t[:, :, :11] / t[:, :, 11:12]                 # mean cycles per frame, [workgroup, wave, phase]
        m = per.mean(axis=0)
        print('  mean cycles per frame and phase (rows: waves 0, 4, 8, 12, 15; last column: sum)')
        print('        ' + ' '.join(f'{n[:10]:>10s}' for n in names))
        for w in (0, 4, 8, 12, 15):
            print(f'  w{w:<4d} ' + ' '.join(f'{v:10.0f}' for v in m[w]) + f' {m[w].sum():10.0f}')
        # steady-state loop, cycles per frame-iteration: every workgroup's sum over its frames / iterations it ran
        per_it = inner.sum(axis=0) / iters_total
        print('  steady-state loop, mean cycles per iteration (check pass, barrier, variable pass, barrier)')
        for w in range(16):
            print(f'  w{w:<4d} ' + ' '.join(f'{v:10.0f}' for v in per_it[w]) + f' {per_it[w].sum():10.0f}')


if __name__ == '__main__':
    main()
