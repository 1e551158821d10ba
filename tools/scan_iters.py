#!/usr/bin/env python3
"""Per-iteration and per-frame cost of the decode kernel: frames at an SNR where nothing
converges, decoded with iteration budgets 0..N; time is linear in the budget."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=100000)
    ap.add_argument('--prec', type=int, default=32)
    ap.add_argument('--snr', type=float, default=1.0)
    ap.add_argument('--code', default='nearEarth')
    ap.add_argument('--budgets', default='0,1,2,4,8,16')
    ap.add_argument('--algo', default='minsum', choices=['minsum', 'spa'])
    ap.add_argument('--no-hard', action='store_true')
    ap.add_argument('--no-ones', action='store_true')
    args = ap.parse_args()
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    qc = {'nearEarth': codes.near_earth, 'wifi': codes.wifi_1944_r56,
          'zero13': lambda: codes.near_earth_zeroed(1, 3),
          'w3': lambda: codes.near_earth().with_block(0, 5, [7, 100, 333]),
          'w1': lambda: codes.near_earth().with_block(1, 9, [42])}[args.code]()
    eng = ldpc_amd.Decoder(qc)
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    F, n = args.frames, eng.n
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    hard = torch.empty((F, eng.nbytes), dtype=torch.uint8, device=dev)
    conv = torch.empty(F, dtype=torch.uint8, device=dev)
    ones = torch.empty(F, dtype=torch.int32, device=dev)
    eng.channel_awgn_device(llr, F, args.snr, seed=1)
    algo = ldpc_amd.ALGO_SUMPRODUCT if args.algo == 'spa' else ldpc_amd.ALGO_MINSUM
    scale = 4.0 * 10 ** (args.snr / 10) if args.algo == 'spa' else 1.0      # sum-product takes 2y/sigma^2
    print('launch', eng.launch_info(algo=algo, prec=args.prec))
    res = []
    for b in [int(x) for x in args.budgets.split(',')]:
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            eng.decode_device(llr, F, b, prec=args.prec, algo=algo, llr_scale=scale, d_hard=None if args.no_hard else hard, d_iters=iters, d_conv=conv,
                              d_ones=None if args.no_ones else ones)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        mi = float(iters.float().mean())
        t = min(ts)
        res.append((b, mi, t))
        print(f'budget {b:3d} mean iters {mi:6.2f}  {t:9.3f} ms  -> {t * 1e3 / F * 256:8.2f} us per frame per CU')
    if len(res) >= 2:
        (b0, m0, t0), (b1, m1, t1) = res[0], res[-1]
        per_it = (t1 - t0) / (m1 - m0) * 1e3 / F * 256
        print(f'per iteration: {per_it:.3f} us per frame per CU (256 CUs); per-frame overhead {t0 * 1e3 / F * 256:.2f} us')


if __name__ == '__main__':
    main()
