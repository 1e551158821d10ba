#!/usr/bin/env python3
"""Decode throughput on NearEarth with gym-style heavy circulants (weight 3..7), float32 min-sum,
frames resident in HBM; one JSON list.  Same timing method as tools/bench_codes.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--algo', choices=('minsum', 'spa'), default='minsum')
    args = ap.parse_args()
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()
    base = codes.near_earth()
    cases = [
        ('NearEarth', base),
        ('block (0,5) weight 3', base.with_block(0, 5, [7, 100, 333])),
        ('block (1,9) weight 7 + block (0,12) weight 1', base.with_block(1, 9, [3, 77, 150, 222, 301, 404, 500]).with_block(0, 12, [9])),
        ('blocks (0,3) weight 4, (1,3) weight 5, (0,8) weight 3', base.with_block(0, 3, [1, 2, 3, 400]).with_block(1, 3, [5, 50, 150, 250, 350]).with_block(0, 8, [10, 20, 30])),
    ]
    F, out = 200000, []
    for name, code in cases:
        with ldpc_amd.Decoder(code) as eng:
            eng.set_stream(stream.cuda_stream)
            llr = torch.empty((F, eng.n), dtype=torch.float32, device=dev)
            iters = torch.empty(F, dtype=torch.int32, device=dev)
            ones = torch.empty(F, dtype=torch.int32, device=dev)
            eng.channel_awgn_device(llr, F, 3.5, seed=11)
            ts = []
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                eng.decode_device(llr, F, 50, prec=32, sign_mode=1, d_iters=iters, d_ones=ones,
                                  algo=ldpc_amd.ALGO_SUMPRODUCT if args.algo == 'spa' else ldpc_amd.ALGO_MINSUM,
                                  llr_scale=4.0 * 10 ** 0.35 if args.algo == 'spa' else 1.0)
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t = min(ts[1:])
            mi = float(iters.float().mean())
            out.append(dict(case=name, frames=F, kernel_ms=t, info_gbps=7154 * F / (t * 1e-3) / 1e9, mean_iters=mi,
                            us_per_frame_iteration_per_cu=t * 1e3 * 256 / F / mi, fer=float((ones != 0).float().mean()),
                            launch=eng.launch_info(prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT if args.algo == 'spa' else ldpc_amd.ALGO_MINSUM)))
            print(f"{name:60s} {out[-1]['info_gbps']:7.2f} Gbit/s  {mi:5.2f} iters  {out[-1]['us_per_frame_iteration_per_cu']:.2f} us/iteration/CU",
                  file=sys.stderr, flush=True)
            del llr
    print(json.dumps(out))


if __name__ == '__main__':
    main()
