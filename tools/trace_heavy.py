#!/usr/bin/env python3
"""Phase times of the heavy-circulant kernel (qc_heavy.hip built with -DHQ_TRACE: the soft-output buffer
receives, per workgroup and wave, the clock cycles spent in check passes, the barrier after them, variable
passes and the barrier after those).

    touch ldpc_amd/csrc/qc_heavy.hip && make -C ldpc_amd/csrc EXTRA=-DHQ_TRACE
    python tools/trace_heavy.py --actions 40
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--actions', type=int, default=40)
    ap.add_argument('--frames', type=int, default=5120)
    ap.add_argument('--snr', type=float, default=3.5)
    args = ap.parse_args()
    import numpy as np
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    code = codes.gym_episode_state(codes.near_earth(), args.actions, seed=20211227)
    eng = ldpc_amd.Decoder(code)
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    F, n = args.frames, eng.n
    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    soft = torch.zeros((256 * 16 * 4 * 2,), dtype=torch.float32, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    eng.channel_awgn_device(llr, F, args.snr, seed=1)
    for rep in range(2):
        soft.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        eng.decode_device(llr, F, 50, prec=32, sign_mode=1, d_iters=iters, d_soft=soft)
        e1.record(stream)
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    it_total = float(iters.sum().item())
    print(f'row degrees {code.row_degrees()}, kernel {ms:.3f} ms, {ms * 1e3 * 256 / it_total:.2f} us per frame-iteration per CU')
    raw = soft.cpu().numpy().view(np.uint64).reshape(256, 16, 4).astype(np.float64)
    per_it = raw.sum(axis=0) / it_total
    print('mean clock cycles per iteration: check pass, barrier, variable pass, barrier, sum (__builtin_readcyclecounter: core clocks)')
    for w in range(16):
        print(f'  wave {w:2d} ' + ' '.join(f'{v:9.0f}' for v in per_it[w]) + f' {per_it[w].sum():9.0f}')


if __name__ == '__main__':
    main()
