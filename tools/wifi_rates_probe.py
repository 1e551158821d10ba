#!/usr/bin/env python3
"""Kernel time of the four 802.11n n = 1944 rates at the SNR each is quoted at (10^6 frames, early stop) and
with every frame forced through all 50 iterations (10^5 hopeless frames): the second figure is the cost of an
iteration, the difference to the first what a frame costs beyond its iterations.  LDPC_HIP_TEST_HOOKS=1 with
LDPC_HIP_NO_PACKED=1 / LDPC_HIP_PACKED_RPT=2|4 selects the other short-code kernel / a group width."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ldpc_amd
from ldpc_amd import codes
for rate, snr in (('1/2', 2.0), ('2/3', 3.0), ('3/4', 3.5), ('5/6', 4.0)):
    code = codes.wifi_1944(rate)
    with ldpc_amd.Decoder(code) as eng:
        F = 1000000
        llr = torch.empty((F, eng.n), dtype=torch.float32, device='cuda')
        iters = torch.empty(F, dtype=torch.int32, device='cuda')
        conv = torch.empty(F, dtype=torch.uint8, device='cuda')
        s = torch.cuda.current_stream(); eng.set_stream(s.cuda_stream)
        eng.channel_awgn_device(llr, F, snr, seed=3)
        for fixed in (False, True):
            mi = 50
            if fixed:
                llr2 = llr[:100000] * 0 - 0.01 + 0.5 * torch.randn((100000, eng.n), device='cuda')   # hopeless frames: all 50 iterations
                buf, n = llr2, 100000
            else:
                buf, n = llr, F
            eng.decode_device(buf, n, mi, prec=32, sign_mode=1, d_iters=iters, d_conv=conv)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); eng.decode_device(buf, n, mi, prec=32, sign_mode=1, d_iters=iters, d_conv=conv); b.record(s)
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            it = iters[:n].float().mean().item()
            E = code.nnz
            print(f'rate {rate} fixed={fixed} frames {n} ms {ms:.2f} mean_iters {it:.2f} fer {(conv[:n]==0).float().mean().item():.4f} '
                  f'ns per frame-iteration per CU {ms*1e6/n/max(it,1e-9)*256:.0f}  edge-iters/s {E*n*it/ms/1e9:.3f}e12 launch {eng.launch_info(prec=32)}')
