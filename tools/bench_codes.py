#!/usr/bin/env python3
"""Secondary throughput figures (not the headline metric): one decode launch per configuration
on frames resident in HBM, timed with events on the launch stream.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ['LDPC_HIP_TEST_HOOKS'] = '1'         # one case forces the older kernel family for comparison
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    dev = torch.device('cuda:0')
    stream = torch.cuda.current_stream()
    cases = [
        # name, code, info bits, frames, snr dB, prec, algo, sign mode
        ('nearEarth min-sum f32', codes.near_earth(), 7154, 400000, 3.5, 32, 0, 0),
        ('nearEarth min-sum f32 on the compressed-row-state kernel (message planes switched off)', codes.near_earth(), 7154, 200000, 3.5, 32, 0, 0),
        ('nearEarth min-sum f64', codes.near_earth(), 7154, 200000, 3.5, 64, 0, 0),
        ('nearEarth sum-product f32', codes.near_earth(), 7154, 200000, 3.5, 32, 1, 0),
        ('nearEarth block (1,3) zeroed min-sum f32', codes.near_earth_zeroed(1, 3), 7154, 200000, 3.5, 32, 0, 0),
        ('nearEarth block (0,5) weight 3 min-sum f32, corrected sign rule', codes.near_earth().with_block(0, 5, [7, 100, 333]), 7154, 100000, 3.5, 32, 0, 1),
        ('802.11n 1944 r5/6 min-sum f32, corrected sign rule', codes.wifi_1944_r56(), 1620, 2000000, 4.0, 32, 0, 1),
        ('802.11n 1944 r5/6 min-sum f64, corrected sign rule', codes.wifi_1944_r56(), 1620, 1000000, 4.0, 64, 0, 1),
        ('802.11n 1944 r5/6 sum-product f32, corrected sign rule', codes.wifi_1944_r56(), 1620, 400000, 4.0, 32, 1, 1),
    ]
    out = []
    for name, code, k, F, snr, prec, algo, sm in cases:
        if 'switched off' in name:
            os.environ['LDPC_HIP_NO_PLANES'] = '1'
        else:
            os.environ.pop('LDPC_HIP_NO_PLANES', None)
        with ldpc_amd.Decoder(code) as eng:
            eng.set_stream(stream.cuda_stream)
            n = eng.n
            llr = torch.empty((F, n), dtype=torch.float32, device=dev)
            iters = torch.empty(F, dtype=torch.int32, device=dev)
            conv = torch.empty(F, dtype=torch.uint8, device=dev)
            ones = torch.empty(F, dtype=torch.int32, device=dev)
            hard = torch.empty((F, eng.nbytes), dtype=torch.uint8, device=dev)
            eng.channel_awgn_device(llr, F, snr, seed=11)
            scale = 4.0 * 10 ** (snr / 10) if algo == 1 else 1.0
            ts = []
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                eng.decode_device(llr, F, 50, prec=prec, algo=algo, sign_mode=sm, llr_scale=scale, d_hard=hard,
                                  d_iters=iters, d_conv=conv, d_ones=ones)
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t = min(ts[1:])
            out.append(dict(case=name, frames=F, snr_db=snr, kernel_ms=t, info_gbps=k * F / (t * 1e-3) / 1e9,
                            frames_per_s=F / (t * 1e-3), mean_iters=float(iters.float().mean()),
                            fer=float((ones != 0).float().mean()), launch=eng.launch_info(algo=algo, prec=prec)))
            print(f"{name:58s} {out[-1]['info_gbps']:8.2f} Gbit/s info  {out[-1]['mean_iters']:6.2f} iters  FER {out[-1]['fer']:.2e}",
                  file=sys.stderr, flush=True)
            del llr, hard
    print(json.dumps(out))


if __name__ == '__main__':
    main()
