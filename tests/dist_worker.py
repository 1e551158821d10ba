"""Worker of the multi-process tests (started by torch.distributed.run, one process per rank).
mode fake: the local evaluation is a deterministic function of (SNR index, global frame
number) -- no decoding -- so the partition, the counter reduction and the per-frame gather
can be checked on CPU ranks.  mode gpu: the real engine on cuda:0 (ranks share the card, counters
reduced over gloo).  mode nccl: the real engine, one rank per card, counters and per-frame data over RCCL."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_eval(snr, count, seed, first_frame, per_frame):
    cnt = np.zeros((len(snr), 6), np.uint64)
    pf = np.zeros((len(snr), count, 4), np.float32)
    for s in range(len(snr)):
        for k in range(count):
            g = first_frame + k
            h = (g * 2654435761 + s * 40503 + seed) & 0xffffffff
            dec, it, unc = h % 7, h % 51, 100 + h % 13
            cnt[s] += np.array([1, dec != 0, dec, it, unc, dec == 3], np.uint64)
            pf[s, k] = (0.5 + s, unc, dec, it)
    return cnt, (pf if per_frame else None)


def fake_variant(variant, snr, frames, seed):
    out = np.zeros((len(snr), 6), np.uint64)
    for s in range(len(snr)):
        out[s] = (frames, variant % 5, variant * 3 + s, frames * 2, 7, 0)
    return out


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    from ldpc_amd import distributed as D
    D.init('nccl' if mode == 'nccl' else 'gloo')
    rank, _, ws = D.world()
    res = {}
    if mode == 'fake':
        cnt, pf = D.sweep(None, [1.0, 2.0, 3.0], 37, seed=5, per_frame=True, evaluate=fake_eval)
        res = dict(cnt=cnt.tolist(), pf=pf.tolist())
        var = D.evaluate_variants(list(range(5)), [1.0, 2.0], 11, seed=1, evaluate=fake_variant)
        res['var'] = var.tolist()
    else:
        from ldpc_amd import codes
        cnt, pf = D.sweep(codes.near_earth(), [3.0, 3.4], 301, seed=99, per_frame=True)
        var = D.evaluate_variants([codes.near_earth_zeroed(0, 3), codes.near_earth_zeroed(1, 7), codes.near_earth()],
                                  [3.2], 64, seed=4)
        res = dict(cnt=cnt.tolist(), pf=pf.tolist(), var=var.tolist())
    if rank == 0:
        json.dump(res, open(out_path, 'w'))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
