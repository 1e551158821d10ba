#!/usr/bin/env python3
"""Randomised parity campaign (not collected by pytest: run it by name on the GPU box).

    python tests/fuzz_campaign.py --seconds 600 --seed 1 --out gpurun_out/fuzz.json

Draws parity-check matrices of many shapes -- block-circulant with circulant sizes from 5 to 511 and block
weights 0 ... 9, NearEarth after random gym episodes, unstructured sparse matrices with light and very heavy
rows -- so that every kernel family of the library is reached (message planes with and without extra edges,
short-code kernels with fixed and with packed records, the table-driven kernel with sign words of 32, 64 and 128
bits, the edge-list kernels with compressed rows and with one message per edge), and for each draws a batch of
channel values (some frames clean, some with erased stretches, some hopeless), an iteration budget, a sign
rule, a precision and a mode (fresh decoder per frame, re-used decoder, min-sum scaling, sum-product; the
device and allocation-free entry points against the host entry point; the sweep driver's launch forms against
each other).  The
library's results must equal the CPU oracle's bit for bit (sum-product, whose transcendentals are not pinned:
the frames the oracle converges are counted, identical or not, and reported).
Prints one progress line every few cases and a JSON summary at the end; exits non-zero at the first mismatch,
after writing the failing case's parameters.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def draw_qc(prng, codes):
    kind = prng.randint(5)
    if kind == 0:        # NearEarth after a gym episode
        n_actions = int(prng.choice([1, 2, 4, 8, 16, 40, 80]))
        return 'nearEarth + %d gym actions' % n_actions, codes.gym_episode_state(codes.near_earth(), n_actions, int(prng.randint(1 << 30)))
    if kind == 1:        # an 802.11n rate with a few blocks changed
        rate = ['1/2', '2/3', '3/4', '5/6'][prng.randint(4)]
        qc = codes.wifi_1944(rate)
        for _ in range(prng.randint(3)):
            b, j = int(prng.randint(qc.mb)), int(prng.randint(qc.nb))
            w = int(prng.randint(0, 3))
            qc = qc.with_block(b, j, sorted(prng.choice(qc.Z, w, replace=False).tolist()))
        return '802.11n r%s mutated' % rate, qc
    # a random block-circulant code
    Z = int(prng.choice([5, 7, 16, 27, 31, 54, 64, 81, 96, 127, 128, 200, 255, 256, 511]))
    mb = int(prng.randint(1, max(2, min(12, 1024 // ((Z + 63) // 64 * 64)) + 1)))
    nb = int(prng.randint(mb + 1, min(32 * mb, mb + 24) + 1))
    heavy = prng.rand() < 0.3
    blocks = []
    for b in range(mb):
        row = []
        for j in range(nb):
            p = prng.rand()
            w = 0 if p < 0.35 else 1 if p < 0.8 else 2 if p < 0.93 else int(prng.randint(3, 10 if heavy else 4))
            w = min(w, Z)
            row.append(tuple(sorted(prng.choice(Z, w, replace=False).tolist())))
        blocks.append(row)
    for b in range(mb):          # every check at least two edges, every column at least one
        while sum(len(x) for x in blocks[b]) < 2:
            j = int(prng.randint(nb))
            blocks[b][j] = tuple(sorted(set(blocks[b][j]) | {int(prng.randint(Z))}))
    for j in range(nb):
        if all(len(blocks[b][j]) == 0 for b in range(mb)):
            blocks[int(prng.randint(mb))][j] = (int(prng.randint(Z)),)
    return 'random QC Z=%d %dx%d%s' % (Z, mb, nb, ' heavy' if heavy else ''), codes.QCCode(Z, blocks)


def draw_dense(prng):
    m = int(prng.choice([6, 24, 60, 97, 250, 500, 1000]))
    n = int(m * prng.choice([1.5, 2, 3, 4]) + prng.randint(0, 9))
    dv = int(prng.choice([2, 3, 4, 6]))
    H = np.zeros((m, n), np.int32)
    for c in range(n):
        H[prng.choice(m, min(dv, m), replace=False), c] = 1
    for r in range(m):
        while H[r].sum() < 2:
            H[r, prng.randint(n)] = 1
    if prng.rand() < 0.3:        # a few very heavy checks
        for r in prng.choice(m, 2, replace=False):
            H[r, prng.choice(n, min(n, int(prng.randint(40, 200))), replace=False)] = 1
    return 'unstructured %dx%d dv=%d max row %d' % (m, n, dv, int(H.sum(1).max())), H


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=300)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    os.environ.setdefault('LDPC_HIP_TEST_HOOKS', '1')
    import oracle as om
    import ldpc_amd
    from ldpc_amd import codes
    from conftest import make_llrs
    prng = np.random.RandomState(args.seed)
    t_end = time.time() + args.seconds
    stats = {'cases': 0, 'frames': 0, 'kernels': {}, 'modes': {}, 'seconds': args.seconds, 'seed': args.seed, 'skipped': 0}
    t_print = time.time()
    while time.time() < t_end:
        if prng.rand() < 0.7:
            label, code = draw_qc(prng, codes)
            H = code.to_dense()
        else:
            label, H = draw_dense(prng)
            code = H
        m, n = H.shape
        if H.sum(1).max() > 400 or n > 16000:
            stats['skipped'] += 1
            continue
        F = int(prng.choice([1, 2, 3, 5, 8, 17, 40, 100]))
        if m * n > 4e6:
            F = min(F, 17)
        snr = float(prng.choice([0.0, 2.0, 3.5, 5.0, 8.0]))
        Y = make_llrs(int(prng.randint(1 << 30)), snr, n, F)
        if F > 1:
            Y[prng.randint(F)] = np.float32(-1.0)                         # a codeword at once
        if F > 2:
            k = prng.randint(F)
            a0 = prng.randint(n)
            Y[k, a0:a0 + prng.randint(1, max(2, n // 3))] = 0.0           # an erased stretch (exact zeros)
        budget = int(prng.choice([0, 1, 2, 5, 12, 50]))
        prec = int(prng.choice([32, 64]))
        sm = int(prng.randint(2))
        mode = ['fresh', 'fresh', 'chained', 'scaled', 'spa', 'device', 'sweep'][prng.randint(7)]
        dt = np.float64 if prec == 64 else np.float32
        orc = om.Oracle(H)
        params = dict(code=label, m=m, n=n, frames=F, snr=snr, budget=budget, prec=prec, sign_mode=sm, mode=mode)
        try:
            with ldpc_amd.Decoder(code) as eng:
                info = eng.launch_info(prec=prec, algo=1 if mode == 'spa' else 0)
                key = ('state in global memory (edge-list kernels)' if not info['lds_bytes'] else
                       'state in LDS, one frame per workgroup' if info['frames_per_wg'] == 1 else
                       'state in LDS, several frames per workgroup')
                if mode == 'spa':
                    scale = float(np.float32(4 * 10 ** (max(snr, 1.0) / 10)))
                    hs, is_, cs, _ = orc.spa(Y.astype(dt), scale, budget, prec=prec, sign_mode=sm, want_soft=False, threads=8)
                    r = eng.decode((Y * np.float32(scale)).astype(dt), budget, prec=prec, algo=ldpc_amd.ALGO_SUMPRODUCT, sign_mode=sm)
                    # sum-product is not bit-pinned (hardware exp2 / log2 / rcp against the C library's): frames the
                    # oracle converges are counted, identical or not; a case fails only if agreement is poor
                    cv = cs == 1
                    same = (r['iters'][cv] == is_[cv]) & (r['hard'][cv] == hs[cv]).all(axis=1)
                    stats['spa_converged_frames'] = stats.get('spa_converged_frames', 0) + int(cv.sum())
                    stats['spa_identical_frames'] = stats.get('spa_identical_frames', 0) + int(same.sum())
                    if cv.sum() and same.mean() < stats.get('spa_worst_case', (1.0, ''))[0]:
                        stats['spa_worst_case'] = (float(same.mean()), json.dumps(params))
                    ok = True
                elif mode == 'device':
                    # device entry points (plain and allocation-free) against the host entry point, which the
                    # other modes pin to the oracle
                    import torch
                    algo = int(prng.randint(2)) if prec == 32 else 0
                    want = eng.decode(Y, budget, prec=prec, sign_mode=sm, want_soft=True, algo=algo)
                    d = torch.from_numpy(Y).cuda()
                    ok = True
                    for noalloc in (False, True):
                        hard = torch.zeros((F, eng.nbytes), dtype=torch.uint8, device='cuda')
                        it_ = torch.zeros(F, dtype=torch.int32, device='cuda')
                        cv_ = torch.zeros(F, dtype=torch.uint8, device='cuda')
                        on_ = torch.zeros(F, dtype=torch.int32, device='cuda')
                        soft = torch.zeros((F, n), dtype=torch.float64 if prec == 64 else torch.float32, device='cuda')
                        try:
                            eng.decode_device(d, F, budget, prec=prec, algo=algo, sign_mode=sm, d_hard=hard, d_iters=it_,
                                              d_conv=cv_, d_ones=on_, d_soft=soft, noalloc=noalloc)
                        except ldpc_amd.LdpcHipError as e:
                            if noalloc and ('workspace' in str(e) or 'edge tables' in str(e)):
                                stats['noalloc_refused'] = stats.get('noalloc_refused', 0) + 1
                                continue
                            raise
                        torch.cuda.synchronize()
                        bits = np.unpackbits(hard.cpu().numpy(), axis=1, bitorder='little')[:, :n]
                        ok = ok and (np.array_equal(bits, want['hard']) and np.array_equal(it_.cpu().numpy(), want['iters']) and
                                     np.array_equal(cv_.cpu().numpy(), want['converged']) and np.array_equal(on_.cpu().numpy(), want['ones']) and
                                     np.array_equal(soft.cpu().numpy(), want['soft']))
                elif mode == 'sweep':
                    # the sweep driver: one launch for all points against one launch per point, noise made inside the
                    # decode kernel against the channel kernel into a buffer
                    snrs = [float(x) for x in prng.choice([1.0, 2.5, 3.5, 5.0, 7.0], int(prng.randint(1, 5)), replace=False)]
                    algo = int(prng.randint(2)) if prec == 32 else 0
                    kw = dict(seed=int(prng.randint(1, 1 << 30)), first_frame=int(prng.randint(1000)), max_iters=budget, prec=prec,
                              algo=algo, sign_mode=sm, per_frame=True)
                    outs = []
                    for env in ({}, {'LDPC_HIP_NO_POINT_BATCH': '1'}, {'LDPC_HIP_NO_FUSED_CHANNEL': '1', 'LDPC_HIP_NO_POINT_BATCH': '1'}):
                        for k_ in ('LDPC_HIP_NO_POINT_BATCH', 'LDPC_HIP_NO_FUSED_CHANNEL'):
                            os.environ.pop(k_, None)
                        os.environ.update(env)
                        outs.append(eng.eval_awgn(snrs, F, **kw))
                    for k_ in ('LDPC_HIP_NO_POINT_BATCH', 'LDPC_HIP_NO_FUSED_CHANNEL'):
                        os.environ.pop(k_, None)
                    ok = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs[1:])
                    ok = ok and bool((outs[0][0][:, 0] == F).all())
                elif mode == 'chained':
                    # A re-used decoder that never converges feeds its messages back frame after frame; after some
                    # tens of frames they overflow (the reference's would too, ldpc.py:270-274, 356).  What min / med3
                    # make of inf - inf is not part of the contract: frames are compared up to the overflow.
                    st = orc.new_state(prec)
                    res, K = [], F
                    for k in range(F):
                        res.append(orc.minsum(Y[k:k + 1].astype(dt), budget, prec=prec, sign_mode=sm, state=st))
                        if not np.isfinite(st).all():
                            K = k
                            break
                    r = eng.decode(Y, budget, prec=prec, sign_mode=sm, want_soft=True, chained=True)
                    if K < F:
                        stats['chained_overflows'] = stats.get('chained_overflows', 0) + 1
                    hard = np.concatenate([x[0] for x in res[:K]]) if K else np.zeros((0, n), np.uint8)
                    iters = np.concatenate([x[1] for x in res[:K]]) if K else np.zeros(0, np.int32)
                    conv = np.concatenate([x[2] for x in res[:K]]) if K else np.zeros(0, np.uint8)
                    soft = np.concatenate([x[3] for x in res[:K]]) if K else np.zeros((0, n), dt)
                    ok = (np.array_equal(r['iters'][:K], iters) and np.array_equal(r['converged'][:K], conv) and
                          np.array_equal(r['hard'][:K], hard) and np.array_equal(r['soft'][:K], soft))
                else:
                    alpha, beta = (0.8125, 0.0) if mode == 'scaled' and prng.rand() < 0.5 else (1.0, 0.25) if mode == 'scaled' else (1.0, 0.0)
                    eng.set_minsum_scaling(alpha, beta)
                    hard, iters, conv, soft = orc.minsum(Y.astype(dt), budget, prec=prec, sign_mode=sm, alpha=alpha, beta=beta, threads=8)
                    r = eng.decode(Y, budget, prec=prec, sign_mode=sm, want_soft=True)
                    # A frame that never converges on a code full of 4-cycles (a circulant of weight 9) multiplies its
                    # messages by up to 8 per iteration and leaves float32 after some 44 of them (float64, the
                    # reference's CPU arithmetic, does not get there in 50).  Up to and including the first infinities
                    # the kernels and the oracle agree; what min / med3 make of the NaNs that follow is not part of the
                    # contract, so a frame the oracle ends with a non-finite posterior is compared in its iteration
                    # count and status only.
                    fin = np.isfinite(soft).all(axis=1)
                    if not fin.all():
                        stats['overflowed_frames'] = stats.get('overflowed_frames', 0) + int((~fin).sum())
                    ok = (np.array_equal(r['iters'], iters) and np.array_equal(r['converged'], conv) and
                          np.array_equal(r['hard'][fin], hard[fin]) and np.array_equal(r['soft'][fin], soft[fin]) and
                          np.array_equal(r['ones'][fin], hard[fin].sum(1)))
        except ldpc_amd.LdpcHipError as e:
            params['error'] = str(e)
            ok = False
        if not ok:
            print('MISMATCH', json.dumps(params), flush=True)
            if args.out:
                extra = {}
                if mode not in ('spa', 'device', 'sweep') and 'error' not in params:
                    extra = dict(want_iters=iters, want_conv=conv, want_hard=hard, got_iters=r['iters'], got_conv=r['converged'],
                                 got_hard=r['hard'])
                np.savez_compressed(args.out + '.case.npz', H=H.astype(np.uint8), Y=Y, **extra)
            if args.out:
                json.dump({'failed': params, 'stats': stats}, open(args.out, 'w'), indent=1)
            sys.exit(1)
        stats['cases'] += 1
        stats['frames'] += F
        stats['kernels'][key] = stats['kernels'].get(key, 0) + 1
        stats['modes']['%s f%d' % (mode, prec)] = stats['modes'].get('%s f%d' % (mode, prec), 0) + 1
        if time.time() - t_print > 20:
            t_print = time.time()
            print('cases %d frames %d last: %s' % (stats['cases'], stats['frames'], label), flush=True)
    stats['result'] = 'every min-sum case bit-identical to the oracle'
    print(json.dumps(stats), flush=True)
    if args.out:
        json.dump(stats, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
