#!/usr/bin/env python3
"""Headline benchmark: decoded info Gbit/s, NearEarth (8176,7154) min-sum, <= 50 iterations,
early stop on zero syndrome, BPSK/AWGN 3.5 dB (sigma = sqrt(0.5/10^(dB/10)), the reference's
SNR definition, ldpc.py:53-55), 10^6 frames per GPU resident in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the decode path over the rank's batch of frames (one kernel launch
of the decode kernel) followed by the reduction of the error counters over ranks (RCCL
all-reduce of 6 uint64 when N > 1).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_INFO = 7154          # message size the metric is quoted on (ldpc.py:486)
N_CODE = 8176
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec)
# algorithmic HBM bytes per frame of the decode kernel: LLR in (8176 x 4) + packed hard
# decision out (1022) + iteration count (4) + status (1) + ones count (4)
BYTES_PER_FRAME = N_CODE * 4 + (N_CODE + 7) // 8 + 4 + 1 + 4


SPA_ENGINE = [None]      # the engine whose sum-product kernel the CPU leg checks (set by main)


def spa_agreement(om, orc, n_threads, frames=20000):
    """Sum-product has no counterpart in the reference, so its only oracle is this build's CPU statement: the
    fraction of frames on which the GPU kernel and that statement agree in (iterations, status, bits), float32,
    NearEarth at two SNR points, 2*10^4 device-made frames each.  Part of the CPU leg: the oracle is the checker."""
    import numpy as np
    import torch
    import ldpc_amd
    eng = SPA_ENGINE[0]
    out = []
    llr = torch.empty((frames, eng.n), dtype=torch.float32, device='cuda')
    for snr in (3.1, 3.3):
        eng.channel_awgn_device(llr, frames, snr, seed=777001)
        eng.sync()
        Y = llr.cpu().numpy()
        scale = 4 * 10 ** (snr / 10)
        hard, iters, conv, _ = orc.spa(Y, scale, 50, prec=32, want_soft=False, threads=n_threads)
        r = eng.decode(Y * np.float32(scale), 50, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT)
        ok = (r['iters'] == iters) & (r['converged'] == conv) & (r['hard'] == hard).all(axis=1)
        cv = conv == 1
        out.append({'code': 'NearEarth', 'snr_db': snr, 'frames': frames, 'identical_fraction': float(ok.mean()),
                    'identical_fraction_among_cpu_converged': float(ok[cv].mean()) if cv.any() else None,
                    'cpu_converged_fraction': float(cv.mean()), 'bar': 0.9999, 'meets_bar': bool(ok.mean() >= 0.9999)})
    return out


def cpu_baseline(llr_dev, n_frames, n_threads, slice_frames=16384):
    """The CPU oracle (C restatement of ldpc.py's decoder, float64) timed on the host cores:
    the first n_frames frames of the batch, fetched and decoded slice by slice (only the
    decode calls are timed)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import numpy as np
    import oracle as om
    from ldpc_amd import codes
    orc = om.Oracle(codes.near_earth().to_dense())
    tn, done, it_all = 0.0, 0, []
    t1 = one = None
    while done < n_frames:
        m = min(slice_frames, n_frames - done)
        y = llr_dev[done:done + m].cpu().numpy().astype(np.float64)
        if t1 is None:
            one = min(m, 4096)
            t0 = time.perf_counter()
            orc.minsum(y[:one], 50, prec=64, want_soft=False, threads=1)
            t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        hard, iters, conv, _ = orc.minsum(y, 50, prec=64, want_soft=False, threads=n_threads)
        tn += time.perf_counter() - t0
        it_all.append(iters)
        done += m
    iters = np.concatenate(it_all)
    spa = spa_agreement(om, orc, n_threads) if SPA_ENGINE[0] is not None else None
    return dict(spa_agreement=spa, value=K_INFO * n_frames / tn / 1e9, unit='Gbit/s', cores=n_threads, kind='port',
                sample=f'first {n_frames} frames of the same batch (3.5 dB, <=50 iterations, float64), '
                       f'{n_threads} OpenMP threads over frames, {tn:.2f} s of decode calls',
                single_core_value=K_INFO * one / t1 / 1e9, single_core_frames=one,
                mean_iters=float(iters.mean())), iters


def wilson(k, n, z=1.96):
    """95 % Wilson score interval of a proportion k / n."""
    if n == 0:
        return [0.0, 1.0]
    p = k / n
    d = 1 + z * z / n
    mid = (p + z * z / (2 * n)) / d
    half = z * ((p * (1 - p) + z * z / (4 * n)) / n) ** 0.5 / d
    return [max(mid - half, 0.0), min(mid + half, 1.0)]


def cpu_model():
    """What /proc/cpuinfo says about the first core: model name (virtual machines often shorten it), family /
    model / stepping, clock and the widest vector extension."""
    want = {'model name': None, 'cpu family': None, 'model': None, 'stepping': None, 'cpu MHz': None, 'flags': ''}
    try:
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k = k.strip()
            if k in want and not want[k]:
                want[k] = v.strip()
            if not line.strip():
                break
    except OSError:
        return None
    flags = want['flags'].split()
    simd = next((f for f in ('avx512f', 'avx2', 'avx', 'sse4_2') if f in flags), None)
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = None
    return {'model_name': want['model name'], 'family_model_stepping': [want['cpu family'], want['model'], want['stepping']],
            'mhz': want['cpu MHz'], 'widest_simd': simd, 'cores_this_process_may_use': usable}


def library_source_sha16():
    """Fingerprint of the sources libldpc_hip.so is built from (the GPU box has no git history to ask): the
    committed counter passes carry the fingerprint they were taken on, and a figure read from them is marked
    stale when the library has changed since."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'ldpc_amd', 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'ldpc_amd', 'csrc', '*.hip')) +
                   glob.glob(os.path.join(ROOT, 'ldpc_amd', 'csrc', '*.cc')) + [os.path.join(ROOT, 'ldpc_amd', 'csrc', 'Makefile'),
                                                                              os.path.join(ROOT, 'include', 'ldpc_hip.h')])
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def committed_counters(prec, frames, spa):
    """HBM bytes per launch and busy fractions from the committed counter passes (rocprofv3 --pmc cannot
    run inside this process), with the file and the commit they were taken on so that a stale constant
    shows; all None for a configuration that was not profiled."""
    prof = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if spa or not os.path.exists(prof):
        return None, None, None
    try:
        pj = json.load(open(prof))
        e = pj.get(f'prec{prec}_frames{frames}')
        if not e:
            return None, None, None
        onchip = {k: e[k] for k in ('valu_busy_frac', 'lds_busy_frac') if k in e}
        now = library_source_sha16()
        src = {'file': 'profiles/hbm_traffic.json', 'summary': e.get('source'), 'library_commit': e.get('library_commit'),
               'library_source_sha16': e.get('library_source_sha16'), 'library_source_sha16_now': now,
               'stale': e.get('library_source_sha16') != now,
               'taken': e.get('taken'), 'measured_in_this_run': False}
        return e['hbm_bytes_per_launch'], onchip, src
    except Exception:
        return None, None, None


def lds_roofline(eng, step, F, mean_iters, k_ms, dev):
    """On-chip roofline of the float32 min-sum kernel, from pass counts measured in one extra launch
    (the kernel counts its full and its syndrome-only check passes): LDS bytes the passes must move
    against (a) the nominal 128 B/clk/CU and (b) the time the same bytes take at the rates measured
    for the instructions that move them (tools/microbench/peaks.hip)."""
    import torch
    E, N, M = 32704, N_CODE, 1022
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    eng.set_pass_counters(cnt)
    step()
    torch.cuda.synchronize()
    eng.set_pass_counters(None)
    n_full, n_synd = (float(x) / F for x in cnt.cpu().tolist())
    # per frame: one first check pass (no previous messages to read), n_full - 1 later ones, n_synd syndrome-only
    # passes, one variable pass per iteration
    by_instr = {
        'ds_read2_b32': (n_full - 1.0) * 8 * E,                      # posterior + previous message per edge
        'ds_read_b32': 1.0 * 4 * E + n_synd * 4 * E + mean_iters * 4 * E + n_full * 4 * M,   # first pass, syndrome-only, variable-pass messages, arg-min shift
        'ds_write_b32': n_full * (4 * E + 4 * M) + mean_iters * 4 * N,                     # messages + arg-min patch, posteriors
    }
    total = sum(by_instr.values())
    t = k_ms * 1e-3
    nominal = 256 * 128 * 2.4e9
    rate = total * F / t
    out = {'bound': 'lds', 'achieved': rate / 1e12, 'peak': nominal / 1e12, 'unit': 'TB/s', 'frac': rate / nominal,
           'lds_bytes_per_frame': total, 'check_passes_per_frame': n_full, 'syndrome_only_passes_per_frame': n_synd,
           'variable_passes_per_frame': mean_iters, 'measured_in_this_run': True,
           'bytes_per_pass': {'first_check': 8 * E + 8 * M, 'check': 12 * E + 8 * M, 'syndrome_only': 4 * E, 'variable': 4 * E + 4 * N}}
    peaks = next((f for f in (os.path.join(ROOT, 'profiles', n) for n in ('r02_onchip_peaks.json', 'r01_onchip_peaks.json'))
                  if os.path.exists(f)), None)
    if peaks:
        try:
            pk = {e['access']: e['bytes_per_clk_per_cu'] for e in json.load(open(peaks))['lds']}
            rates = {'ds_read2_b32': pk['ds_read2_b32 stride 20'], 'ds_read_b32': pk['ds_read_b32 stride 20'],
                     'ds_write_b32': pk['ds_write_b32 stride 20']}
            clk = sum(by_instr[k] / rates[k] for k in by_instr)          # LDS clocks per frame on one CU at the measured rates
            floor_s = clk * F / 256 / 2.4e9
            out['measured_mix'] = {'rates_bytes_per_clk_per_cu': rates, 'source': 'profiles/' + os.path.basename(peaks),
                                   'lds_time_floor_ms': floor_s * 1e3, 'frac': floor_s / t,
                                   'bytes_by_instruction_per_frame': by_instr}
        except Exception:
            pass
    return out


def fixed_iterations(eng, llr, F, args, algo, spa, hard, iters, conv, ones, stream, rank):
    """Forced 50 iterations (no frame converges at 0 dB): the per-iteration rate without early stop."""
    import torch
    Ff = min(F, 65536)
    eng.channel_awgn_device(llr[:Ff], Ff, 0.0, seed=args.seed + 1, first_frame=0)
    ms = None
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        eng.decode_device(llr, Ff, args.max_iters, prec=args.prec, algo=algo, llr_scale=4.0 if spa else 1.0,
                          d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    out = {'info_gbps': K_INFO * Ff / (ms * 1e-3) / 1e9, 'frames': Ff, 'snr_db': 0.0,
           'mean_iters': float(iters[:Ff].float().mean()), 'kernel_ms': ms}
    eng.channel_awgn_device(llr[:Ff], Ff, args.snr, seed=args.seed, first_frame=rank * F)   # restore the batch
    torch.cuda.synchronize()
    return out


def secondary(eng, llr, F, args, hard, iters, conv, ones, stream, dev):
    """Other configurations of the same path, one warm-up and one timed launch each, after the headline's
    timed region (not part of `value`): info Gbit/s from events on the launch stream, or wall time for the
    sweep driver (which synchronises itself)."""
    import torch
    import ldpc_amd
    from ldpc_amd import codes
    out = []

    def timed(name, e, buf, n_frames, k_info, **kw):
        ms = None
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            e.decode_device(buf, n_frames, args.max_iters, d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones, **kw)
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
        it = iters[:n_frames]
        out.append({'case': name, 'frames': n_frames, 'kernel_ms': ms, 'info_gbps': k_info * n_frames / (ms * 1e-3) / 1e9,
                    'mean_iters': float(it.float().mean()), 'fer': float((ones[:n_frames] != 0).float().mean()),
                    'launch': e.launch_info(algo=kw.get('algo', 0), prec=kw.get('prec', 32))})

    # the headline batch in float64 (the arithmetic that is bit-identical to the reference's CPU decoder) and sum-product
    timed('NearEarth min-sum float64, %.1f dB' % args.snr, eng, llr, F, K_INFO, prec=64)
    timed('NearEarth sum-product float32, %.1f dB' % args.snr, eng, llr, F, K_INFO, prec=32, algo=ldpc_amd.ALGO_SUMPRODUCT,
          llr_scale=4.0 * 10 ** (args.snr / 10))
    out.append(dict(case='NearEarth min-sum float32, all 50 iterations forced (0 dB)',
                    **fixed_iterations(eng, llr, F, args, ldpc_amd.ALGO_MINSUM, False, hard, iters, conv, ones, stream, 0)))
    # a whole sweep point: channel + decode + counters (ldpc_eval_awgn), wall time of the call
    ws = None
    for _ in range(2):
        t0 = time.perf_counter()
        cnt, _ = eng.eval_awgn([args.snr], F, seed=args.seed + 7, prec=32)
        ws = time.perf_counter() - t0
    out.append({'case': 'sweep point: channel + decode + counters in one call, NearEarth min-sum float32, %.1f dB' % args.snr,
                'frames': F, 'wall_ms': ws * 1e3, 'info_gbps': K_INFO * F / ws / 1e9, 'mean_iters': float(cnt[0, 3]) / F,
                'fer': float(cnt[0, 1]) / F})
    # gym-style mutations of NearEarth (ldpc_env.py:293-317): one circulant of weight 3, one of weight 7
    base = codes.near_earth()
    Fm = min(F, 200000)
    for name, code in (('NearEarth with block (0,5) of weight 3, min-sum float32', base.with_block(0, 5, [7, 100, 333])),
                       ('NearEarth with block (1,9) of weight 7, min-sum float32', base.with_block(1, 9, [3, 77, 150, 222, 301, 404, 500]))):
        with ldpc_amd.Decoder(code, device=eng.device) as e2:
            e2.set_stream(stream.cuda_stream)
            timed(name + ', corrected sign rule, %.1f dB' % args.snr, e2, llr, Fm, K_INFO, prec=32, sign_mode=ldpc_amd.SIGN_CORRECTED)
            if 'weight 3' in name:
                timed(name.replace('min-sum', 'sum-product') + ', corrected sign rule, %.1f dB' % args.snr, e2, llr, Fm, K_INFO, prec=32,
                      sign_mode=ldpc_amd.SIGN_CORRECTED, algo=ldpc_amd.ALGO_SUMPRODUCT, llr_scale=4.0 * 10 ** (args.snr / 10))
    # the state of a gym episode after 40 actions (the environment keeps every action's circulant): beyond the
    # message-plane kernels' limits, on the table-driven kernel; a poor code, so most frames run all iterations
    late = codes.gym_episode_state(base, 40, seed=20211227)
    with ldpc_amd.Decoder(late, device=eng.device) as e2:
        e2.set_stream(stream.cuda_stream)
        timed('NearEarth after a gym episode of 40 actions (checks of %s edges), min-sum float32, corrected sign rule, %.1f dB'
              % (' and '.join(str(d) for d in late.row_degrees()), args.snr), e2, llr, min(F, 20000), K_INFO, prec=32,
              sign_mode=ldpc_amd.SIGN_CORRECTED)
    # 802.11n n = 1944 codes (corrected sign rule: the reference's rule does not decode odd check degrees)
    for rate, qc in codes.wifi_1944_all().items():
        with ldpc_amd.Decoder(qc, device=eng.device) as e2:
            e2.set_stream(stream.cuda_stream)
            Fw = min(2_000_000, (llr.numel() // e2.n))
            buf = llr.view(-1)[:Fw * e2.n].view(Fw, e2.n)
            snr = {'1/2': 2.0, '2/3': 3.0, '3/4': 3.5, '5/6': 4.0}[rate]
            e2.channel_awgn_device(buf, Fw, snr, seed=args.seed + 3)
            hw = hard.view(-1)[:Fw * e2.nbytes].view(Fw, e2.nbytes) if hard.numel() >= Fw * e2.nbytes else None
            ms = None
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                e2.decode_device(buf, Fw, args.max_iters, prec=32, sign_mode=ldpc_amd.SIGN_CORRECTED, d_hard=hw)
                e1.record(stream)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
            k_info = e2.n - e2.m
            out.append({'case': '802.11n n=1944 rate %s min-sum float32, corrected sign rule, %.1f dB' % (rate, snr), 'frames': Fw,
                        'kernel_ms': ms, 'info_gbps': k_info * Fw / (ms * 1e-3) / 1e9, 'launch': e2.launch_info(prec=32)})
    # the headline batch back in place
    eng.channel_awgn_device(llr, F, args.snr, seed=args.seed, first_frame=0)
    torch.cuda.synchronize()
    return out


def launch_ranks(n_ranks, script_argv, timeout=None):
    """`python bench.py --gpus N` outside a launcher: start N ranks of `script_argv` under
    torch.distributed.run (one process per GPU, the reference's device fan-out,
    ldpcCUDA.py:891-906) and hand their exit status back.  The children inherit stdout, so
    rank 0's JSON line is this process's output.  Called before this process has imported
    torch or touched HIP: the parent never owns a GPU context."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    env['MASTER_ADDR'] = '127.0.0.1'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(n_ranks)}',
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + list(script_argv)
    sys.stdout.flush()
    try:
        return subprocess.run(cmd, env=env, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=1_000_000, help='frames per GPU per step')
    ap.add_argument('--snr', type=float, default=3.5)
    ap.add_argument('--prec', type=int, default=32, choices=(32, 64))
    ap.add_argument('--max-iters', type=int, default=50)
    ap.add_argument('--algo', choices=('minsum', 'spa'), default='minsum',
                    help="'spa' times the sum-product kernel instead (secondary figure, not the headline metric)")
    ap.add_argument('--cpu-frames', type=int, default=262144)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the untimed secondary configurations measured after the headline region')
    ap.add_argument('--seed', type=int, default=460101)
    ap.add_argument('--print-source-sha', action='store_true', help='print the fingerprint of the library sources and exit')
    ap.add_argument('--fixed50', action='store_true',
                    help='also time one launch at 0 dB, where every frame runs all 50 iterations (extra field; off by '
                         'default so that a profile of the default command sees only identical launches)')
    args = ap.parse_args()
    if args.print_source_sha:
        print(library_source_sha16())
        return

    # --gpus N with no launcher around this process: start the N ranks here (before torch or HIP
    # is touched) and relay rank 0's line.  Under torch.distributed.run the launcher's world wins,
    # and a --gpus that contradicts it is an error rather than a silently mislabelled line.
    if 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        if args.gpus < 1:
            sys.exit('bench.py: --gpus must be >= 1')
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    elif int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks'
                 % (args.gpus, os.environ.get('WORLD_SIZE')))

    # stdout carries exactly one line, the JSON result: RCCL and gloo print banners straight to file
    # descriptor 1 when a process group comes up, so descriptor 1 points at stderr until the result is ready
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    import ldpc_amd
    from ldpc_amd import codes

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # LDPC_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer cards than ranks
    # (ranks share cards, counters are reduced through host memory); the default is RCCL.
    backend = os.environ.get('LDPC_BENCH_BACKEND', 'nccl')
    n_cards = torch.cuda.device_count()
    card = local_rank % max(n_cards, 1) if backend != 'nccl' else local_rank
    # LDPC_BENCH_FORCE_DIST=1: initialise the process group for a single rank too (checks the RCCL
    # path on a one-card box: python -m torch.distributed.run --nproc-per-node 1 bench.py)
    use_dist = world > 1 or (os.environ.get('LDPC_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(card)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', card))
        else:
            dist.init_process_group(backend)
    dev = torch.device('cuda', card)
    torch.cuda.set_device(dev)

    F = args.frames
    eng = ldpc_amd.Decoder(codes.near_earth(), device=card)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    n, nbytes = eng.n, eng.nbytes

    llr = torch.empty((F, n), dtype=torch.float32, device=dev)
    hard = torch.empty((F, nbytes), dtype=torch.uint8, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    conv = torch.empty(F, dtype=torch.uint8, device=dev)
    ones = torch.empty(F, dtype=torch.int32, device=dev)
    unc = torch.empty(F, dtype=torch.int32, device=dev)

    # synthetic input: every rank draws its own frames (frame ids rank*F ...), on the device
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.channel_awgn_device(llr, F, args.snr, seed=args.seed, first_frame=rank * F, d_uncoded=unc)
    torch.cuda.synchronize()
    t_channel = time.perf_counter() - t0

    spa = args.algo == 'spa'
    algo = ldpc_amd.ALGO_SUMPRODUCT if spa else ldpc_amd.ALGO_MINSUM
    llr_scale = 4.0 * 10 ** (args.snr / 10) if spa else 1.0     # sum-product takes 2y/sigma^2

    def step():
        eng.decode_device(llr, F, args.max_iters, prec=args.prec, algo=algo, llr_scale=llr_scale,
                          d_hard=hard, d_iters=iters, d_conv=conv, d_ones=ones)

    def counters():
        o64 = ones.to(torch.int64)
        c = torch.stack([torch.tensor(F, device=dev, dtype=torch.int64),
                         (ones != 0).sum(), o64.sum(), iters.sum(dtype=torch.int64),
                         unc.sum(dtype=torch.int64), (conv == 0).sum(), (o64 * o64).sum()])
        if use_dist:
            if backend == 'nccl':
                dist.all_reduce(c)      # the path's only exchange: error counters (RCCL)
            else:
                h = c.cpu()
                dist.all_reduce(h)
                c = h.to(dev)
        return c

    for _ in range(args.warmup):
        step()
        counters()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        step()
        ev[k][1].record(stream)
        c = counters()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    c = c.cpu().numpy()

    if rank == 0:
        total_frames = int(c[0])
        value = K_INFO * total_frames * args.steps / elapsed / 1e9
        k_ms = float(np.mean(kernel_ms))
        achieved = BYTES_PER_FRAME * F / (k_ms * 1e-3) / 1e9
        info = eng.launch_info(algo=algo, prec=args.prec)
        mean_iters = float(c[3]) / total_frames
        kernel = 'qc_planes_kernel<511, 8, false, false, 0>' if args.prec == 32 else 'qc_regular_kernel<double, 2, 2, 8, 1, false>'
        if spa:
            kernel = 'qc_planes_kernel<511, 8, true, false, 0>' if args.prec == 32 else 'csr_spa_kernel<double>'
        traffic, onchip, traffic_source = committed_counters(args.prec, F, spa)
        lds = None
        if args.prec == 32 and not spa:
            lds = lds_roofline(eng, step, F, mean_iters, k_ms, dev)
        fixed = None
        if args.fixed50:
            fixed = fixed_iterations(eng, llr, F, args, algo, spa, hard, iters, conv, ones, stream, rank)
        fer, ber = float(c[1]) / total_frames, float(c[2]) / (total_frames * N_CODE)
        # 95 % intervals: Wilson for the frame error rate; for the bit error rate the frame is the sampling unit
        # (bit errors come in bursts inside failed frames), so the spread of errors per frame sets the width
        var_frame = max(float(c[6]) / total_frames - (float(c[2]) / total_frames) ** 2, 0.0)
        ber_half = 1.96 * (var_frame / total_frames) ** 0.5 / N_CODE
        out = {
            'metric': 'decoded info Gbit/s, NearEarth (8176,7154) %s @50 iters' % ('sum-product' if spa else 'min-sum'),
            'value': value, 'unit': 'Gbit/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32' if args.prec == 32 else 'f64', 'data': 'synthetic',
            'config': {'workload': f'NearEarth (8176,7154) {"sum-product" if spa else "min-sum"}, <=%d iterations with early stop, BPSK/AWGN '
                                   f'%.2f dB (reference SNR definition), %d frames per GPU resident in HBM, '
                                   f'LLR float32 in / packed hard bits + iterations + status out'
                                   % (args.max_iters, args.snr, F),
                       'frames_per_gpu': F, 'snr_db': args.snr, 'max_iters': args.max_iters,
                       'parallelism': f'frames sharded over {world} GPU(s), counters all-reduced'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': kernel, 'kernel_ms': k_ms,
                         'algorithmic_bytes_per_frame': BYTES_PER_FRAME, 'onchip': onchip,
                         'note': 'message passing runs out of LDS and registers; HBM carries only LLR-in and '
                                 'results-out, so the HBM fraction is small by design.  The kernel is bound by '
                                 'vector-instruction issue and LDS bandwidth (roofline_lds: measured in this run; '
                                 'traffic / onchip: from the committed counter passes named in traffic_source, '
                                 'not measured in this run)'},
            'roofline_lds': lds,
            'fixed_iterations': fixed,
            'stats': {'frames': total_frames, 'mean_iters': mean_iters,
                      'fer': fer, 'fer_ci95': wilson(int(c[1]), total_frames),
                      'ber': ber, 'ber_ci95': [max(ber - ber_half, 0.0), ber + ber_half],
                      'uncoded_ber': float(c[4]) / (total_frames * N_CODE),
                      'not_codeword': int(c[5])},
            'launch': info,
            'codeword_gbps': value * N_CODE / K_INFO,
            'frames_per_s': total_frames * args.steps / elapsed,
            'channel_fill_s': t_channel,
        }
        if world == 1 and not args.no_secondary and not spa and args.prec == 32:
            out['secondary'] = secondary(eng, llr, F, args, hard, iters, conv, ones, stream, dev)
        if not args.no_cpu and world == 1 and not spa:
            ncpu = min(16, os.cpu_count() or 1)
            m = min(args.cpu_frames, F)
            SPA_ENGINE[0] = eng
            eng.set_stream(None)
            base, it_cpu = cpu_baseline(llr, m, ncpu)
            base.update(cpu_model=cpu_model(), os_cpu_count=os.cpu_count(),
                        cores_note='OpenMP threads used = min(16, os.cpu_count()): the GPU box gives one job 16 host threads')
            out['cpu_baseline'] = base
            if args.prec == 64:
                out['cpu_gpu_iters_equal'] = bool(np.array_equal(it_cpu, iters[:m].cpu().numpy()))
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
