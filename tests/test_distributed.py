"""N > 1 path: frames of a sweep split over ranks, counters all-reduced, per-frame data
all-gathered; variants round-robined.  CPU part: world_size 2 and 3 over gloo with a
synthetic local evaluation.  GPU part: two ranks sharing the card, real engine -- the
counters must not depend on the number of ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'dist_worker.py')


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(n, mode, tmp_path):
    out = tmp_path / f'{mode}_{n}.json'
    if n == 1:
        cmd = [sys.executable, WORKER, mode, str(out)]
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
               '--master-addr', '127.0.0.1', '--master-port', str(free_port()), WORKER, mode, str(out)]
        env = dict(os.environ)
    subprocess.run(cmd, check=True, timeout=600, env=env, cwd=ROOT)
    return json.load(open(out))


def test_frame_range_partitions():
    from ldpc_amd.distributed import frame_range
    for total in (0, 1, 7, 37, 1000):
        for ws in (1, 2, 3, 8):
            parts = [frame_range(total, r, ws) for r in range(ws)]
            assert sum(c for _, c in parts) == total
            pos = 0
            for first, c in parts:
                assert first == pos
                pos += c
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


@pytest.mark.parametrize('ws', [2, 3])
def test_sharded_sweep_equals_single_process_gloo(tmp_path, ws):
    one = run_world(1, 'fake', tmp_path)
    many = run_world(ws, 'fake', tmp_path)
    assert one['cnt'] == many['cnt']
    assert one['pf'] == many['pf']            # global frame order restored by the gather
    assert one['var'] == many['var']
    assert np.array(one['cnt'])[:, 0].tolist() == [37, 37, 37]


@pytest.mark.gpu
def test_sharded_sweep_equals_single_process_on_gpu(tmp_path):
    one = run_world(1, 'gpu', tmp_path)
    two = run_world(2, 'gpu', tmp_path)
    assert one['cnt'] == two['cnt']
    assert one['pf'] == two['pf']
    assert one['var'] == two['var']
    cnt = np.array(one['cnt'])
    assert cnt[:, 0].tolist() == [301, 301] and cnt[1, 1] <= cnt[0, 1]


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no launcher): two ranks on the one card (gloo rehearsal),
    one JSON line with n_gpus 2, and counters equal to one rank decoding the same 2F frames."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['LDPC_BENCH_BACKEND'] = 'gloo'

    def run(gpus, frames):
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--frames', str(frames),
                            '--steps', '2', '--warmup', '1', '--no-cpu', '--no-secondary'], capture_output=True, text=True,
                           timeout=600, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1
        return json.loads(lines[0])
    two = run(2, 20000)
    one = run(1, 40000)
    assert two['n_gpus'] == 2 and one['n_gpus'] == 1
    assert two['stats'] == one['stats'] and two['stats']['frames'] == 40000


def _cards():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU runtime)


@pytest.mark.gpu
def test_sharded_sweep_over_rccl_one_rank_per_card(tmp_path):
    """The reference's device fan-out (ldpcCUDA.py:891-906) as this package does it: one process per card,
    frames sharded by global frame number, counters all-reduced and per-frame columns all-gathered over RCCL.
    Skipped on a one-card box; on a node with several cards the result must equal one rank's."""
    n = _cards()
    if n < 2:
        pytest.skip('one rank per GPU over RCCL: needs two cards or more')
    one = run_world(1, 'gpu', tmp_path)
    for ws in sorted({2, min(n, 8)}):
        many = run_world(ws, 'nccl', tmp_path)
        assert one['cnt'] == many['cnt'], ws
        assert one['pf'] == many['pf'], ws
        assert one['var'] == many['var'], ws


@pytest.mark.gpu
def test_bench_over_rccl_equals_gloo_rehearsal():
    """Plain `python bench.py --gpus 2`, one rank per card, counters all-reduced over RCCL: the same statistics as
    the two-rank gloo rehearsal on one card and as one rank decoding all the frames.  Skipped on a one-card box."""
    if _cards() < 2:
        pytest.skip('one rank per GPU over RCCL: needs two cards or more')
    base = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LDPC_BENCH_BACKEND')}

    def run(gpus, frames, backend):
        env = dict(base)
        if backend:
            env['LDPC_BENCH_BACKEND'] = backend
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--frames', str(frames),
                            '--steps', '2', '--warmup', '1', '--no-cpu', '--no-secondary'], capture_output=True, text=True,
                           timeout=600, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1
        return json.loads(lines[0])
    rccl = run(2, 20000, None)
    gloo = run(2, 20000, 'gloo')
    one = run(1, 40000, None)
    assert rccl['n_gpus'] == 2 and rccl['config']['parallelism'].startswith('frames sharded over 2')
    assert rccl['stats'] == gloo['stats'] == one['stats'] and rccl['stats']['frames'] == 40000
