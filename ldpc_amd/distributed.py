"""One process per GPU: the Monte-Carlo sweep sharded over ranks.

Frames, SNR points and code variants are independent, so the decode path has no data
exchange at all; the only collective is the sum of the error counters (a few hundred bytes:
latency-bound on xGMI, link bandwidth is irrelevant) and, when the caller wants the
per-frame scatter data the gym reward is fitted on, one all-gather of float32 [frames, 4].
The reference spreads the same work as nTx // nDevices frames per device with one forked
process per device (ldpcCUDA.py:891-932); here the noise is a counter-based stream indexed
by the GLOBAL frame number, so the counters do not depend on the number of ranks.

Launch with ``python -m torch.distributed.run --nproc-per-node N ...`` (RANK, LOCAL_RANK,
WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the environment).  Backend "nccl" is RCCL.
"""
import os

import numpy as np

from .engine import ALGO_MINSUM, SIGN_REFERENCE, Decoder, device_count


def world():
    """(rank, local_rank, world_size) from the launcher's environment (1 process if absent)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend=None):
    """Join the process group (no-op for a single process).  backend None: "nccl" when every
    local rank has a GPU of its own, else "gloo"."""
    rank, local_rank, ws = world()
    if ws == 1:
        return
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend is None:
        n_dev = torch.cuda.device_count()
        backend = 'nccl' if n_dev > local_rank and n_dev >= int(os.environ.get('LOCAL_WORLD_SIZE', ws)) else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group(backend)


def local_device():
    """Card of this rank: LOCAL_RANK, wrapped when there are more local ranks than cards
    (ranks then share a card; meant for rehearsals on a one-GPU box)."""
    return world()[1] % max(device_count(), 1)


def _require_group():
    """The launcher started several ranks (WORLD_SIZE > 1): their shards are only a result once they
    have been summed, so a forgotten init() is an error, not a silent per-rank answer."""
    if world()[2] > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('WORLD_SIZE > 1 but torch.distributed is not initialised: call '
                               'ldpc_amd.distributed.init() (or init_process_group) before sweep / evaluate_variants')


def frame_range(total, rank, world_size):
    """Contiguous share of `total` frames: (first, count); the first total % world ranks get one more."""
    base, rem = divmod(int(total), int(world_size))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def _reduce_device():
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def allreduce_counters(counters):
    """Element-wise sum over ranks of a uint64 array (returned as uint64, same shape)."""
    import torch
    import torch.distributed as dist
    c = np.ascontiguousarray(counters, dtype=np.uint64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return c.copy()
    t = torch.from_numpy(c.view(np.int64).copy()).to(_reduce_device())
    dist.all_reduce(t)
    return t.cpu().numpy().view(np.uint64).reshape(c.shape)


def allgather_frames(local, counts, axis=1):
    """Concatenate every rank's per-frame block along `axis` in rank order.  counts[r] =
    frames of rank r (known to all ranks from frame_range)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.array(local, copy=True)
    local = np.ascontiguousarray(np.moveaxis(local, axis, 0), dtype=np.float32)   # frames first
    width = max(counts)
    pad = np.zeros((width,) + local.shape[1:], np.float32)
    pad[:local.shape[0]] = local
    dev = _reduce_device()
    mine = torch.from_numpy(pad).to(dev)
    parts = [torch.empty_like(mine) for _ in counts]
    dist.all_gather(parts, mine)
    out = np.concatenate([p.cpu().numpy()[:k] for p, k in zip(parts, counts)], axis=0)
    return np.moveaxis(out, 0, axis)


def sweep(code, snr_db, frames_per_snr, seed, max_iters=50, prec=32, algo=ALGO_MINSUM,
          sign_mode=SIGN_REFERENCE, per_frame=False, evaluate=None, engine=None):
    """BER/FER sweep with the frames of every SNR point split over the ranks.

    Returns (counters uint64 [n_snr, 6] summed over ranks, per_frame float32
    [n_snr, frames_per_snr, 4] in global frame order or None) on EVERY rank.
    `evaluate(snr, count, seed, first_frame, per_frame)` may replace the local engine call
    (it must return what Decoder.eval_awgn returns)."""
    _require_group()
    rank, local_rank, ws = world()
    first, count = frame_range(frames_per_snr, rank, ws)
    snr = np.atleast_1d(np.asarray(snr_db, dtype=np.float64))
    own = None
    if evaluate is None:
        eng = engine
        if eng is None:
            eng = own = Decoder(code, device=local_device())

        def evaluate(snr, count, seed, first_frame, per_frame):
            return eng.eval_awgn(snr, count, seed, first_frame=first_frame, max_iters=max_iters, prec=prec,
                                 algo=algo, sign_mode=sign_mode, per_frame=per_frame)
    try:
        if count > 0:
            cnt, pf = evaluate(snr, count, seed, first, per_frame)
        else:
            cnt = np.zeros((len(snr), 6), np.uint64)
            pf = np.zeros((len(snr), 0, 4), np.float32) if per_frame else None
    finally:
        if own is not None:
            own.close()
    cnt = allreduce_counters(cnt)
    if per_frame:
        pf = allgather_frames(pf, [frame_range(frames_per_snr, r, ws)[1] for r in range(ws)], axis=1)
    return cnt, pf


def evaluate_variants(variants, snr_db, frames_per_snr, seed, max_iters=50, prec=32, algo=ALGO_MINSUM,
                      sign_mode=SIGN_REFERENCE, evaluate=None):
    """Many parity-check matrices, each evaluated whole on one rank (variant v on rank
    v mod world): the batch evaluation of circulant-zeroed NearEarth variants
    (testMatricesGeneratorScript.py:23-34) or of the matrices a gym episode proposes.
    Every variant sees the same noise (same seed, frames 0..frames_per_snr-1), as the
    reference's environment re-seeds per evaluation (ldpc_env.py:367-371).
    Returns counters uint64 [n_variants, n_snr, 6] on every rank."""
    _require_group()
    rank, local_rank, ws = world()
    snr = np.atleast_1d(np.asarray(snr_db, dtype=np.float64))
    out = np.zeros((len(variants), len(snr), 6), np.uint64)
    for v in range(rank, len(variants), ws):
        if evaluate is not None:
            out[v] = evaluate(variants[v], snr, frames_per_snr, seed)
        else:
            with Decoder(variants[v], device=local_device()) as eng:
                out[v], _ = eng.eval_awgn(snr, frames_per_snr, seed, max_iters=max_iters, prec=prec,
                                          algo=algo, sign_mode=sign_mode)
    return allreduce_counters(out)
