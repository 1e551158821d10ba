#!/usr/bin/env python3
"""Two (or N) processes sum their sweep counters through the C library's own communicator
(ldpc_comm_* / ldpc_allreduce_counters), no torch.distributed: rank 0 writes the id to a file, the
others read it.  usage: comm_two_ranks.py <world> [<id file>]; every rank uses GPU rank % device_count."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank_main(rank, world, path):
    import numpy as np
    import ldpc_amd
    from ldpc_amd import codes
    if rank == 0:
        ident = ldpc_amd.comm_unique_id()
        with open(path + '.tmp', 'wb') as f:
            f.write(ident)
        os.replace(path + '.tmp', path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 60:
                raise SystemExit('no id file')
            time.sleep(0.05)
        ident = open(path, 'rb').read()
    dev = rank % ldpc_amd.device_count()
    F = 2000
    with ldpc_amd.Comm(dev, rank, world, ident) as comm, ldpc_amd.Decoder(codes.near_earth(), device=dev) as eng:
        cnt, _ = eng.eval_awgn([3.0, 3.4], F, seed=9, first_frame=rank * F)
        total = comm.allreduce(cnt)
        if rank == 0:
            whole, _ = eng.eval_awgn([3.0, 3.4], F * world, seed=9)
            print('sum over ranks', total.tolist())
            print('one rank, all frames', whole.tolist())
            assert np.array_equal(total, whole)
            print('OK')


if __name__ == '__main__':
    if 'LDPC_COMM_RANK' in os.environ:
        rank_main(int(os.environ['LDPC_COMM_RANK']), int(sys.argv[1]), sys.argv[2])
    else:
        world = int(sys.argv[1])
        path = sys.argv[2] if len(sys.argv) > 2 else '/tmp/ldpc_comm_id_%d' % os.getpid()
        if os.path.exists(path):
            os.remove(path)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(world), path],
                                  env=dict(os.environ, LDPC_COMM_RANK=str(r))) for r in range(world)]
        rcs = []
        for p in procs:
            try:
                rcs.append(p.wait(timeout=120))
            except subprocess.TimeoutExpired:
                p.kill()
                rcs.append(124)
        if os.path.exists(path):
            os.remove(path)
        sys.exit(max(rcs))
