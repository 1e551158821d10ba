"""Stand-in rank program for the test of bench.launch_ranks: joins a gloo group, sums a
per-rank counter, rank 0 prints one JSON line.  `fail` makes rank 1 exit non-zero."""
import json
import os
import sys

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else 'ok'
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if mode == 'fail' and rank == 1:
    sys.exit(3)
dist.init_process_group('gloo')
t = torch.tensor([rank + 1, 100], dtype=torch.int64)
dist.all_reduce(t)
if rank == 0:
    print(json.dumps({'n_gpus': world, 'sum': t.tolist(), 'master': os.environ.get('MASTER_ADDR')}), flush=True)
dist.destroy_process_group()
