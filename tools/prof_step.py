import sys, os, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import numpy as np
from ldpc_amd import codes, gym_eval
base = codes.gym_episode_state(codes.near_earth(), 40, seed=20211227)
prng = np.random.RandomState(1)
ev = gym_eval.CodeEvaluator(seed=3, deviceRng=True)
ev.step(base)
qcs = [base.with_block(int(prng.randint(2)), int(prng.randint(16)), sorted(prng.choice(511, int(prng.randint(1, 8)), replace=False).tolist())) for _ in range(100)]
pr = cProfile.Profile(); pr.enable()
for qc in qcs: ev.step(qc)
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumtime').print_stats(28); print(s.getvalue()[:5000])
