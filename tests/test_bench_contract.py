"""The committed bench line (profiles/r02_f32_planes_bench.json, written by `python bench.py` on an
MI355X) carries every field the measurement contract names, and the profile evidence beside it is
consistent with it: the rocprofv3 kernel average agrees with the bench's own kernel time, and the
HBM traffic from the counter passes is the algorithmic traffic to within a few percent."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')


def _load(name):
    with open(os.path.join(PROF, name)) as f:
        return json.load(f)


def test_bench_line_fields():
    d = _load('r02_f32_planes_bench.json')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'Gbit/s' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['unit'] == d['unit'] and 0 < c['value'] < d['value']
    # value = info bits of all frames of all steps / wall time
    frames = d['config']['frames_per_gpu'] * d['n_gpus']
    assert abs(d['value'] - 7154 * frames / (d['ms_per_step'] * 1e-3) / 1e9) / d['value'] < 1e-6
    # achieved = algorithmic bytes per launch / kernel time
    assert abs(r['achieved'] - r['algorithmic_bytes_per_frame'] * d['config']['frames_per_gpu'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6 * r['achieved']


def test_rocprof_average_agrees_with_bench_kernel_time():
    under = _load('r02_f32_planes_bench_under_rocprof.json')
    with open(os.path.join(PROF, 'r02_f32_planes_kernel_stats.csv')) as f:
        rows = list(csv.DictReader(f))
    k = [r for r in rows if 'qc_planes_kernel' in r['Name']]
    assert len(k) == 1 and int(k[0]['Calls']) == under['steps'] + under['warmup'] + 1      # + the launch that counts check passes
    avg_ms = float(k[0]['AverageNs']) * 1e-6
    assert abs(avg_ms - under['roofline']['kernel_ms']) / avg_ms < 0.01
    assert float(k[0]['Percentage']) > 90.0                      # the decode kernel is the run


def test_counter_traffic_is_the_algorithmic_traffic():
    s = _load('r02_f32_planes_pmc_summary.json')['derived']
    assert 0.99 < s['traffic_over_algorithmic'] < 1.05
    d = _load('r02_f32_planes_bench.json')
    # (the bench reads the traffic from the counter summary committed at the time it runs)
    assert abs(d['roofline']['traffic'] - s['hbm_bytes_per_launch']) < 1e-3 * s['hbm_bytes_per_launch']


# ---- `bench.py --gpus N` started without a launcher spawns its own ranks
import subprocess
import sys

STUB = os.path.join(ROOT, 'tests', 'bench_rank_stub.py')


def _launch(mode, n=2):
    code = ("import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules; "
            "sys.exit(bench.launch_ranks(%d, [%r, %r], timeout=300))" % (ROOT, n, STUB, mode))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    return subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)


def test_launch_ranks_relays_rank0_line():
    p = _launch('ok', 2)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    assert lines[0] == {'n_gpus': 2, 'sum': [3, 200], 'master': '127.0.0.1'}


def test_launch_ranks_reports_a_failed_rank():
    p = _launch('fail', 2)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith('{')]


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and 'WORLD_SIZE=4' in p.stderr and p.stdout == ''
