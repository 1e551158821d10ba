#!/bin/bash
# usage: tools/refresh_profiles.sh <round tag, e.g. r02> <commit of the library being profiled>
# (on the GPU box, from the repo root).  Bench line, rocprofv3 kernel statistics of the same command, counter
# passes of the decode launch and of a fused sweep point, secondary configurations; everything under
# gpurun_out/<tag>/, from where the summaries are copied into profiles/ by hand.
set -u
export TMPDIR=/tmp
R=${1:-r02}; COMMIT=${2:-unknown}
O=gpurun_out/$R
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu --no-secondary > $O/bench_under_rocprof.json 2> $O/prof.err || { echo rocprof failed; exit 1; }
bash tools/pmc_passes.sh $O/pmc --frames 1000000 --snr 3.5 --budget 50 > $O/pmc.txt 2>&1 || { echo pmc failed; tail -5 $O/pmc.txt; exit 1; }
python3 tools/pmc_summary.py $O/pmc 1000000 33735 $O/f32_planes_pmc_summary.json "tools/pmc_passes.sh: python3 tools/prof_decode.py --frames 1000000 --snr 3.5 --budget 50 (library commit $COMMIT)" > /dev/null || { echo summary failed; exit 1; }
PROF_SCRIPT=tools/prof_sweep.py bash tools/pmc_passes.sh $O/pmc_sweep --frames 1000000 --snr 3.5 > $O/pmc_sweep.txt 2>&1 || { echo pmc sweep failed; tail -5 $O/pmc_sweep.txt; exit 1; }
python3 tools/pmc_summary.py $O/pmc_sweep 1000000 21 $O/f32_fused_sweep_pmc_summary.json "tools/pmc_passes.sh: python3 tools/prof_sweep.py --frames 1000000 --snr 3.5 (library commit $COMMIT); algorithmic bytes per frame = iterations 4 + status 1 + ones 4 + uncoded 4 + noise sum 8" > /dev/null || { echo sweep summary failed; exit 1; }
timeout -k 10 240 python3 tools/bench_codes.py > $O/secondary.json 2> $O/secondary.err || { echo secondary failed; exit 1; }
timeout -k 10 240 python3 tools/bench_heavy.py > $O/heavy.json 2> $O/heavy.err || { echo heavy failed; exit 1; }
python3 - $O $COMMIT $(python3 bench.py --print-source-sha) <<'PY'
import json, sys, time
o, commit, sha = sys.argv[1], sys.argv[2], sys.argv[3]
s = json.load(open(o + '/f32_planes_pmc_summary.json'))['derived']
json.dump({'prec32_frames1000000': {'hbm_bytes_per_launch': s['hbm_bytes_per_launch'], 'source': 'profiles/%s_f32_planes_pmc_summary.json' % o.split('/')[-1],
                                    'library_commit': commit, 'library_source_sha16': sha, 'taken': time.strftime('%Y-%m-%d'),
                                    'valu_busy_frac': s['valu_busy_frac'], 'lds_busy_frac': s['lds_busy_frac']}},
          open(o + '/hbm_traffic.json', 'w'), indent=1)
PY
# the bench line last, reading the counter passes just taken (traffic_source.stale: false)
cp $O/hbm_traffic.json profiles/hbm_traffic.json
cp $O/f32_planes_pmc_summary.json profiles/${R}_f32_planes_pmc_summary.json
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { echo bench failed; exit 1; }
echo done
