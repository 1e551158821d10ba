set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r01
timeout -k 10 300 python3 bench.py > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.err || { echo bench failed; exit 1; }
timeout -k 10 300 python3 bench.py --fixed50 --no-cpu > gpurun_out/r01/bench_fixed50.json 2> gpurun_out/r01/bench_fixed50.err || { echo fixed50 failed; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01/prof -- python3 bench.py --no-cpu > gpurun_out/r01/bench_under_rocprof.json 2> gpurun_out/r01/prof.err || { echo rocprof failed; exit 1; }
bash tools/pmc_passes.sh gpurun_out/r01/pmc --frames 1000000 --snr 3.5 --budget 50 > gpurun_out/r01/pmc.txt 2>&1 || { echo pmc failed; tail -5 gpurun_out/r01/pmc.txt; exit 1; }
timeout -k 10 240 python3 tools/bench_codes.py > gpurun_out/r01/secondary.json 2> gpurun_out/r01/secondary.err || { echo secondary failed; exit 1; }
echo done
