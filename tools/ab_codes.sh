#!/bin/bash
# usage: tools/ab_codes.sh <lib.so> [<lib.so> ...]   (on the GPU box, from the repo root)
# tools/bench_codes.py once per library build, on the same card: case name and info Gbit/s.
set -u
for L in "$@"; do
  echo "== $L"
  LDPC_HIP_LIBRARY=$PWD/$L timeout -k 10 240 python3 tools/bench_codes.py 2>/dev/null | python3 -c "import json,sys; [print(' ', c['case'], round(c['info_gbps'],2)) for c in json.load(sys.stdin)]" || { echo failed; exit 1; }
done
