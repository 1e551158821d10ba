#!/bin/bash
# usage: tools/ab.sh <reps> <lib.so> [<lib.so> ...]   (on the GPU box, from the repo root)
# Alternates bench.py runs between builds of the library on the same card and prints the kernel
# time of each run: box-to-box spread (about 1 %) is larger than most single optimisations.
set -u
REPS=$1; shift
for r in $(seq $REPS); do
  for L in "$@"; do
    out=$(LDPC_HIP_LIBRARY=$PWD/$L timeout -k 10 200 python3 bench.py --steps 5 --no-cpu --no-secondary 2>/dev/null) || { echo "run failed: $L"; exit 1; }
    echo "$L $(echo "$out" | grep -o "\"kernel_ms\": [0-9.]*" | head -1)"
  done
done
