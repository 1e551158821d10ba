#!/bin/bash
# usage: [PROF_SCRIPT=tools/prof_sweep.py] tools/pmc_passes.sh <outdir> <script args...>   (run from the repo root on the GPU box)
# One rocprofv3 --pmc run per counter group (no trace options), results as CSV under <outdir>/passN.
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $group --output-format csv -d "$OUT/pass$i" -- python3 ${PROF_SCRIPT:-tools/prof_decode.py} "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/pass$i.log"; exit 1; }
done <<GROUPS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE GRBM_COUNT FETCH_SIZE
WRITE_SIZE SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
GROUPS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'minsum' in k or 'regular' in k or 'planes' in k or 'heavy' in k or 'spa' in k:
            tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
for k in sorted(tot): print(f'{k:28s} {tot[k]:18.0f}  (dispatches {n[k]})')
PY
