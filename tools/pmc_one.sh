#!/bin/bash
# usage: [LDPC_HIP_LIBRARY=...] tools/pmc_one.sh <outdir> <prof_decode.py args...>   (GPU box, repo root)
# One counter pass (instruction counts, LDS activity and bank conflicts) of one decode launch; prints the sums
# over the decode kernel's dispatches.
set -u
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT" -- python3 tools/prof_decode.py "$@" > "$OUT/run.log" 2>&1 || { echo "pass failed"; tail -3 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'minsum' in k or 'regular' in k or 'planes' in k or 'heavy' in k or 'spa' in k:
            tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
for k in sorted(tot): print(f'{k:24s} {tot[k]:16.0f}  (dispatches {n[k]})')
if tot.get('SQ_LDS_IDX_ACTIVE'): print('bank conflicts / LDS active cycles: %.3f' % (tot['SQ_LDS_BANK_CONFLICT'] / tot['SQ_LDS_IDX_ACTIVE']))
PY
