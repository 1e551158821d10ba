#!/bin/bash
# usage: tools/pmc_kernels.sh <outdir>    (on the GPU box, from the repo root)
# Counter passes (tools/pmc_passes.sh) of one decode launch per kernel family, condensed by tools/pmc_summary.py
# into <outdir>/<case>.json.  Algorithmic bytes per frame: LLR in (4 B per position) + hard bits out + 9 B of
# iterations / status / ones.
set -u
O=$1; mkdir -p $O
run() {   # name frames bytes_per_frame args...
  local name=$1 frames=$2 bytes=$3; shift 3
  bash tools/pmc_passes.sh $O/$name --frames $frames "$@" > $O/$name.txt 2>&1 || { echo "$name: passes failed"; tail -3 $O/$name.txt; return 1; }
  python3 tools/pmc_summary.py $O/$name $frames $bytes $O/$name.json "tools/pmc_passes.sh: python3 tools/prof_decode.py --frames $frames $*" > /dev/null || { echo "$name: summary failed"; return 1; }
  echo "$name done"
}
run f64_regular        200000 33735 --prec 64 --snr 3.5 &&
run f32_spa_planes     200000 33735 --algo spa --snr 3.5 &&
run f32_heavy_w3       200000 33735 --code nearEarth_w3 --sign-mode 1 --snr 3.5 &&
run f32_heavy_w7       200000 33735 --code nearEarth_w7 --sign-mode 1 --snr 3.5 &&
run f32_heavyqc_ep40   20000  33735 --code nearEarth_episode40 --sign-mode 1 --snr 3.5 &&
LDPC_HIP_TEST_HOOKS=1 LDPC_HIP_NO_HEAVYQC=1 run f32_table_driven_ep40 20000 33735 --code nearEarth_episode40 --sign-mode 1 --snr 3.5 &&
run f32_spa_heavy_w3   200000 33735 --code nearEarth_w3 --algo spa --sign-mode 1 --snr 3.5 &&
run f32_packed_wifi12  1000000 8028 --code wifi_1/2 --sign-mode 1 --snr 2.0 &&
run f32_slots_wifi56   1000000 8028 --code wifi_5/6 --sign-mode 1 --snr 4.0 &&
echo all done
