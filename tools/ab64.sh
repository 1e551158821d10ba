for r in 1 2 3; do for L in ldpc_amd/libldpc_hip_planes.so ldpc_amd/libldpc_hip_aos.so; do
  out=$(LDPC_HIP_LIBRARY=$PWD/$L timeout -k 10 200 python3 bench.py --prec 64 --steps 3 --no-cpu --no-secondary 2>/dev/null) || { echo failed $L; exit 1; }
  echo "$L $(echo "$out" | grep -o "\"kernel_ms\": [0-9.]*" | head -1)"; done; done
