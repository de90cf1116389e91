#!/bin/bash
# bench variants of one build: name=ENV... per line -> gpurun_out/r2_var_<name>.json
set -u
mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --steps 4 > gpurun_out/r2_var_$name.json 2> gpurun_out/r2_var_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/r2_var_$name.err; return 1; }
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]; a=json.load(open(f'gpurun_out/r2_var_{n}.json')); r=a['roofline']
print(f"{n:28s} {a['value']:8.1f} Mrays/s  trace {r['kernel_ms']['trace']:.1f} shadow {r['kernel_ms']['shadow']:.1f} shade {r['kernel_ms']['shade']:.1f} wall {r['kernel_ms']['wall']:.1f}  packets/ray {r['packet_tests_per_ray']:.2f}")
PY
}
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  run $name ${envs//,/ } X=1 || exit 1
done
