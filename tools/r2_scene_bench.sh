#!/bin/bash
# bench lines of the other workloads: tools/r2_scene_bench.sh name:scene:ENV,... ...
mkdir -p gpurun_out
for v in "$@"; do
  name=${v%%:*}; rest=${v#*:}; scene=${rest%%:*}; envs=${rest#*:}
  env ${envs//,/ } X=1 timeout -k 10 200 python bench.py --scene $scene --no-cpu-baseline --steps 5 > gpurun_out/r2_sc_$name.json 2> gpurun_out/r2_sc_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/r2_sc_$name.err; exit 1; }
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]; a=json.load(open(f'gpurun_out/r2_sc_{n}.json')); r=a['roofline']
print(f"{n:24s} {a['value']:8.1f} Mrays/s  ms/step {a['ms_per_step']:.2f} trace {r['kernel_ms']['trace']:.1f} shadow {r['kernel_ms']['shadow']:.1f} shade {r['kernel_ms']['shade']:.1f} wall {r['kernel_ms']['wall']:.1f} launches {r['launches']}")
PY
done
