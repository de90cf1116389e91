#!/bin/bash
# 10 M-triangle atrium: Mrays/s against the XCD chunk size (blocks)
mkdir -p gpurun_out
for c in 0 8 32 256 2048; do
  SRX_XCD_CHUNK=$c timeout -k 10 300 python bench.py --tris 10000000 --no-cpu-baseline --no-fast-tree --steps 3 --fast-build > gpurun_out/tenm_$c.json 2> gpurun_out/tenm_$c.err || { echo "chunk $c FAILED"; tail -3 gpurun_out/tenm_$c.err; exit 1; }
  python - $c <<'PY'
import json,sys
c=sys.argv[1]; a=json.load(open(f'gpurun_out/tenm_{c}.json')); r=a['roofline']
print(f"chunk {c}: {a['value']:.1f} Mrays/s ms/step {a['ms_per_step']:.2f} trace {r['kernel_ms']['trace']:.1f} shadow {r['kernel_ms']['shadow']:.1f}")
PY
done
