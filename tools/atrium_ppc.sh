#!/bin/bash
# headline workload: Mrays/s against the number of sample passes that share the launches (N = 1)
mkdir -p gpurun_out
for b in 1 2 4 8; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-fast-tree --steps 8 --warmup $b --passes-per-call $b > gpurun_out/atrppc_$b.json 2> gpurun_out/atrppc_$b.err || { echo "B=$b FAILED"; tail -3 gpurun_out/atrppc_$b.err; exit 1; }
  python - $b <<'PY'
import json,sys
b=sys.argv[1]; a=json.load(open(f'gpurun_out/atrppc_{b}.json')); r=a['roofline']
print(f"B={b}: {a['value']:.1f} Mrays/s ms/step {a['ms_per_step']:.2f} trace/step {r['kernel_ms']['trace']/a['steps']:.2f} shadow/step {r['kernel_ms']['shadow']/a['steps']:.2f} shade/step {r['kernel_ms']['shade']/a['steps']:.2f} launches {r['launches']}")
PY
done
