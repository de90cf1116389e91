#!/bin/bash
# diffuse box (C5): Mrays/s against the number of sample passes that share the launches
mkdir -p gpurun_out
for b in 1 2 4 8; do
  timeout -k 10 200 python bench.py --scene diffuse_box --tris 50000 --no-cpu-baseline --no-fast-tree --steps 16 --warmup 8 --passes-per-call $b > gpurun_out/boxppc_$b.json 2> gpurun_out/boxppc_$b.err || { echo "B=$b FAILED"; tail -3 gpurun_out/boxppc_$b.err; exit 1; }
  python - $b <<'PY'
import json,sys
b=sys.argv[1]; a=json.load(open(f'gpurun_out/boxppc_{b}.json')); r=a['roofline']
print(f"B={b}: {a['value']:.1f} Mrays/s ms/step {a['ms_per_step']:.2f} trace/step {r['kernel_ms']['trace']/a['steps']:.2f} shade/step {r['kernel_ms']['shade']/a['steps']:.2f} launches {r['launches']}")
PY
done
