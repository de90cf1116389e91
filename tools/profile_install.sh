#!/bin/bash
# Copies what tools/profile_refresh.sh <tag> collected into profiles/ and re-derives the JSON summaries. Usage: profile_install.sh <tag>
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/$1
S1=$(ls $O/stats1/vm/*kernel_stats.csv); S2=$(ls $O/stats2/vm/*kernel_stats.csv)
cp $O/pmc_raw.txt $R/profiles/r01_pmc_raw.txt
cp $S1 $R/profiles/r01_kernel_stats_single_stream.csv
cp $S2 $R/profiles/r01_kernel_stats.csv
cp $O/bench.json $R/profiles/r01_bench.json
python3 $R/tools/pmc_derive.py $O/pmc_raw.txt $S1 4 > $R/profiles/r01_pmc_derived.json
python3 - "$R" <<'PY'
import json, sys
R = sys.argv[1]
t = json.load(open(R + "/profiles/r01_traffic.json"))
d = json.load(open(R + "/profiles/r01_pmc_derived.json"))
fg = d["k_trace<0>"]["fetch_GB"] * 1e9
t["fetch_size_kb_per_pass"] = round(fg / 2 / 1000, 1)
t["traffic_bytes_per_launch"] = int(round(fg / 9, -5))
json.dump(t, open(R + "/profiles/r01_traffic.json", "w"), indent=2)
b = json.load(open(R + "/profiles/r01_bench.json"))
for k in ("k_trace<0>", "k_trace<2>", "k_shade"):
    print(k, {f: round(d[k][f], 4) for f in ("ms_per_pass", "valu_busy", "waves_per_simd", "l1_hit", "l2_hit", "fetch_GB", "ta_busy")})
print("bench", round(b["value"], 1), round(b["ms_per_step"], 2), round(b["roofline"]["achieved"], 1), round(b["roofline"]["avg_launch_ms"], 3), b["roofline"]["kernel_ms"], b["cpu_baseline"]["value"])
print("traffic per launch", t["traffic_bytes_per_launch"])
PY
head -6 $S2 | cut -c1-110
