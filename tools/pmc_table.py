#!/usr/bin/env python3
"""profiles/rNN_pmc_summary.md from the per-workload profiles/rNN_pmc_derived*.json (tools/pmc_derive.py): one row per
workload and kernel. Usage: python tools/pmc_table.py r03 > profiles/r03_pmc_summary.md   (CPU only)"""
import glob, json, os, sys

tag = sys.argv[1]
names = {"k_trace<0>": "closest hit `k_trace<0>`", "k_trace<2>": "shadow `k_trace<2/3>`", "k_shade": "shade `k_shade`"}
print(f"# {tag}: counters of the three hot kernels on every BASELINE.json workload\n")
print("One kernel at a time on the chip (`SRX_SHADOW_STREAM=0`), one 1-spp 1080p pass; `rocprofv3 --pmc`, one run per counter group\n"
      f"(`tools/pmc_passes.sh`, `tools/pmc_passes_mem.sh` -> `profiles/{tag}_pmc_raw*.txt`), times from `rocprofv3 --kernel-trace --stats`\n"
      f"of the same command (`profiles/{tag}_kernel_stats_single_stream*.csv`); derived by `tools/pmc_derive.py`, tabulated by `tools/pmc_table.py`.\n"
      "Memory side = (`FETCH_SIZE` x 2 + `WRITE_SIZE`) past the L2s (the guide's gfx950 correction), per ray of the kernel's own kind and\n"
      "as a rate while the kernel runs; L2 requests = `TCP_TCC_READ_REQ` x 128 B.\n")
print("| workload | kernel | ms per pass | VALU per ray | VALU busy | waves per SIMD | waiting | L1 hit | L2 hit | L1-miss latency (cycles) | TA busy | TD busy (stalled on L2) | LDS bank conflicts | memory side B per ray | TB/s (of 8) | L2 requests B per ray |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
order = ["", "_interior", "_instanced10M", "_10Mtris", "_diffuse_box", "_diffuse_box_mcm32"]
for suf in order:
    f = f"profiles/{tag}_pmc_derived{suf}.json"
    if not os.path.exists(f):
        continue
    d = json.load(open(f)); w = d["workload"]
    label = f"{w['scene']}, {w['tris']:,} tris, gen {w['max_generations']}, mcm {w['monte_carlo_max']}".replace(",", " ", 1) if False else \
            f"{w['scene']} {w['tris']} tris, {w['max_generations']} generations, `monte_carlo_max` {w['monte_carlo_max']}"
    for k in ("k_trace<0>", "k_trace<2>", "k_shade"):
        if k not in d:
            continue
        x = d[k]; n = d["shadow_rays_per_pass"] if k == "k_trace<2>" else d["rays_per_pass"]
        tbs = x["hbm_bytes_per_ray"] * n / (x["ms_per_pass"] * 1e-3) / 1e12
        print(f"| {label} | {names[k]} | {x['ms_per_pass']:.2f} | {x['valu_per_ray']:.0f} | {x['valu_busy']:.0%} | {x['waves_per_simd']:.1f} | {x['wait_any']:.0%} | "
              f"{x['l1_hit']:.0%} | {x['l2_hit']:.0%} | {x['l1_miss_latency_cycles']:.0f} | {x['ta_busy']:.0%} | {x['td_busy']:.0%} ({x['td_tc_stall_of_busy']:.0%}) | "
              f"{x['lds_bank_conflict']:.0%} | {x['hbm_bytes_per_ray']:.0f} | {tbs:.2f} ({tbs / 8:.2f}) | {x['l2_request_bytes_per_ray']:.0f} |")
        label = ""
print("\nReading it: DESIGN.md section 6 (\"What binds\") and section 5 (the shade kernel's rounds and hit list; `profiles/r03_trace_isa.md` for the\n"
      "closest-hit loop's instruction budget).")
