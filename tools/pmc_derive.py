#!/usr/bin/env python3
"""Derives the per-kernel figures of profiles/rNN_pmc_summary.md and the per-ray figures bench.py reads
(profiles/rNN_pmc_derived.json) from a pmc_summary.py text file (counters of ONE 1-spp pass), a kernel_stats.csv and
the bench line of the same workload.
Usage: pmc_derive.py <pmc_raw.txt> <kernel_stats_single_stream.csv> <passes in the stats run> <bench.json>"""
import csv, json, re, sys
from collections import defaultdict

def fold(name):
    """k_trace<MODE, K, SPILL> -> k_trace<MODE>, and the one-sun shadow form (MODE 3) under the general form's name (MODE 2)."""
    k = re.sub(r"<(\d), \d(?:, \w+)?>", r"<\1>", name.replace("void ", ""))
    k = re.sub(r"^k_shade<\d>$", "k_shade", k)          # waves per block
    return "k_trace<2>" if k == "k_trace<3>" else k

tot = defaultdict(float); disp = defaultdict(int)
for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+(\S+)\s+total\s+(\S+)\s+per_dispatch\s+\S+\s+dispatches\s+(\d+)", line)
    if m and not line.startswith("#"):
        k = fold(m.group(1).strip())     # fold the K variants of a mode
        tot[(k, m.group(2))] += float(m.group(3)); disp[(k, m.group(2))] += int(m.group(4))
ms = defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    k = fold(r["Name"].split("(")[0])
    ms[k] += float(r["TotalDurationNs"]) / 1e6 / int(sys.argv[3])
bench = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
out = {}
if bench:
    import re as _re
    wl = bench["config"]["workload"]
    m = _re.search(r"(\d+) tris, (\d+x\d+)", wl)
    mg = _re.search(r"max_generations (\d+)", wl); mc = _re.search(r"monte_carlo_max (\d+)", wl)
    scene = "atrium" if wl.startswith("procedural Sponza-class atrium") else wl.split(",")[0].strip()
    out["workload"] = {"scene": scene, "tris": int(m.group(1)), "res": m.group(2), "max_generations": int(mg.group(1)) if mg else 8,
                       "monte_carlo_max": int(mc.group(1)) if mc else 32}
    out["rays_per_pass"] = bench["config"]["rays_per_step"]
    out["shadow_rays_per_pass"] = bench["config"]["shadow_rays_per_step"]
for k in ("k_trace<0>", "k_trace<2>", "k_shade"):
    g = lambda c: tot[(k, c)]
    t = ms[k]; cyc = t * 1e-3 * 2.4e9
    if t <= 0:                      # the workload never launches this kernel (no sun: no shadow kernel)
        continue
    d = {"ms_per_pass": t, "launches": disp[(k, "SQ_WAVES")], "waves": g("SQ_WAVES"), "valu": g("SQ_INSTS_VALU"), "salu": g("SQ_INSTS_SALU"),
         "vmem_rd": g("SQ_INSTS_VMEM_RD"), "lds": g("SQ_INSTS_LDS"),
         "valu_busy": g("SQ_INSTS_VALU") / 1024 * 2 / cyc, "waves_per_simd": g("SQ_WAVE_CYCLES") * 4 / (cyc * 1024),
         "cycles_per_wave": g("SQ_WAVE_CYCLES") * 4 / max(1, g("SQ_WAVES")),
         "wait_inst_any": g("SQ_WAIT_INST_ANY") / max(1, g("SQ_WAVE_CYCLES")), "wait_any": g("SQ_WAIT_ANY") / max(1, g("SQ_WAVE_CYCLES")),
         "l1_hit": 1 - g("TCP_TCC_READ_REQ_sum") / max(1, g("TCP_TOTAL_CACHE_ACCESSES_sum")), "l2_hit": g("TCC_HIT_sum") / max(1, g("TCC_REQ_sum")),
         "l1_miss_latency_cycles": g("TCP_TCC_READ_REQ_LATENCY_sum") / max(1, g("TCP_TCC_READ_REQ_sum")),
         "fetch_GB": g("FETCH_SIZE") * 1000 * 2 / 1e9, "fetch_TBps": g("FETCH_SIZE") * 1000 * 2 / (t * 1e-3) / 1e12,
         "ta_busy": g("TA_TA_BUSY_sum") / 256 / cyc, "td_busy": g("TD_TD_BUSY_sum") / 256 / cyc,
         "td_tc_stall_of_busy": g("TD_TC_STALL_sum") / max(1, g("TD_TD_BUSY_sum")),
         "lds_bank_conflict": g("SQ_LDS_BANK_CONFLICT") / max(1, g("SQ_LDS_IDX_ACTIVE"))}
    d["write_GB"] = g("WRITE_SIZE") * 1000 / 1e9
    if bench:
        n = out["shadow_rays_per_pass"] if k == "k_trace<2>" else out["rays_per_pass"]
        # memory-side bytes: FETCH_SIZE is KB and counts 128-B requests at 64 B on gfx950 (x2); WRITE_SIZE is KB
        d["hbm_bytes_per_ray"] = (g("FETCH_SIZE") * 1000 * 2 + g("WRITE_SIZE") * 1000) / n
        d["l2_request_bytes_per_ray"] = g("TCP_TCC_READ_REQ_sum") * 128 / n
        d["l1_access_bytes_per_ray"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") * 128 / n
        d["valu_per_ray"] = g("SQ_INSTS_VALU") / n
    out[k] = d
print(json.dumps(out, indent=1))
