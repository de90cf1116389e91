#!/usr/bin/env python3
"""Headline benchmark of the trace-and-shade path: Mrays/s at 1080p on the Sponza-class scene.

A "step" is one sample pass (1 spp) of the hot path over the whole 1920x1080 frame of BASELINE.json's
configs[1] workload (procedural 262,144-triangle atrium, max_generations 8, monte_carlo_max 32): primary
rays -> closest hit -> shade/bounce -> compaction -> ... until every path has ended, then accumulation.
Rays = process_ray invocations (primary + every bounce generation), the reference's own count
(SceneManager.h:83-90); shadow rays are reported beside it. The scene, tables and framebuffers are resident
in HBM before the timed region starts.

Multi-GPU (one rank per GPU): the frame is sharded by tile rows dealt round-robin over the ranks, with the scene
replicated (strong scaling, total work fixed); no collective on the data path, one framebuffer all-gather over RCCL at
the end of the timed region. `python3 bench.py --gpus N` started plainly (no WORLD_SIZE in the environment) starts its
own N ranks through torch.distributed.run BEFORE anything touches the GPU and passes rank 0's line on; started by a
launcher it takes RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment. The reference needs no launcher
because its workers are threads of one process (SceneManager.h:392-404, :441-459); a rank per GPU is their counterpart.
A WORLD_SIZE that differs from --gpus is an error, never a silent one-rank run.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per unit for the closest-hit kernel, reference record sizes:
# 288 B per node test, 448 B per packet test, 64 B PackagedRay read + 32 B CastResults write per ray
NODE_B, PACKET_B, RAY_B, HIT_B = 288, 448, 64, 32
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=262144)
    ap.add_argument("--res", type=str, default="1920x1080")
    ap.add_argument("--ray-capacity", type=int, default=1 << 29, help="ray queue slots (132 B each: 71 GB of the 288 GB at 2^29, so a whole 1080p generation fits one launch even in the transmissive interior)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20241111)
    ap.add_argument("--scene", default="atrium", choices=["atrium", "interior", "diffuse_box", "instanced"],
                    help="atrium = the headline workload (configs[1]); interior = configs[2] (C3), diffuse_box = configs[4] (C5), "
                         "instanced = configs[3] (C4: --tris / 1000 instances of a 1 000-triangle mesh, flattened)")
    ap.add_argument("--mcm", type=int, default=0, metavar="N", help="monte_carlo_max override (Scene.h:29): C5 is run with 1 (its default here) and 32")
    ap.add_argument("--adaptive", type=int, default=0, metavar="MIN_SAMPLES",
                    help="not the headline run: after the timed steps, run the adaptive-sampling loop (controller's minimum_samples = "
                         "MIN_SAMPLES, analysis every 4 visits) and report it under config.adaptive")
    ap.add_argument("--fast-build", action="store_true", help="not the headline run: package the scene with srx_package_create_fast (a different tree, "
                    "tight leaves; hits equal the reference tree's through the order tables) and report the same metric over it")
    ap.add_argument("--no-fast-tree", action="store_true", help="skip the extra leg that repeats the measurement over the fast builder's tree (N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: exchange staged through the host)")
    ap.add_argument("--passes-per-call", type=int, default=0, metavar="B",
                    help="sample passes (steps) that share their kernel launches; 0 = the number of ranks, so a rank's launches stay as "
                         "large as the single GPU's (every step's work is still done inside the timed region)")
    ap.add_argument("--share-of", type=int, default=1, metavar="N",
                    help="rehearsal on one GPU: render only rank 0's share of an N-rank run (no exchange); value = this share's rays/s")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--print-launch-plan", action="store_true", help="print the command a plain --gpus N run would start for its ranks, and exit (CPU test)")
    return ap.parse_args()


# Ceilings of MI355X_MICROARCH.md: HBM3E 8 TB/s; rows gathered from a table that sits in the XCDs' L2s, 16.8-18.8 TB/s
# chip-wide (section "Indexed rows: gather into LDS", the lower figure is used)
L2_GATHER_PEAK_GBS = 16800.0


def pmc_profile(n_tris, res, scene, max_generations, monte_carlo_max):
    """Per-ray counter figures of the dominant kernel from the committed PMC profiles (rocprofv3 --pmc needs runs of its
    own, so they cannot be collected inside the timed run): profiles/rNN_pmc_derived*.json, made by tools/pmc_derive.py
    from the raw counter sums beside them, the newest round's file whose workload is this run's. (None, None) when no
    committed profile covers the workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_derived*.json")), reverse=True):
        try:
            with open(path) as fh:
                t = json.load(fh)
            w = t["workload"]
            if (w.get("scene", "atrium") == scene and w["tris"] == n_tris and w["res"] == res and w["max_generations"] == max_generations
                    and w.get("monte_carlo_max", 32) == monte_carlo_max):
                return t["k_trace<0>"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def host_cpu():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # on the GPU box the job's CPU share is what OMP_NUM_THREADS says (16 per GPU), not the machine's core count
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or usable
    return model, os.cpu_count() or 1, max(1, min(usable, share))


def cpu_baseline(desc, cam, seed, threads_all):
    """the CPU path as the reference runs it — node and packet tests as 8-wide AVX2 + FMA intrinsics
    (oracle/sro_trace_avx2.h, bit-identical to the scalar restatement: tests/test_oracle_cpu.py), reference RNG mode,
    std::thread workers over interleaved rows like assign_iterative_groups (SceneManager.h:441-459) — timed on bounded
    centre bands of the same frame: rows for 1 thread, 12 threads (the reference's own default, SceneManager.h:292) and
    all the cores this job may use, each the median of 3 runs after one warm-up. The only place bench.py touches oracle/."""
    from oracle import sro
    S = sro.OracleScene(desc, build_tables=True)
    S.set_simd(True)
    W, H = int(cam["res_x"]), int(cam["res_y"])
    acc = np.zeros((W * H, 4), np.float32)

    def timed(threads, rows):
        rows = max(4, min(H, rows // 4 * 4))
        y0 = (H - rows) // 2
        S.render_pass(cam, (0, y0, W, 4), 0, seed, acc, None, mode=1, threads=threads)          # warm-up
        rates, rays, secs = [], 0, []
        for rep in range(3):
            t0 = time.perf_counter()
            st = S.render_pass(cam, (0, y0, W, rows), 1 + rep, seed, acc, None, mode=1, threads=threads)
            dt = time.perf_counter() - t0
            rates.append(float(st[0]) / dt / 1e6); rays = int(st[0]); secs.append(dt)
        k = int(np.argsort(rates)[1])
        return {"threads": threads, "value": rates[k], "unit": "Mrays/s", "runs": [round(r, 4) for r in rates],
                "sample": f"centre {W}x{rows} band, 1 spp: {rays} rays in {secs[k]:.2f} s (median of 3)"}

    # band heights that keep every run near 2 s: ~0.2-0.4 Mrays/s per core, ~31 rays per pixel
    rows_1 = 8
    one = timed(1, rows_1)
    per_core = max(one["value"], 0.05)
    rows_for = lambda t: int(2.0 * per_core * t * 0.8 * 1e6 / (31.0 * W))
    out_rows = [one]
    if threads_all >= 12:
        out_rows.append(timed(12, rows_for(12)))
    allr = timed(threads_all, rows_for(threads_all)) if threads_all > 1 else one
    if threads_all > 1:
        out_rows.append(allr)
    model, visible, _ = host_cpu()
    return {
        "value": allr["value"], "unit": "Mrays/s", "cores": threads_all, "kind": "port",
        "simd": "avx2+fma intrinsics, 8-wide node and packet tests (oracle/sro_trace_avx2.h), g++ -O3 -mavx2 -mfma -ffp-contract=off",
        "cpu_model": model, "host_cpus_visible": visible,
        "sample": allr["sample"], "rows": out_rows,
    }


def fast_tree_leg(args, RE, desc, cam, rect, share, accum, aux, B, barrier):
    """Not the headline: the same passes over the tree of the opt-in fast builder (srx_package_create_fast: binned SAH,
    tight leaves; hits equal the reference tree's through the order tables except on exact ties and for rays with a zero
    direction component, tests/test_fast_builder_gpu.py). `value` stays the bit-compatible tree's."""
    from raytracing_amd import engine
    t0 = time.perf_counter()
    PSf = engine.package(desc, fast=True)
    t_pack = time.perf_counter() - t0
    RE.load_scene(PSf)
    RE.set_timing(True, ray_capacity=args.ray_capacity)
    steps = max(1, min(args.steps, 3))
    RE.trace_pass(cam, rect, 0, args.seed, accum, aux, ray_capacity=args.ray_capacity, interleave=share, n_passes=1)      # warm-up
    barrier()
    t0 = time.perf_counter()
    tot = {}
    for p in range(1, 1 + steps, B):
        c = RE.trace_pass(cam, rect, p, args.seed, accum, aux, ray_capacity=args.ray_capacity, interleave=share, n_passes=min(B, 1 + steps - p))
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
    barrier()
    dt = time.perf_counter() - t0
    rays = float(tot["rays_processed"])
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "steps": steps, "ms_per_step": dt / steps * 1e3, "scene_package_s": t_pack,
            "builder": "srx_package_create_fast (opt-in; a different tree behind the order tables)",
            "node_tests_per_ray": tot["nodes"] / max(1.0, rays), "packet_tests_per_ray": tot["packets"] / max(1.0, rays),
            "stack_depth": RE.stack_bound()}


def adaptive_loop(args, RE, cam, rect, share, accum, aux, W, H, world, rank, grp, barrier):
    """SceneManager's mode-1 loop (SceneManager.h:970-993) on the GPU: visits of every 4x4 block with the
    controller's priorities, analyze_noise + balance_threads every 4 visits (the reference: every 2 s). With several
    ranks the second half of balance_threads (:693-773) runs too: after every analysis the tile rows are re-dealt by
    their burden (tiles.Deal, the same deal on every rank from the same priorities; the framebuffers are gathered under
    the old deal first, because a rank that takes rows over continues their running means)."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import engine, tiles
    NC = engine.NoiseController(W, H, seed=args.seed, minimum_samples=args.adaptive)
    err = torch.zeros(W * H, dtype=torch.float32, device="cuda")
    prio = torch.ones(NC.n_blocks, dtype=torch.float32, device="cuda")
    deal = tiles.Deal(W, H, world) if world > 1 and args.share_of <= 1 else None
    visit = args.warmup + args.steps
    log, analyses_ms, visits_ms, rays, moved_total = [], 0.0, 0.0, 0, 0
    while visit < args.warmup + args.steps + 4 * 12 + args.adaptive:
        barrier()
        t0 = time.perf_counter()
        for _ in range(4):
            c = RE.trace_visit(cam, rect, visit, args.seed, accum, aux, block_priority=prio, ray_capacity=args.ray_capacity,
                               interleave=share, deal=deal, part=rank)
            rays += c["rays_processed"]
            visit += 1
        barrier()
        t1 = time.perf_counter()
        if deal is None:
            seen = tiles.allreduce_error_map(err, NC.error_map(accum, aux, err, interleave=share), H, W)
        else:
            seen = NC.error_map(accum, aux, err, deal=deal, part=rank)
            if grp is not None:
                grp.allgather_rows_dealt(deal, err, 1)
                mn = torch.tensor([int(seen)], dtype=torch.int32, device="cuda")
                grp.allreduce(None, mn)
                seen = int(mn.item())
            else:
                tiles.gather_framebuffer_dealt(err.view(H * W, 1), deal)
                mn = torch.tensor([int(seen)], dtype=torch.int64)
                dist.all_reduce(mn, op=dist.ReduceOp.MIN)
                seen = int(mn.item())
        NC.analyze_noise(err, seen)
        updated = NC.balance(seen, prio)
        moved = 0
        if deal is not None and updated:
            if grp is not None:
                grp.gather_framebuffer_dealt(deal, accum, aux)
            else:
                tiles.gather_framebuffer_dealt([accum, aux], deal)
            moved = deal.rebalance(prio)
            moved_total += moved
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        visits_ms += (t1 - t0) * 1e3
        analyses_ms += (t2 - t1) * 1e3
        R = NC.regions()
        lum = tiles.luminance(accum[:, :3])
        log.append({"negative_luminance_pixels": int((lum < 0).sum()), "nan_error_pixels": int(torch.isnan(err).sum()),"visit": visit, "min_samples": seen, "regions": int(len(R)), "active_regions": int(R["active"].sum()),
                    "mean_priority": float(prio.mean()), "visits_ms": (t1 - t0) * 1e3, "analysis_ms": (t2 - t1) * 1e3, "tile_rows_moved": moved})
    NC.close()
    out = {"minimum_samples": args.adaptive, "visits_ms": visits_ms, "analyses_ms": analyses_ms, "rays_this_rank": rays,
           "rounds": log[:2] + log[-3:]}
    if deal is not None:
        _, _, burdens = deal.tables()
        out["deal"] = {"tile_rows_moved": moved_total, "burdens": [float(b) for b in burdens], "rows_of_this_rank": int(len(deal.rows_of(rank)))}
    return out


def launch_plan(argv, gpus, port):
    """The command a plain `python3 bench.py --gpus N ...` runs instead of itself: N ranks of this file on this node
    (what the reference's spawn_threads / assign_iterative_groups do with threads, SceneManager.h:392-404, :441-459)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """--gpus N > 1 without a launcher's environment: start the N ranks as a child job (nothing in this process has
    touched the GPU or imported torch yet) and exit with its status. The children see WORLD_SIZE and take the other path."""
    import subprocess
    cmd = launch_plan(argv, args.gpus, free_port())
    if args.print_launch_plan:
        print(json.dumps({"launch_plan": cmd}), flush=True)
        return 0
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if args.print_launch_plan:
        print(json.dumps({"launch_plan": None}), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # both directions: a launcher with more ranks than --gpus, and --gpus N inside a one-rank environment
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to measure a job of a different size")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible and there is no CPU path")
    if args.one_device:
        if args.backend == "nccl" and world > 1:
            raise SystemExit("bench.py: --one-device puts every rank on cuda:0, which RCCL cannot do: use --backend gloo")
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but {torch.cuda.device_count()} device(s) are visible "
                         f"(--gpus {args.gpus} needs one GPU per rank; --one-device --backend gloo rehearses on one)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from raytracing_amd import engine, scenes

    W, H = (int(v) for v in args.res.split("x"))
    if args.scene == "atrium":
        desc = scenes.atrium(args.tris, res=(W, H), subdiv=8, max_generations=8)
    elif args.scene == "interior":
        desc = scenes.interior(args.tris, res=(W, H), subdiv=8)
    elif args.scene == "instanced":
        desc = scenes.instanced(max(1, args.tris // 1000), 1000, res=(W, H), subdiv=8)
    else:
        desc = scenes.diffuse_box(args.tris, res=(W, H), subdiv=8)
    if args.mcm > 0:
        desc.monte_carlo_max = args.mcm
    cam = desc.camera
    t0 = time.perf_counter()
    PS = engine.package(desc, fast=args.fast_build)
    t_pack = time.perf_counter() - t0
    RE = engine.RayEngine(local_rank)
    RE.load_scene(PS)
    RE.set_timing(True, ray_capacity=args.ray_capacity)

    # this rank's share of the frame: tile rows (4 pixel rows) dealt round-robin over the ranks
    from raytracing_amd import tiles
    rect = (0, 0, W, H)
    share = (tiles.TILE, rank, world) if args.share_of <= 1 else (tiles.TILE, 0, args.share_of)
    accum = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
    aux = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # steps = sample passes; B of them travel through the same kernel launches (same samples as one call per pass)
    B = args.passes_per_call if args.passes_per_call > 0 else max(world, args.share_of)
    B = max(1, min(B, ((1 << 24) - 1) // (W * H)))
    # size the renderer's per-pass sample buffers for B passes before anything is timed (4 rows into scratch framebuffers)
    scratch = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
    RE.trace_pass(cam, (0, 0, W, min(H, tiles.TILE)), 0, args.seed, scratch, scratch.clone(), ray_capacity=args.ray_capacity, n_passes=B)
    del scratch
    for p in range(0, args.warmup, B):
        RE.trace_pass(cam, rect, p, args.seed, accum, aux, ray_capacity=args.ray_capacity, interleave=share, n_passes=min(B, args.warmup - p))
    # The two exchange steps of a multi-GPU frame, inside the timed region: the adaptive-sampling controller's payload
    # (every rank's rows of the per-pixel error map, all-gathered, + the minimum sample count) and the framebuffer
    # all-gather (main + aux in one collective). Through the library's own srx_group (RCCL loaded by libsrx.so) when the
    # ranks sit on GPUs of their own; over torch.distributed otherwise (gloo rehearsals) or if the group cannot be made.
    grp, exchange_path, NCx, err, nranks_seen = None, "none (one rank)", None, None, 1
    if world > 1:
        NCx = engine.NoiseController(W, H, seed=args.seed, device=local_rank)
        err = torch.zeros(W * H, dtype=torch.float32, device="cuda")
        if args.backend == "nccl":
            # the C-ABI path or nothing: a failure to make the group fails the run (no measurement on another path)
            grp = tiles.Group.from_torch_dist()
            exchange_path = "srx_group (C-ABI, RCCL)"
            nranks_seen = grp.nranks_seen()
            if nranks_seen != world:
                raise SystemExit(f"bench.py: srx_group reports {nranks_seen} ranks, the job has {world}")
        else:
            exchange_path = "torch.distributed (" + args.backend + "), rows packed with torch index ops (rehearsal, not RCCL)"
            nranks_seen = dist.get_world_size()

    def exchange():
        if world <= 1:
            return
        seen = NCx.error_map(accum, aux, err, interleave=share)
        if grp is not None:
            grp.complete_error_map(err, seen, H, W)
            grp.gather_framebuffer(accum, aux, H, W)
        else:
            tiles.allreduce_error_map(err, seen, H, W)
            tiles.gather_framebuffer_interleaved([accum, aux], H, W)

    exchange()      # warm-up of the exchanges as well (communicator set-up); they only fill in the other ranks' rows
    barrier()
    t0 = time.perf_counter()
    tot = {}
    for p in range(args.warmup, args.warmup + args.steps, B):
        c = RE.trace_pass(cam, rect, p, args.seed, accum, aux, ray_capacity=args.ray_capacity, interleave=share,
                          n_passes=min(B, args.warmup + args.steps - p))
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
    exchange()
    barrier()
    dt = time.perf_counter() - t0

    adaptive = None
    if args.adaptive > 0:
        adaptive = adaptive_loop(args, RE, cam, rect, share, accum, aux, W, H, world, rank, grp, barrier)

    keys = ["rays_processed", "shadows_cast", "nodes", "packets", "shadow_nodes", "shadow_packets", "hits",
            "trace_launches", "shadow_launches", "shade_launches"]
    vals = torch.tensor([float(tot[k]) for k in keys] + [tot["trace_ms"], tot["shadow_ms"], tot["shade_ms"], dt],
                        dtype=torch.float64, device="cuda")
    if world > 1:
        if args.backend != "nccl":
            vals = vals.cpu()
        mx = vals.clone()
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dt = float(mx[-1])
    agg = {k: float(vals[i]) for i, k in enumerate(keys)}
    trace_ms, shadow_ms, shade_ms = (float(vals[len(keys) + i]) for i in range(3))

    if rank == 0:
        rays = agg["rays_processed"]
        value = rays / dt / 1e6
        # roofline of the dominant kernel (closest-hit traversal), algorithmic bytes over its own HIP-event time
        alg_bytes = NODE_B * agg["nodes"] + PACKET_B * agg["packets"] + (RAY_B + HIT_B) * rays
        launches = max(1.0, agg["trace_launches"])
        alg_gbs = alg_bytes / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        avg_ms = trace_ms / launches
        rays_per_launch = rays / launches
        # Memory-side traffic and cache-path figures per ray come from the committed PMC profile of this workload (kernels
        # alone on one stream); times are this run's HIP events. Every frac below is a fraction of a ceiling that can bind.
        # per-ray figures of the 1-GPU profile of this workload; a rank's share has the same rays per pixel
        prof, prof_path = pmc_profile(len(desc.tris), args.res, desc.name, desc.max_generations, desc.monte_carlo_max)
        traffic = prof["hbm_bytes_per_ray"] * rays_per_launch if prof else None
        hbm_gbs = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
        l2_bytes = prof["l2_request_bytes_per_ray"] * rays_per_launch if prof else None
        l2_gbs = l2_bytes / (avg_ms * 1e-3) / 1e9 if l2_bytes else None
        roofline = {
            "kernel": "k_trace<0> (closest-hit traversal)",
            "bound": "hbm",
            "achieved": hbm_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": hbm_gbs / HBM_PEAK_GBS if hbm_gbs else None,
            "traffic": traffic,
            "traffic_source": (f"{prof_path} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this workload, per ray) x rays per launch"
                               if prof_path else None),
            "launches": int(launches),
            "avg_launch_ms": avg_ms,
            "algorithmic": {
                "what": "SURVEY 8(d): 288 B x node tests + 448 B x packet tests + 96 B per ray, tests counted by the kernel",
                "bytes_per_launch": alg_bytes / launches,
                "bytes_per_ray": alg_bytes / max(1.0, rays),
                "GBps": alg_gbs,
                "over_traffic": (alg_bytes / launches) / traffic if traffic else None,
                "note": "not a roofline fraction: the tree (nodes + packets) is served by L1 / L2 / Infinity Cache, so the "
                        "algorithmic bytes exceed what reaches the memory side; the ratio is cache reuse, not waste",
            },
            "l2_gather": {
                "what": "bytes requested from the L2s (TCP_TCC_READ_REQ x 128 B) against the guide's ceiling for rows gathered out of L2",
                "achieved": l2_gbs, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s",
                "frac": l2_gbs / L2_GATHER_PEAK_GBS if l2_gbs else None,
            },
            "units_busy": ({k: prof[k] for k in ("valu_busy", "ta_busy", "td_busy", "td_tc_stall_of_busy", "waves_per_simd", "wait_any",
                                                 "l1_hit", "l2_hit") if k in prof} if prof else None),
            "limiter": (f"latency of a trip (pop -> dependent gather -> ~148 VALU + 36 SALU for a wave-trip with both kinds of octet -> push) "
                        f"x {(agg['nodes'] + agg['packets']) / max(1.0, rays):.1f} trips per ray at 8 waves per SIMD; no throughput unit is saturated "
                        "(DESIGN.md section 6, profiles/r03_trace_isa.md)"),
            "concurrency": ("shadow kernels run on a second stream beside this kernel: avg_launch_ms includes the time they share the chip")
                           if os.environ.get("SRX_SHADOW_STREAM", "1") != "0" else "one stream",
            # the same with the bytes a test actually uses of a record (264 of the 288-B node, 324 of the
            # 448-B packet), and what the kernel's own layout fetches per test (2 x 128-B node lines, 3 packet lines)
            "useful_bytes_per_ray": (264 * agg["nodes"] + 324 * agg["packets"]) / max(1.0, rays) + RAY_B + HIT_B,
            "fetched_bytes_per_ray": (256 * agg["nodes"] + 384 * agg["packets"]) / max(1.0, rays) + 24 + 20,
            "node_tests_per_ray": agg["nodes"] / max(1.0, rays),
            "packet_tests_per_ray": agg["packets"] / max(1.0, rays),
            "shadow_node_tests_per_shadow_ray": agg["shadow_nodes"] / max(1.0, agg["shadows_cast"]),
            "shadow_packet_tests_per_shadow_ray": agg["shadow_packets"] / max(1.0, agg["shadows_cast"]),
            "kernel_ms": {"trace": trace_ms, "shadow": shadow_ms, "shade": shade_ms, "wall": dt * 1e3},
        }
        out = {
            "metric": "Mrays/s (primary + diffuse bounce) at 1080p",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"procedural Sponza-class atrium, {args.tris} tris, {W}x{H}, 1 spp per step, max_generations 8, "
                             f"monte_carlo_max 32 ({'BASELINE.json configs[1]' if (args.tris, args.res) == (262144, '1920x1080') else 'configs[1] scaled'})")
                            if args.scene == "atrium" else
                            f"{desc.name}, {len(desc.tris)} tris, {W}x{H}, 1 spp per step, max_generations {desc.max_generations}, "
                            f"monte_carlo_max {desc.monte_carlo_max} (not the headline workload)",
                "parallelism": f"interleaved tile rows x{world}, scene replicated",
                "exchange": exchange_path,
                "nranks_seen": nranks_seen,
                "launcher": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or "none (one process)",
                "rays_per_step": rays / args.steps,
                "shadow_rays_per_step": agg["shadows_cast"] / args.steps,
                "ray_capacity": args.ray_capacity,
                "scene_package_s": t_pack,
                "builder": "srx_package_create_fast (opt-in: a different tree, not the reference's triangle numbering)" if args.fast_build
                           else "srx_package_create (bit-compatible with Scene::package)",
                "stack_depth": RE.stack_bound(),
                "passes_per_call": B,
            },
            "roofline": roofline,
        }
        if adaptive is not None:
            out["config"]["adaptive"] = adaptive
        if world == 1 and not args.fast_build and not args.no_fast_tree and args.share_of <= 1:
            out["fast_tree"] = fast_tree_leg(args, RE, desc, cam, rect, share, accum, aux, B, barrier)
        if not args.no_cpu_baseline and world == 1:
            threads = args.cpu_threads or host_cpu()[2]
            out["cpu_baseline"] = cpu_baseline(desc, cam, args.seed, threads)
        print(json.dumps(out), flush=True)
    RE.close()
    if world > 1:
        if NCx is not None:
            NCx.close()
        if grp is not None:
            grp.close()             # the library's communicator goes before torch's, and not at interpreter shutdown
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
