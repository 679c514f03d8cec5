#!/usr/bin/env python3
"""bench.py -- Mpts/s of Pipeline.ingest -> finalize on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A step = one ingest + finalize of one synthetic cloud that is already resident in HBM, on a
fresh pre-created pipeline (the reference's protocol creates the pipeline before the clock,
scripts/benchmarks/benchmark_glyph_full.py:80-97).  Finalized bands stay in HBM
(PipelineConfig.result_location = Device); the PCIe-inclusive rates are in DESIGN.md.

Default workload = BASELINE.json configs[1] ("C2"): 50 M uniform points, 4096 x 4096 grid,
Point glyph, Sum + Count + Average on one channel.  With N GPUs the grid is row-block sharded:
4096 x (4096*N) cells, 50 M points per GPU generated inside that GPU's block (weak scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch                      # first: the engine must share torch's HIP runtime
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
import pcr                        # noqa: E402
from pcr.distributed import ShardedPipeline   # noqa: E402

HBM_PEAK_GBS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (description, glyph, reductions, bytes per point of compulsory input traffic)
    "C2": ("50M uniform pts, 4096^2, Point, Sum+Count+Average", "point", ("Sum", "Count", "Average"), 20),
    "point_avg": ("Point, Average", "point", ("Average",), 20),
    "C4": ("clustered (10k hotspots), Point, Max+Min", "point", ("Max", "Min"), 20),
    "gauss1": ("Gaussian sigma=1 r<=4, WeightedAverage", "gauss", 1.0, 20),
    "gauss1.8": ("Gaussian sigma=1.8 r<=6, WeightedAverage", "gauss", 1.8, 20),
    "gauss2": ("Gaussian sigma=2 r<=6, WeightedAverage", "gauss", 2.0, 20),
    "gauss4": ("Gaussian sigma=4 r<=12 (C3), WeightedAverage", "gauss", 4.0, 20),
    "gauss16": ("Gaussian sigma=16 r<=48, WeightedAverage", "gauss", 16.0, 20),
    "line16": ("Line hl=16 per-point direction (C3), WeightedAverage", "line", 16.0, 24),
}


def make_specs(workload):
    _, glyph, arg, _ = WORKLOADS[workload]
    specs = []
    if glyph == "point":
        for name in arg:
            r = pcr.ReductionSpec()
            r.value_channel = "value"
            r.type = getattr(pcr.ReductionType, name)
            specs.append(r)
    elif glyph == "gauss":
        max_r = 12.0 if arg == 4.0 else min(4.0 * arg, 64.0)
        specs.append(pcr.gaussian_splat_spec("value", default_sigma=arg, max_radius_cells=max_r))
    else:
        specs.append(pcr.line_splat_spec("value", direction_channel="direction",
                                         default_half_length=arg, max_radius_cells=arg + 2.0))
    return specs


def make_points(workload, n, G, y_lo, y_hi, seed):
    """x ~ U(2, G-2); y ~ U(y_lo+2, y_hi-2) (world units, cell size 1); value ~ U(0,1)."""
    rng = np.random.default_rng(seed)
    if workload == "C4":
        k = 10_000
        cx = rng.uniform(2, G - 2, k)
        cy = rng.uniform(y_lo + 2, y_hi - 2, k)
        idx = np.arange(n) % k
        x = np.clip(cx[idx] + rng.normal(0, 2.0, n), 0, G)
        y = np.clip(cy[idx] + rng.normal(0, 2.0, n), y_lo, y_hi)
    else:
        x = rng.uniform(2, G - 2, n)
        y = rng.uniform(y_lo + 2, y_hi - 2, n)
    v = rng.uniform(0, 1, n).astype(np.float32)
    ch = {}
    if WORKLOADS[workload][1] == "line":
        ch["direction"] = rng.uniform(0, np.pi, n).astype(np.float32)
    return x, y, v, ch


def make_cloud(x, y, v, ch):
    c = pcr.PointCloud.create(len(x))
    c.set_x_array(x)
    c.set_y_array(y)
    c.add_channel("value", pcr.DataType.Float32)
    c.set_channel_array_f32("value", v)
    for name, arr in ch.items():
        c.add_channel(name, pcr.DataType.Float32)
        c.set_channel_array_f32(name, arr)
    return c


def cpu_baseline(workload, G, sample_pts, seed):
    """The CPU oracle (a single-threaded port of the reference's algorithm, without its sort)
    timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcr_oracle_py as O
    _, glyph, arg, _ = WORKLOADS[workload]
    x, y, v, ch = make_points(workload, sample_pts, G, 0.0, float(G), seed)
    og = O.make_grid((0.0, 0.0, float(G), float(G)))
    t0 = time.perf_counter()
    if glyph == "point":
        for name in arg:                                  # the reference runs one pass per ReductionSpec
            O.run(og, {"Sum": O.SUM, "Count": O.COUNT, "Average": O.AVERAGE, "Max": O.MAX, "Min": O.MIN}[name], x, y, v)
    elif glyph == "gauss":
        max_r = 12.0 if arg == 4.0 else min(4.0 * arg, 64.0)
        O.run(og, O.WEIGHTED_AVERAGE, x, y, v,
              glyph=O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=arg, sigma_y=arg, max_radius=max_r))
    else:
        O.run(og, O.WEIGHTED_AVERAGE, x, y, v,
              glyph=O.make_glyph(O.GLYPH_LINE, half_length=arg, max_radius=arg + 2.0), **ch)
    dt = time.perf_counter() - t0
    return {"value": round(sample_pts / dt / 1e6, 4), "unit": "Mpts/s", "cores": 1, "kind": "port",
            "sample": f"{sample_pts} pts of the same workload on the {G}^2 grid, ingest+finalize, "
                      f"{dt:.1f} s, single thread, no sort (oracle/pcr_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--points", type=int, default=50_000_000, help="points per GPU")
    ap.add_argument("--grid", type=int, default=4096, help="grid width = rows per GPU")
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU when not square (C5 shard: --grid 16384 --rows 2048)")
    ap.add_argument("--path", default="auto", choices=["auto", "direct", "binned", "moments"])
    ap.add_argument("--cpu-sample", type=int, default=-1, help="points of the CPU baseline sample (0 = skip)")
    ap.add_argument("--host-result", action="store_true", help="finalize into host memory (PCIe-inclusive)")
    ap.add_argument("--host-cloud", action="store_true", help="ingest a host-resident cloud (PCIe-inclusive)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N > 1 code path on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        # communicator set-up is not part of any step
        warm = torch.zeros(1, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(warm)
        dist.barrier()

    G, n = args.grid, args.points
    R = args.rows if args.rows > 0 else G                  # rows per GPU
    H = R * world                                          # one R-row block per GPU
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(H))
    cfg.grid.cell_size_x, cfg.grid.cell_size_y = 1.0, -1.0
    cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU
    cfg.cuda_device_id = local_rank
    cfg.reductions = make_specs(args.workload)
    cfg.result_location = pcr.MemoryLocation.Host if args.host_result else pcr.MemoryLocation.Device
    cfg.scatter_path = {"auto": 0, "direct": 1, "binned": 2, "moments": 3}[args.path]
    # scratch arena (routing keys + records) sized at create, outside the clock, as the reference sizes
    # its MemoryPool in Pipeline::create (src/engine/pipeline.cpp:167-184)
    cfg.gpu_pool_size_bytes = 16 * n + (64 << 20)

    # this rank's rows [rank*G, (rank+1)*G) <=> world y in (H - (rank+1)*G, H - rank*G)
    y_hi = float(H - rank * R)
    y_lo = y_hi - R
    x, y, v, ch = make_points(args.workload, n, G, y_lo, y_hi, seed=42 + rank)
    cloud = make_cloud(x, y, v, ch)
    if not args.host_cloud:
        cloud = cloud.to_device()
    del x, y, v, ch

    total = args.warmup + args.steps
    pipes = [ShardedPipeline(cfg, rank, world, device_id=local_rank) for _ in range(total)]

    def step(sp):
        sp.ingest(cloud)
        sp.finalize()

    for sp in pipes[:args.warmup]:
        step(sp)
    for sp in pipes[args.warmup:]:
        sp.pipe.profile_enable(True)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for sp in pipes[args.warmup:]:
        step(sp)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel HIP-event times of the timed steps (this rank)
    kernels = {}
    for sp in pipes[args.warmup:]:
        for name, (launches, ms) in sp.pipe.profile_read(True).items():
            k = kernels.setdefault(name, [0, 0.0])
            k[0] += launches
            k[1] += ms
    info = pipes[-1].pipe.last_scatter()

    if rank == 0:
        desc, glyph, _, bpp = WORKLOADS[args.workload]
        ms_per_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed / 1e6
        out = {
            "metric": "Mpts/s ingest->finalize",
            "value": round(value, 2), "unit": "Mpts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "points_per_gpu": n,
                       "grid": f"{G}x{H}", "rows_per_gpu": R, "glyph": glyph,
                       "reductions": [str(r.type).split(".")[-1] for r in cfg.reductions],
                       "scatter_path": info["path"], "lds_tile": list(info["lds_tile"]),
                       "num_bins": info["num_bins"],
                       "input": "host-resident (H2D inside step)" if args.host_cloud else "device-resident",
                       "result": "host (D2H inside step)" if args.host_result else "device-resident",
                       "parallelism": f"row-block x{world}"},
        }
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k][1])
            launches, ms = kernels[dom]
            avg_ms = ms / launches
            # points one launch of that kernel processes: all of the ingest, except that the binning scatter is two
            # launches (whole chunks under "k_bin_scatter", the ragged last chunk under "k_bin_scatter_tail")
            n_launch = n
            if dom == "k_bin_scatter":
                chunk = 16384 if info["num_bins"] <= 2048 else 8192
                n_launch = (n // chunk) * chunk
            achieved = bpp * n_launch / (avg_ms * 1e-3) / 1e9
            traffic = None        # HBM bytes per launch of the dominant kernel, from committed rocprofv3 PMC runs
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f).get(args.workload, {})
                if dom in pmc and pmc.get("points_per_launch") == n:
                    traffic = pmc[dom]["read_bytes"] + pmc[dom]["write_bytes"]
            except OSError:
                pass
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                               "traffic": traffic, "avg_kernel_ms": round(avg_ms, 4),
                               "algorithmic_bytes_per_launch": bpp * n_launch}
            out["kernels_ms_per_step"] = {k: round(v[1] / args.steps, 4) for k, v in sorted(kernels.items())}
        sample = args.cpu_sample
        if sample < 0:
            sample = {"point": 50_000_000, "gauss": 0, "line": 8_000_000}[glyph]          # ~10 s of single-thread CPU work
            if glyph == "gauss":
                sigma = WORKLOADS[args.workload][2]
                sample = int(2e9 / (2 * min(3 * sigma, 64) + 1) ** 2 / 4)     # ~10-20 s of cell updates
        if world == 1 and sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.workload, G, min(sample, n), seed=42)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
