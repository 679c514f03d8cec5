#!/usr/bin/env python3
"""bench.py -- Mpts/s of Pipeline.ingest -> finalize on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 runs either way: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (WORLD_SIZE set:
this process is a rank), or as plain `python bench.py --gpus N ...` -- then this process only starts that launcher as a
child (before torch or the GPU are touched), lets rank 0's JSON line through and exits with the child's code.

A step = one ingest + finalize of one synthetic cloud that is already resident in HBM, on a fresh
pre-created pipeline (the reference's protocol creates the pipeline before the clock,
scripts/benchmarks/benchmark_glyph_full.py:80-97).  Finalized bands stay in HBM
(PipelineConfig.result_location = Device).  Rank 0 prints ONE JSON line.

N = 1 (default): BASELINE.json configs[1] ("C2"): 50 M uniform points, 4096 x 4096 grid, Point glyph,
Sum + Count + Average on one channel -- the headline `value`.  The same run also reports
  per_glyph     Point / Average alone (and with TWO ingests into one pipeline per step: the second ingest read-modify-writes
                the state planes, the first stores into fresh ones), clustered = configs[3] (10 000 hotspots, Max + Min),
                Line hl=16 / Gaussian sigma = 1, 4, 16 on the same 50 M points and grid (configs[2] settings;
                protocol scripts/benchmarks/benchmark_glyph_full.py:92-97,119-133)
  e2e_host      the drop-in default: host-resident cloud in, host-resident result out (PCIe inside the step)
  cpu_baseline  the reference's CPU stages (oracle/pcr_cpu_pipeline.cpp) on this box's host cores,
                1 thread and all cores, on a bounded sample.

N > 1 (default): BASELINE.json configs[4] ("C5"), STRONG scaling: one fixed 16384 x 16384 grid, 1 B uniform
points in total, row blocks from pcr.distributed.row_block (2048 rows at N = 8: the blocks cut the 4096-row
reference tiles, so the touched-tile union and -- for Gaussians -- the halo rows really cross RCCL).  Every
rank generates, on its device, the 1e9/N points of the cloud that fall in its block (seed 42 + rank): points
arrive routed by y.  `value` = Point / Average; `per_glyph.gauss1` = Gaussian sigma = 1 (r <= 4) / Average in
the same run; `one_gpu_same_problem` = the same 1 B points on rank 0's GPU alone (the other ranks wait), so that the line
carries its own denominator (`speedup_vs_one_gpu`).  `selfcheck` (untimed): points valid over all ranks == points
generated; Gaussian plane sums over all ranks' state rows before the exchange == over the owned rows after it; the
touched-tile union agrees on every rank -- the first run on a real transport validates its own exchange.  --unrouted adds `unrouted`: the same Point step when each rank is
instead handed an ARBITRARY 1/N of the cloud and the step includes the device-side partition + all-to-all
(pcr.distributed.route_cloud).
--weak keeps round 1's shape (4096 x 4096*N grid, 50 M points per GPU, tile-aligned blocks, no collective).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
# torch / pcr are imported by _imports(), AFTER main() has decided whether this process is a rank or only the launcher of
# the ranks: the launcher must not touch the GPU (it starts `python -m torch.distributed.run` as a child and waits).
torch = dist = pcr = ShardedPipeline = row_block = None


def _imports():
    global torch, dist, pcr, ShardedPipeline, row_block
    import torch as _torch
    import torch.distributed as _dist
    import pcr as _pcr
    from pcr.distributed import ShardedPipeline as _SP, row_block as _rb
    torch, dist, pcr, ShardedPipeline, row_block = _torch, _dist, _pcr, _SP, _rb


def self_launch(n, limit_s=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks with torch.distributed.run as a
    CHILD process in its own process group (never exec: this process has not touched the GPU and stays that way), let rank
    0's JSON line through on the inherited stdout, and return the child's exit code.  The rendezvous is torchrun's own
    (`--standalone`: it binds its store to a free port itself -- no pick-then-use race on a busy box).  A run that exceeds
    the wall-clock limit (PCR_BENCH_LAUNCH_TIMEOUT seconds, default 1500) has its whole process group killed and the
    launcher exits with 124: a hung collective ends the run instead of the driver's patience."""
    import signal
    import subprocess
    if limit_s is None:
        limit_s = float(os.environ.get("PCR_BENCH_LAUNCH_TIMEOUT", "1500"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--standalone", "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PCR_BENCH_SELF_LAUNCHED="1")
    child = subprocess.Popen(cmd, env=env, start_new_session=True)          # its own process group: killable as a whole
    try:
        return child.wait(timeout=limit_s)
    except subprocess.TimeoutExpired:
        sys.stderr.write(f"bench.py: the {n}-rank run exceeded {limit_s:.0f} s; killing its process group\n")
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(child.pid, sig)                                  # exactly the group this process started
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


HBM_PEAK_GBS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (description, glyph, reductions | sigma | half length, bytes per point of compulsory input traffic)
    "C2": ("50M uniform pts, 4096^2, Point, Sum+Count+Average", "point", ("Sum", "Count", "Average"), 20),
    "point_avg": ("Point, Average", "point", ("Average",), 20),
    "C4": ("clustered (10k hotspots), Point, Max+Min", "point", ("Max", "Min"), 20),
    "gauss1": ("Gaussian sigma=1 r<=4, WeightedAverage", "gauss", 1.0, 20),
    "gauss1.8": ("Gaussian sigma=1.8 r<=6, WeightedAverage", "gauss", 1.8, 20),
    "gauss2": ("Gaussian sigma=2 r<=6, WeightedAverage", "gauss", 2.0, 20),
    "gauss4": ("Gaussian sigma=4 r<=12 (C3), WeightedAverage", "gauss", 4.0, 20),
    "gauss16": ("Gaussian sigma=16 r<=48, WeightedAverage", "gauss", 16.0, 20),
    "line16": ("Line hl=16 per-point direction (C3), WeightedAverage", "line", 16.0, 24),
    "C5_point": ("1B uniform pts, 16384^2, Point, Average", "point", ("Average",), 20),
    "C5_gauss1": ("1B uniform pts, 16384^2, Gaussian sigma=1 r<=4, Average", "gauss", 1.0, 20),
}
PER_GLYPH = ("point_avg", "C4", "line16", "gauss1", "gauss4", "gauss16")
PER_GLYPH_NAME = {"C4": "clustered"}          # key of the leg in the line's per_glyph object

# HBM bytes per point a kernel has to move BY ITS OWN DESIGN (what it reads + what it writes, not the whole path's
# 20 B/point): the honest numerator for that kernel's own bandwidth, reported as roofline.kernel_own next to the
# contract's path-level figure.  Point path, C2: profiles/r02_C2_rocprof.md.
KERNEL_OWN_BYTES_PER_POINT = {
    "k_bin_count": 20,        # x, y in; 4-byte routing key out
    "k_bin_scatter": 16,      # key + value in; 8-byte record out
    "k_tile_accum": 8,        # record in (+ the state planes, per cell)
    "k_bin16_count": 16,      # x, y in
    "k_bin16_scatter": 36,    # x, y, value in; 16-byte record out (+ 4 per extra channel)
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
        spec = pcr.gaussian_splat_spec("value", default_sigma=arg, max_radius_cells=max_r)
        if workload.startswith("C5"):
            spec.type = pcr.ReductionType.Average                    # configs[4]: "Point + Gaussian sigma=1, Average"
        specs.append(spec)
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


def device_cloud_uniform(n, x_lo, x_hi, y_lo, y_hi, seed):
    """n uniform points generated ON the device, straight into a device-resident pcr.PointCloud."""
    c = pcr.PointCloud.create(max(n, 1), pcr.MemoryLocation.Device)
    if c is None:
        raise MemoryError("bench: cannot allocate the device cloud")
    c.add_channel("value", pcr.DataType.Float32)
    c.resize(n)
    ptrs = c.device_ptrs()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    for name, typestr, lo, hi in (("x", "<f8", x_lo, x_hi), ("y", "<f8", y_lo, y_hi), ("value", "<f4", 0.0, 1.0)):
        t = torch.as_tensor(pcr.DeviceArrayView(ptrs[name], (max(n, 1),), typestr, owner=c), device="cuda")
        t[:n].uniform_(lo, hi, generator=gen)
    torch.cuda.synchronize()
    return c


def csrc_sha():
    """Fingerprint of the kernel sources: a PMC traffic figure is only quoted for the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pointcloud-raster_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:16]


def cpu_baseline(workload, G, sample_pts, seed, budget_s=14.0):
    """The reference's CPU stages (oracle/pcr_cpu_pipeline.cpp: OpenMP assign, serial std::sort, per-update
    omp critical, finalize -- one pass per ReductionSpec, as the reference runs them) timed on this box's
    host cores at 1 thread and at all cores, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcr_oracle_py as O
    _, glyph, arg, _ = WORKLOADS[workload]
    x, y, v, ch = make_points(workload, sample_pts, G, 0.0, float(G), seed)
    og = O.make_grid((0.0, 0.0, float(G), float(G)))
    rmap = {"Sum": O.SUM, "Count": O.COUNT, "Average": O.AVERAGE, "Max": O.MAX, "Min": O.MIN}
    if glyph == "point":
        runs = [(rmap[name], None) for name in arg]
    elif glyph == "gauss":
        max_r = 12.0 if arg == 4.0 else min(4.0 * arg, 64.0)
        runs = [(O.WEIGHTED_AVERAGE, O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=arg, sigma_y=arg, max_radius=max_r))]
    else:
        runs = [(O.WEIGHTED_AVERAGE, O.make_glyph(O.GLYPH_LINE, half_length=arg, max_radius=arg + 2.0))]
    ncores = os.cpu_count() or 1
    try:
        ncores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    out = {}
    stages = {}
    # all cores: every update takes the `omp critical` lock, so with many threads the accumulate stage crawls (measured
    # on the GPU box, 256 threads: 0.06 Mpts/s) -- a much smaller sample keeps the default run short
    all_pts = min(sample_pts, 150_000 if glyph == "point" else 40_000)
    # 8 threads on >= 1 M points: the setting oracle/calibration.json was taken at (the reference's own thread-scaling
    # table stops at its core count, docs/BENCHMARK_RESULTS.md:52-55)
    t8_pts = min(sample_pts, 1_000_000 if glyph == "point" else 250_000)
    for label, threads, m in (("1", 1, sample_pts), ("8", min(8, ncores), t8_pts), ("all", ncores, all_pts)):
        chm = {k: a[:m] for k, a in ch.items()}
        t0 = time.perf_counter()
        for rtype, gl in runs:
            _, st = O.cpu_pipeline_run(og, rtype, x[:m], y[:m], v[:m], glyph=gl, threads=threads, **chm)
            for k, s in st.items():
                stages.setdefault(label, {}).setdefault(k, 0.0)
                stages[label][k] += s
        dt = time.perf_counter() - t0
        out[label] = (m / dt / 1e6, dt, m)
        if dt > budget_s and label == "1":             # keep the default run short on a slow host
            break
    cal = {}
    try:
        with open(os.path.join(ROOT, "oracle", "calibration.json")) as f:
            cal = json.load(f)["cases"]
    except (OSError, KeyError, ValueError):
        pass
    res = {"value": round(out["1"][0], 4), "unit": "Mpts/s", "cores": 1, "kind": "port",
           "sample": f"{sample_pts} pts of the same workload on the {G}^2 grid, ingest+finalize of every ReductionSpec, "
                     f"{out['1'][1]:.1f} s at 1 thread (oracle/pcr_cpu_pipeline.cpp: reference stages incl. its serial sort)",
           "nproc": ncores,
           "stage_seconds_1_thread": {k: round(s, 3) for k, s in stages.get("1", {}).items()}}
    if "8" in out:
        res["threads_8"] = {"value": round(out["8"][0], 4), "cores": min(8, ncores), "seconds": round(out["8"][1], 2),
                            "sample_points": out["8"][2],
                            "stage_seconds": {k: round(s, 3) for k, s in stages.get("8", {}).items()}}
    if "all" in out:
        res["all_cores"] = {"value": round(out["all"][0], 4), "cores": ncores, "seconds": round(out["all"][1], 2),
                            "sample_points": out["all"][2],
                            "stage_seconds": {k: round(s, 3) for k, s in stages.get("all", {}).items()},
                            "note": "every update under omp critical, serial sort: flat or slower with threads, as the "
                                    "reference (docs/BENCHMARK_RESULTS.md:52-55)"}
    if cal:
        res["port_over_reference"] = {k: c["port_over_reference"] for k, c in cal.items()}
        res["calibration"] = "oracle/calibration.json: port vs the true reference, both timed in the build container"
    # beside the port: the PRODUCT's own ExecutionMode.CPU (the host engine of host/src/host_engine.cpp -- what a drop-in user
    # of `exec_mode = CPU` gets), same sample, same reductions in ONE pipeline, at 1 thread and on all cores
    try:
        prod = {}
        for label, threads in (("1", 1), ("all", 0)):        # 0: every CPU the process may use (affinity AND cgroup quota)
            cfg = pcr.PipelineConfig()
            cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
            cfg.grid.cell_size_x, cfg.grid.cell_size_y = 1.0, -1.0
            cfg.grid.compute_dimensions()
            cfg.exec_mode = pcr.ExecutionMode.CPU
            cfg.cpu_threads = threads
            cfg.reductions = make_specs(workload)
            pipe = pcr.Pipeline.create(cfg)
            if pipe is None or pipe.engine() != "host":
                raise RuntimeError("ExecutionMode.CPU did not give the host engine: " + pcr.pipeline_create_error())
            cloud = make_cloud(x, y, v, ch)
            best = None
            for _ in range(2):                                   # (the first pass also pays the OpenMP team's start-up)
                pipe = pcr.Pipeline.create(cfg)
                t0 = time.perf_counter()
                pipe.ingest(cloud)
                pipe.finalize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            prod[label] = {"value": round(sample_pts / best / 1e6, 3), "cores": pipe.host_threads(), "seconds": round(best, 3)}
        res["product_host_engine"] = dict(prod, unit="Mpts/s", sample_points=sample_pts,
                                          what="pcr.Pipeline with exec_mode = CPU (stripe-owned direct scatter, no sort, no "
                                               "atomics), ingest + finalize of the same sample; checked against the oracle by "
                                               "tests/test_host_engine.py")
    except Exception as exc:                                     # informational: never cost the line
        res["product_host_engine"] = {"error": repr(exc)}
    return res


def time_steps(pipes, cloud, warmup, world, backend, ingests=1):
    """W untimed + K timed steps of ingest+finalize.  Returns (elapsed seconds MAX over ranks, per-kernel ms of the
    dominant kernel over the TIMED steps, per-kernel ms of every kernel over the WARM-UP steps).

    HIP events serialise the stream (~4 us per bracketed launch: 52 us on a 0.69 ms step with every kernel bracketed),
    so the timed steps bracket ONE kernel -- the dominant one, known from the fully bracketed warm-up steps -- and
    otherwise run as they do in production."""
    def step(sp):
        for _ in range(ingests):
            sp.ingest(cloud)
        sp.finalize()

    def drain(sps):
        kernels = {}
        for sp in sps:
            for name, (launches, ms) in sp.pipe.profile_read(True).items():
                k = kernels.setdefault(name, [0, 0.0])
                k[0] += launches
                k[1] += ms
        return kernels

    tables = []
    for sp in pipes[:warmup]:
        sp.pipe.profile_enable(True)
        step(sp)
        tables.append(drain([sp]))
    # The MEAN over the warm-up steps but the first (which is cold): with the last step alone the Point path's count and scatter
    # passes, 5 % apart, changed places from run to run and the line's roofline kernel with them.
    use = tables[1:] if len(tables) > 1 else tables
    warm = {}
    for t in use:
        for name, (launches, ms) in t.items():
            k = warm.setdefault(name, [0, 0.0])
            k[0] += launches / len(use)
            k[1] += ms / len(use)
    dom = max(warm, key=lambda k: warm[k][1]) if warm else ""
    on = os.environ.get("PCR_BENCH_NO_PROFILE") != "1"
    for sp in pipes[warmup:]:
        sp.pipe.profile_enable(on, dom)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for sp in pipes[warmup:]:
        step(sp)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, drain(pipes[warmup:]), warm


def measured_copy_gbs():
    """Device-to-device copy bandwidth of THIS box (read + write bytes per second): the practical roof next to the 8 TB/s
    data-sheet peak (SURVEY section 8d asks for both).  Three figures over the same 1 GiB buffers:
      float4_kernel / float4_kernel_nt   the library's hand-written float4 copy (pcr_hip_copy_kernel: 16 B per lane and
                                         access, four in flight per lane; plain and non-temporal) -- the shape
                                         MI355X_MICROARCH.md quotes 6.29 TB/s for; the better of the two is the yardstick
      torch                              torch's Tensor.copy_ (the runtime's blit kernel), round 3's denominator."""
    import ctypes as C
    from pcr import _cabi as A
    L = A.lib()
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")       # 1 GiB
    b = torch.empty_like(a)
    a.fill_(1.0)
    out = {}
    for _ in range(2):
        b.copy_(a)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(8):
        b.copy_(a)
    t1.record()
    torch.cuda.synchronize()
    out["torch"] = 2.0 * (1 << 30) / (t0.elapsed_time(t1) / 8 * 1e-3) / 1e9
    nbytes = a.numel() * 4
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        A.check(L.pcr_hip_event_create(C.byref(e)))
    for key, nt in (("float4_kernel", 0), ("float4_kernel_nt", 1)):
        for _ in range(2):
            A.check(L.pcr_hip_copy_kernel(b.data_ptr(), a.data_ptr(), nbytes, nt, None))
        A.check(L.pcr_hip_event_record(ev[0], None))                   # the null stream: the one the copies are launched on
        for _ in range(8):
            A.check(L.pcr_hip_copy_kernel(b.data_ptr(), a.data_ptr(), nbytes, nt, None))
        A.check(L.pcr_hip_event_record(ev[1], None))
        A.check(L.pcr_hip_stream_synchronize(None))
        ms = C.c_float()
        A.check(L.pcr_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        out[key] = 2.0 * nbytes / (ms.value / 8 * 1e-3) / 1e9
    for e in ev:
        L.pcr_hip_event_destroy(e)
    ok = bool(torch.equal(a[:1 << 20], b[:1 << 20]))
    del a, b
    if not ok:
        raise RuntimeError("copy kernel did not copy")
    return out


def roofline_of(kernels, info, n, bpp, workload, traffic_db):
    """Dominant kernel by summed HIP-event time; achieved = algorithmic bytes of one launch / its average duration."""
    if not kernels:
        return None, None
    dom = max(kernels, key=lambda k: kernels[k][1])
    launches, ms = kernels[dom]
    avg_ms = ms / launches
    n_launch = n                      # one launch of any binned-path kernel processes all N points of the ingest
    achieved = bpp * n_launch / (avg_ms * 1e-3) / 1e9
    traffic = None                    # HBM bytes per launch of that kernel from a rocprofv3 PMC run of THESE kernel sources
    pmc = traffic_db.get(workload, {})
    if traffic_db.get("csrc_sha") == csrc_sha() and dom in pmc and pmc.get("points_per_launch") == n:
        traffic = pmc[dom]["read_bytes"] + pmc[dom]["write_bytes"]
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "avg_kernel_ms": round(avg_ms, 4),
            "algorithmic_bytes_per_launch": bpp * n_launch}
    # the kernel's OWN bytes (its reads + writes by design; PMC traffic when it was measured on these sources): the
    # path-level 20 B/point above charges one pass of several with the whole path's input
    own = KERNEL_OWN_BYTES_PER_POINT.get(dom)
    own_bytes = traffic if traffic is not None else (own * n_launch if own else None)
    if own_bytes is not None:
        own_rate = own_bytes / (avg_ms * 1e-3) / 1e9
        roof["kernel_own"] = {"bytes_per_launch": own_bytes, "source": "pmc" if traffic is not None else "design",
                              "achieved": round(own_rate, 2), "frac": round(own_rate / HBM_PEAK_GBS, 5)}
    return roof, dom


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: C2 on one GPU, C5_point (+ per_glyph gauss1) on several")
    ap.add_argument("--points", type=int, default=0, help="points per GPU (default 50 M; C5: 1e9 / N)")
    ap.add_argument("--grid", type=int, default=0, help="grid width (default 4096; C5: 16384)")
    ap.add_argument("--rows", type=int, default=0, help="--weak only: rows per GPU when not square")
    ap.add_argument("--height", type=int, default=0,
                    help="N > 1 (strong): grid rows when not square -- e.g. 4 ranks on 16384 x 8192 have the 2048-row blocks of the "
                         "8-rank default (a one-GPU box admits six processes on its card, not eight)")
    ap.add_argument("--weak", action="store_true", help="N > 1: round 1's weak-scaled shape (tile-aligned blocks, no collective)")
    ap.add_argument("--path", default="auto", choices=["auto", "direct", "binned", "moments"])
    ap.add_argument("--cpu-sample", type=int, default=-1, help="points of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip the per_glyph / e2e_host legs")
    ap.add_argument("--unrouted", action="store_true",
                    help="N > 1: also time the Point step from an UNROUTED cloud (device partition + all-to-all inside the step)")
    ap.add_argument("--host-result", action="store_true", help="finalize into host memory (PCIe-inclusive)")
    ap.add_argument("--host-cloud", action="store_true", help="ingest a host-resident cloud (PCIe-inclusive)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N > 1 code path on a one-GPU box")
    ap.add_argument("--same-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    ap.add_argument("--no-selfcheck", action="store_true", help="N > 1: skip the untimed exchange self-check")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"],
                    help="N > 1: transport of the timed exchange -- torch.distributed (default) or the library's own "
                         "pcr_hip_comm_* over RCCL (include/pcr_hip.h); with the nccl backend the OTHER one is run once, "
                         "untimed, and compared bit for bit (native_exchange in the line)")
    ap.add_argument("--native-check-limit", type=float, default=120.0,
                    help="seconds the untimed native-vs-torch exchange check may take before the line is printed without it")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the driver's form: plain `python bench.py --gpus N`.  Decided before torch / pcr are imported.
        sys.exit(self_launch(args.gpus))
    # a GPU pipeline that cannot get its device is an error here, never a host-engine fallback (the only host-engine run of
    # this file asks for ExecutionMode.CPU by name: cpu_baseline's product leg)
    os.environ.setdefault("PCR_REQUIRE_GPU_ENGINE", "1")
    _imports()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("PCR_BENCH_PG_TIMEOUT", "600")))   # a stuck collective ends the run
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout)
        # communicator set-up is not part of any step
        warm = torch.zeros(1, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(warm)
        dist.barrier()

    strong = world > 1 and not args.weak
    workload = args.workload or ("C5_point" if strong else "C2")
    if strong:
        G = args.grid or 16384
        H = args.height or G
        n = args.points or 1_000_000_000 // world
    else:
        G = args.grid or 4096
        R = args.rows if args.rows > 0 else G                  # rows per GPU
        H = R * world                                          # one R-row block per GPU
        n = args.points or 50_000_000

    def make_cfg(wl):
        cfg = pcr.PipelineConfig()
        cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(H))
        cfg.grid.cell_size_x, cfg.grid.cell_size_y = 1.0, -1.0
        cfg.grid.compute_dimensions()
        cfg.exec_mode = pcr.ExecutionMode.GPU
        cfg.cuda_device_id = local_rank
        cfg.reductions = make_specs(wl)
        cfg.result_location = pcr.MemoryLocation.Host if args.host_result else pcr.MemoryLocation.Device
        cfg.scatter_path = {"auto": 0, "direct": 1, "binned": 2, "moments": 3}[args.path]
        # scratch arena (routing keys + records) sized at create, outside the clock, as the reference sizes
        # its MemoryPool in Pipeline::create (src/engine/pipeline.cpp:167-184)
        cfg.gpu_pool_size_bytes = 24 * n + (64 << 20)
        return cfg

    # ---- this rank's points ---------------------------------------------------------------------
    r0, r1 = row_block(rank, world, H)
    y_hi = float(H - r0)                                       # rows [r0, r1) <=> world y in (H - r1, H - r0)
    y_lo = float(H - r1)
    if strong:
        cloud = device_cloud_uniform(n, 2.0, G - 2.0, y_lo + (2.0 if rank == world - 1 else 0.0),
                                     y_hi - (2.0 if rank == 0 else 0.0), seed=42 + rank)
    else:
        x, y, v, ch = make_points(workload, n, G, y_lo, y_hi, seed=42 + rank)
        if WORKLOADS[workload][1] != "line" and not args.no_extras and world == 1:
            ch = dict(ch, direction=np.random.default_rng(4242).uniform(0, np.pi, n).astype(np.float32))   # per_glyph line16
        host_cloud = make_cloud(x, y, v, ch)
        cloud = host_cloud if args.host_cloud else host_cloud.to_device()
        del x, y, v, ch

    def run(wl, steps, warmup, the_cloud=None, cfg_edit=None, ingests=1):
        cfg = make_cfg(wl)
        if cfg_edit:
            cfg_edit(cfg)
        pipes = [ShardedPipeline(cfg, rank, world, device_id=local_rank, comm=args.comm if world > 1 else "torch")
                 for _ in range(warmup + steps)]
        elapsed, kernels, warm = time_steps(pipes, the_cloud if the_cloud is not None else cloud, warmup, world, args.backend,
                                            ingests)
        info = pipes[-1].pipe.last_scatter()
        sp = pipes[-1]
        extra = {"halo_rows": sp.halo, "tiles_local": sp.tiles_local, "comm": sp.comm_kind if world > 1 else None,
                 "collectives_per_step": dict(zip(("p2p_messages", "all_reduces"), sp.collectives_per_step())),
                 "halo_bytes_sent_per_step": sp.halo_bytes_per_step()}
        if world > 1 and not sp.tiles_local:
            # one extra, separately timed exchange on a finished pipeline (events on the engine stream)
            dist.barrier()
            sp.exchange(timed=True)
            extra["exchange_ms"] = round(sp.exchange_ms, 4)
        extra["_warm"] = {k: round(v[1], 4) for k, v in sorted(warm.items())}
        if WORKLOADS[wl][1] == "line" and "k_tile_line" in warm:
            # "achieved LDS-atomic ops/s next to the HBM fraction".  Every cell a segment visits is ONE
            # ds_add_u64 lane-operation in the tile's LDS window (k_tile_line_rec: visit count and fixed-point value share a
            # 64-bit word when both planes are wanted; round 3 issued one atomic per plane); the visits are the weight
            # plane's total (exact: integers in f32 below 2^24 per cell), read after the timed steps
            try:
                visits = sum(float(t.double().sum().item()) for t, kind in sp._plane_tensors() if kind == 2)
                nplanes = len(sp._plane_tensors())
                ms_line = warm["k_tile_line"][1] / max(warm["k_tile_line"][0], 1)
                extra["lds_atomic_lane_ops"] = visits
                extra["lds_atomic_lane_ops_per_s"] = visits / (ms_line * 1e-3)
                extra["lds_plane_updates_per_s"] = visits * nplanes / (ms_line * 1e-3)
            except Exception as exc:
                extra["lds_atomic_lane_ops_per_s"] = repr(exc)
        del pipes
        return elapsed, kernels, info, cfg, extra

    def selfcheck():
        """N > 1, outside every timed region: the first execution of the exchange on a real transport validates itself.
        (1) every generated point is valid on exactly one rank: sum over ranks of points_valid == points_total;
        (2) Gaussian leg: the weight held by ALL ranks' planes (owned rows + halo rows) before the exchange equals the weight
            in the OWNED rows after it -- nothing lost, nothing counted twice, whatever moved over the wire;
        (3) the touched-tile union is identical on every rank.
        Built in stages: the LOCAL part of a stage runs under try/except, then every rank takes part in the stage's
        collective whatever happened to it, carrying an error flag -- a rank that failed (out of memory creating the extra
        pipeline, an engine error) makes every rank skip the rest together instead of leaving the others in an all-reduce."""
        cpu = args.backend != "nccl"

        def allsum(vals, dtype):
            t = torch.tensor(vals, dtype=dtype, device="cpu" if cpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.cpu().tolist()

        def local(fn):
            try:
                return fn(), None
            except Exception as exc:
                return None, repr(exc)

        res = {"comm": args.comm}

        def stage_failed(name, err):
            """Collective: True on every rank when any rank's local part failed."""
            bad = allsum([1 if err else 0], torch.int64)[0]
            if bad:
                res["ok"] = False
                res["error"] = f"stage {name}: failed on {int(bad)} rank(s)" + (f"; this rank: {err}" if err else "")
            return bool(bad)

        # ---- stage 1: Point leg, valid points
        box = {}

        def s1():
            box["sp"] = ShardedPipeline(make_cfg(workload), rank, world, device_id=local_rank, comm=args.comm)
            box["sp"].ingest(cloud)
            box["sp"].pipe.synchronize()
            return int(box["sp"].pipe.last_scatter()["points_valid"])

        valid, err = local(s1)
        if stage_failed("point ingest", err):
            return res
        valid_total, offered_total = allsum([valid, int(n)], torch.int64)
        _, err = local(lambda: box["sp"].finalize())              # (its collectives are the exchange under test)
        if stage_failed("point finalize", err):
            return res
        box.clear()
        res["points_valid_all_ranks"] = int(valid_total)
        res["points_total"] = int(offered_total)
        ok = int(valid_total) == int(offered_total)

        # ---- stage 2: Gaussian leg, plane sums around the exchange
        def s2():
            spg = ShardedPipeline(make_cfg("C5_gauss1"), rank, world, device_id=local_rank, comm=args.comm)
            box["spg"] = spg
            spg.ingest(cloud)
            spg.pipe.synchronize()
            planes = spg._plane_tensors()
            s0 = spg.pipe.state_row_begin()
            o0, o1 = spg.own
            pre = [float(t.double().sum().item()) for t, _ in planes]
            halo_pre = [float((t.double().sum() - t[o0 - s0:o1 - s0].double().sum()).item()) for t, _ in planes]
            torch.cuda.synchronize()
            return pre, halo_pre

        got, err = local(s2)
        if stage_failed("gauss ingest", err):
            return res
        pre, halo_pre = got
        spg = box["spg"]
        _, err = local(lambda: (spg.exchange(), spg.pipe.synchronize(), torch.cuda.synchronize()))
        if stage_failed("gauss exchange", err):
            return res

        def s3():
            planes = spg._plane_tensors()
            s0 = spg.pipe.state_row_begin()
            o0, o1 = spg.own
            post = [float(t[o0 - s0:o1 - s0].double().sum().item()) for t, _ in planes]
            return post, int(spg._touched.to(torch.int64).sum().item())

        got, err = local(s3)
        if stage_failed("gauss sums", err):
            return res
        post, mine = got
        tot = allsum(pre + post + halo_pre, torch.float64)
        k = len(pre)
        pre_t, post_t, halo_t = tot[:k], tot[k:2 * k], tot[2 * k:]
        rel = [abs(a - b) / max(abs(a), 1e-30) for a, b in zip(pre_t, post_t)]
        res["gauss_planes"] = k
        res["gauss_plane_sums_before_exchange"] = [round(v, 3) for v in pre_t]
        res["gauss_plane_sums_owned_rows_after"] = [round(v, 3) for v in post_t]
        res["gauss_halo_rows_sum_before_exchange"] = [round(v, 3) for v in halo_t]
        res["gauss_max_rel_diff"] = max(rel) if rel else 0.0
        ok = ok and all(r <= 2e-6 for r in rel)          # f32 merges of f32 plane cells, summed in f64
        if not spg.tiles_local:
            ok = ok and all(v > 0 for v in halo_t)       # the blocks cut reference tiles: the halo rows DID hold weight
        t_all = allsum([mine], torch.int64)[0]
        res["touched_tiles"] = mine
        ok = ok and t_all == mine * world
        _, err = local(lambda: spg.pipe.finalize())
        if stage_failed("gauss finalize", err):
            return res

        # ---- stage 4: the pipeline is finalized AGAIN (state survives finalize, src/engine/pipeline.cpp:1344-1364): a second
        # exchange with nothing new accumulated must change nothing -- the apron rows that were sent hold the identity again --
        # and after a second ingest of the same cloud the owned rows hold exactly twice the weight (round 4's exchange left
        # the apron rows as they were and counted the first halo again at every finalize)
        def s4():
            spg.finalize()
            spg.pipe.synchronize()
            torch.cuda.synchronize()
            planes = spg._plane_tensors()
            s0 = spg.pipe.state_row_begin()
            o0, o1 = spg.own
            again = [float(t[o0 - s0:o1 - s0].double().sum().item()) for t, _ in planes]
            # (the apron rows themselves, not a difference of two large sums: exactly 0 is what is asserted)
            halo_now = [float((t[:o0 - s0].double().abs().sum() + t[o1 - s0:].double().abs().sum()).item()) for t, _ in planes]
            spg.ingest(cloud)
            spg.finalize()
            spg.pipe.synchronize()
            torch.cuda.synchronize()
            twice = [float(t[o0 - s0:o1 - s0].double().sum().item()) for t, _ in planes]
            return again, halo_now, twice

        got, err = local(s4)
        box.clear()
        del spg
        if stage_failed("gauss second finalize", err):
            return res
        again, halo_now, twice = got
        unchanged = allsum([0 if a == b else 1 for a, b in zip(again, post)], torch.int64)
        tot2 = allsum(halo_now + twice, torch.float64)
        halo_left, twice_t = tot2[:k], tot2[k:]
        rel2 = [abs(t2 - 2 * a) / max(abs(2 * a), 1e-30) for t2, a in zip(twice_t, pre_t)]
        res["refinalize_owned_rows_unchanged"] = not any(unchanged)
        res["refinalize_halo_rows_sum"] = [round(v, 6) for v in halo_left]
        res["second_ingest_max_rel_diff_vs_twice"] = max(rel2) if rel2 else 0.0
        ok = ok and not any(unchanged) and all(v == 0 for v in halo_left) and all(r <= 4e-6 for r in rel2)
        res["ok"] = bool(ok)
        return res

    def native_exchange_check():
        """N > 1 over RCCL, untimed, once: the SAME Gaussian round through both transports of the exchange -- the library's
        own pcr_hip_comm_halo_reduce (ncclSend / ncclRecv to rank +- 1, agreed geometry, merge kernel) and torch.distributed's
        batch_isend_irecv -- must leave bit-identical owned rows (both merge `mine + neighbour's` per cell).  Runs under the
        caller's deadline (a hung collective must not cost the line)."""
        cpu = args.backend != "nccl"
        res = {}
        sps = {}
        err = None
        # Set-up in stages, each closed by an all-reduce of an error flag: a rank whose local part fails (no cloud, out of
        # memory) must not leave the others waiting inside the NEXT stage's collective (creating a native pipeline broadcasts
        # the communicator's id).
        def agreed(stage, fn):
            nonlocal err
            if not err:
                try:
                    fn()
                except Exception as exc:
                    err = f"{stage}: {exc!r}"
            flag = torch.tensor([1 if err else 0], dtype=torch.int64, device="cpu" if cpu else "cuda")
            dist.all_reduce(flag)
            return int(flag.item())

        def make(kind):
            sps[kind] = ShardedPipeline(make_cfg("C5_gauss1"), rank, world, device_id=local_rank, comm=kind)

        def ingest_both():
            # ONE ingest, then the same bits in both pipelines: the Gaussian tiles merge into the planes with float atomics, so
            # two ingests of one cloud differ in the last bit -- what is compared here is the exchange, not the scatter
            a, b = sps["torch"], sps["native"]
            a.ingest(cloud)
            a.pipe.synchronize()
            torch.cuda.synchronize()
            for (ta, _), (tb, _) in zip(a._plane_tensors(), b._plane_tensors()):
                tb.copy_(ta)
            b._touched.copy_(a._touched)
            torch.cuda.synchronize()

        for stage, fn in (("prerequisites", lambda: cloud.count()), ("torch pipeline", lambda: make("torch")),
                          ("native pipeline", lambda: make("native")), ("ingest", ingest_both)):
            bad = agreed(stage, fn)
            if bad:
                return {"ok": False, "error": f"set-up failed on {bad} rank(s)" + (f"; this rank: {err}" if err else "")}
        for kind in ("torch", "native"):
            sps[kind].exchange(timed=True)
            sps[kind].pipe.synchronize()
            res[f"exchange_ms_{kind}"] = round(sps[kind].exchange_ms, 4)
        torch.cuda.synchronize()
        a, b = sps["torch"], sps["native"]
        s0 = a.pipe.state_row_begin()
        o0, o1 = a.own
        same = True
        for (ta, _), (tb, _) in zip(a._plane_tensors(), b._plane_tensors()):
            same = same and bool(torch.equal(ta[o0 - s0:o1 - s0].view(torch.int32), tb[o0 - s0:o1 - s0].view(torch.int32)))
        same = same and bool(torch.equal(a._touched, b._touched))
        t = torch.tensor([1 if same else 0], dtype=torch.int64, device="cpu" if cpu else "cuda")
        dist.all_reduce(t)
        res["ranks_bit_identical"] = int(t.item())
        res["ok"] = int(t.item()) == world
        res["halo_bytes_sent_per_step"] = b.halo_bytes_per_step()
        for sp in sps.values():
            sp.pipe.finalize()
            sp.close()
        return res

    traffic_db = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            traffic_db = json.load(f)
    except (OSError, ValueError):
        pass

    elapsed, kernels, info, cfg, extra = run(workload, args.steps, args.warmup)
    warm_table = extra.pop("_warm")
    check = None
    if strong and not args.no_selfcheck:
        try:
            check = selfcheck()
        except Exception as exc:                                      # a broken check must be visible, not fatal to the line
            check = {"ok": False, "error": repr(exc)}

    out = None
    if rank == 0:
        desc, glyph, _, bpp = WORKLOADS[workload]
        ms_per_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed / 1e6
        out = {
            "metric": "Mpts/s ingest->finalize",
            "value": round(value, 2), "unit": "Mpts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{workload}: {desc}", "points_per_gpu": n, "points_total": n * world,
                       "grid": f"{G}x{H}", "rows_per_gpu": r1 - r0, "glyph": glyph,
                       "reductions": [str(r.type).split(".")[-1] for r in cfg.reductions],
                       "scatter_path": info["path"], "lds_tile": list(info["lds_tile"]),
                       "num_bins": info["num_bins"],
                       "input": ("host-resident (H2D inside step)" if args.host_cloud else "device-resident") +
                                (", pre-routed by y (each rank holds the points of its row block)" if world > 1 else ""),
                       "result": "host (D2H inside step)" if args.host_result else "device-resident",
                       "parallelism": f"row-block x{world}",
                       # as the process group itself reports them (nccl IS RCCL on ROCm)
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "backend": ({"nccl": "rccl"}.get(dist.get_backend(), dist.get_backend())) if world > 1 else None,
                       "state_init": "inside the clock: planes are allocated at Pipeline.create but left undefined; the first scatter "
                                     "defines every cell (binned Point path: stored by the tile pass; other paths: k_state_init fill)",
                       **extra},
        }
        if check is not None:
            out["selfcheck"] = check
        roof, dom = roofline_of(kernels, info, n, bpp, workload, traffic_db)
        if roof:
            out["roofline"] = roof
            out["kernels_ms_per_step"] = dict(warm_table, _note="warm-up steps (mean of all but the first), every kernel bracketed by HIP events "
                                              "(which cost the stream ~4 us each); the timed steps bracket the dominant kernel only")
            # the whole step against the same roof: algorithmic bytes in + finalized bands out
            step_bytes = bpp * n + 4 * G * (r1 - r0) * len(cfg.reductions)
            out["step_roofline"] = {"algorithmic_bytes": step_bytes,
                                    "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 2), "unit": "GB/s",
                                    "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            if world == 1 and not args.same_device:
                try:
                    rates = measured_copy_gbs()
                    copy = max(rates["float4_kernel"], rates["float4_kernel_nt"])
                    roof["measured_copy_GBps"] = round(copy, 1)         # hand-written float4 copy kernel (pcr_hip_copy_kernel)
                    roof["measured_copy_kernel"] = {k: round(v, 1) for k, v in rates.items() if k != "torch"}
                    roof["torch_copy_GBps"] = round(rates["torch"], 1)   # round 3's denominator, kept for comparison
                    roof["frac_of_measured_copy"] = round(roof["achieved"] / copy, 5)
                    out["step_roofline"]["frac_of_measured_copy"] = round(out["step_roofline"]["achieved"] / copy, 5)
                except Exception as exc:                                # informational: never cost the headline
                    roof["measured_copy_GBps"] = repr(exc)

    # ---- the rest of the metric, same run ---------------------------------------------------------
    if not args.no_extras and args.workload is None and not args.host_cloud and not args.host_result:
        k_extra, w_extra = max(3, min(args.steps, 5)), 2
        per_glyph = {}
        legs = [(wl, 1) for wl in (PER_GLYPH if world == 1 else (("C5_gauss1",) if strong else ()))]
        if world == 1:
            legs.insert(1, ("point_avg", 2))       # two ingests into ONE pipeline: the second pays the planes' read-modify-write
        for wl, ingests in legs:
            name = PER_GLYPH_NAME.get(wl, wl.replace("C5_", "")) + ("_two_ingests" if ingests > 1 else "")
            leg_cloud = None
            try:
                if wl == "C4":
                    # BASELINE configs[3]: 10 000 hotspots, offsets N(0, 2 cells) (recipe after
                    # python/pcr/test_generators.py:560-633), generated on the host, resident before the clock
                    xc, yc, vc, _ = make_points("C4", n, G, y_lo, y_hi, seed=42 + rank)
                    leg_cloud = make_cloud(xc, yc, vc, {}).to_device()
                    del xc, yc, vc
                e2, k2, i2, c2, x2 = run(wl, k_extra, w_extra, the_cloud=leg_cloud, ingests=ingests)
                warm2 = x2.pop("_warm")
            except Exception as exc:                                   # a failing leg must not cost the headline
                per_glyph[name] = {"error": repr(exc)}
                continue
            finally:
                del leg_cloud
            if rank == 0:
                bpp2 = WORKLOADS[wl][3]
                roof2, dom2 = roofline_of(k2, i2, n, bpp2, wl, traffic_db)
                step_bytes2 = bpp2 * n * ingests + 4 * G * (r1 - r0) * len(c2.reductions)
                ms2 = e2 / k_extra * 1e3
                per_glyph[name] = {
                    "workload": WORKLOADS[wl][0] + (" -- two ingests of the cloud into one pipeline per step" if ingests > 1 else ""),
                    "ms_per_step": round(ms2, 4),
                    "Mpts/s": round(n * ingests * world * k_extra / e2 / 1e6, 2), "steps": k_extra,
                    "scatter_path": i2["path"], "dominant_kernel": dom2,
                    "roofline_frac": roof2["frac"] if roof2 else None,
                    "step_roofline_frac": round(step_bytes2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "kernels_ms_per_step": warm2,
                    **({"lds_atomic_lane_ops_per_s": x2["lds_atomic_lane_ops_per_s"],
                        "lds_atomic_lane_ops_per_step": x2["lds_atomic_lane_ops"],
                        "lds_plane_updates_per_s": x2["lds_plane_updates_per_s"]} if "lds_atomic_lane_ops" in x2 else {}),
                    **({"exchange": x2} if world > 1 else {})}
        if rank == 0:
            out["per_glyph"] = per_glyph
            base = out["ms_per_step"]
            if world == 1 and "gauss16" in per_glyph and "ms_per_step" in per_glyph["gauss16"]:
                out["gauss16_over_point"] = round(per_glyph["gauss16"]["ms_per_step"] / base, 3)

        if world == 1:
            # the drop-in default: PointCloud.create(n) is Host, result_location defaults to Host
            try:
                def host_result(c):
                    c.result_location = pcr.MemoryLocation.Host
                e3, _, _, _, _ = run(workload, 3, 1, the_cloud=host_cloud, cfg_edit=host_result)
                ms3 = e3 / 3 * 1e3
                moved = 20 * n + 4 * G * H * len(cfg.reductions)
                out["e2e_host"] = {"ms_per_step": round(ms3, 3), "Mpts/s": round(n / ms3 / 1e3, 2),
                                   "bytes_over_pcie": moved, "link_GBps": round(moved / ms3 / 1e6, 2),
                                   "what": "host-resident numpy cloud in (x, y, value staged H2D), host-resident bands out"}
            except Exception as exc:
                out["e2e_host"] = {"error": repr(exc)}
        elif strong and args.unrouted:
            # the same Point step from an UNROUTED cloud: each rank is handed an arbitrary 1/N of the points (uniform
            # over the whole grid); device-side partition + all-to-all to the owners are inside the step
            try:
                any_cloud = device_cloud_uniform(n, 2.0, G - 2.0, 2.0, H - 2.0, seed=1042 + rank)
                cfg_u = make_cfg(workload)
                pipes = [ShardedPipeline(cfg_u, rank, world, device_id=local_rank) for _ in range(1 + 3)]
                pipes[0].ingest_unrouted(any_cloud)
                pipes[0].finalize()
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for sp in pipes[1:]:
                    sp.ingest_unrouted(any_cloud)
                    sp.finalize()
                torch.cuda.synchronize()
                dist.barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                                 device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                if rank == 0:
                    ms_u = float(t.item()) / 3 * 1e3
                    out["unrouted"] = {"ms_per_step": round(ms_u, 3), "Mpts/s": round(n * world / ms_u / 1e3, 2),
                                       "what": "each rank handed an arbitrary 1/N of the cloud; step = device partition by "
                                               "owner + all-to-all of x, y, value (20 B/point, (N-1)/N of them cross xGMI) "
                                               "+ ingest + exchange + finalize"}
                del pipes, any_cloud
            except Exception as exc:
                if rank == 0:
                    out["unrouted"] = {"error": repr(exc)}

    if strong and not args.no_extras and args.workload is None:
        # the SAME problem on ONE GPU (rank 0 alone, the other ranks wait): the denominator of "x-fold at N GPUs on the
        # 16384^2 grid".  The driver's own N = 1 run measures C2 (4096^2), which is a different problem.
        one = None
        if rank == 0:
            try:
                # (this rank's own cloud stays: the native-exchange check below ingests it again -- round 4 deleted it here to
                # save 2.5 GB of 288, and rank 0 then failed the check's set-up while the others waited in its broadcast)
                total_pts = n * world
                whole = device_cloud_uniform(total_pts, 2.0, G - 2.0, 2.0, H - 2.0, seed=42)
                cfg1 = make_cfg(workload)
                cfg1.gpu_pool_size_bytes = 24 * total_pts + (64 << 20)
                pipes = [pcr.Pipeline.create(cfg1) for _ in range(3)]
                if any(p is None for p in pipes):
                    raise RuntimeError(pcr.pipeline_create_error())
                pipes[0].ingest(whole)
                pipes[0].finalize()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for p in pipes[1:]:
                    p.ingest(whole)
                    p.finalize()
                torch.cuda.synchronize()
                ms1 = (time.perf_counter() - t0) / 2 * 1e3
                one = {"ms_per_step": round(ms1, 3), "Mpts/s": round(total_pts / ms1 / 1e3, 2), "steps": 2,
                       "what": f"{workload} with all {total_pts} points on rank 0's GPU alone (same grid, unsharded pipeline)"}
                del pipes, whole
            except Exception as exc:
                one = {"error": repr(exc)}
            out["one_gpu_same_problem"] = one
            if "Mpts/s" in one:
                out["speedup_vs_one_gpu"] = round(out["value"] / one["Mpts/s"], 3)
        dist.barrier()

    # ---- N > 1 over RCCL: the transport that was NOT timed runs the same Gaussian round once and must agree bit for bit
    hard_exit = False
    if strong and world > 1 and not args.no_selfcheck and (args.backend == "nccl" or args.comm == "native"):
        import threading
        lock = threading.Lock()
        state = {"printed": False}

        def bail():
            # a collective of the check never returned: the measurements above are complete -- print them and leave
            try:                                     # where every thread of this rank is stuck, for whoever reads the log
                import faulthandler
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            except Exception:
                pass
            with lock:
                if state["printed"]:
                    return
                state["printed"] = True
                if rank == 0:
                    out["native_exchange"] = {"ok": False, "error": f"no answer within {args.native_check_limit:.0f} s (a hung "
                                              "collective); the timed legs above are unaffected"}
                    print(json.dumps(out), flush=True)
            os._exit(3)                              # the line is out; a hung collective is still a failed run

        timer = threading.Timer(args.native_check_limit, bail)
        timer.daemon = True
        timer.start()
        try:
            ne = native_exchange_check()
        except Exception as exc:
            ne = {"ok": False, "error": repr(exc)}
        timer.cancel()
        with lock:
            if state["printed"]:
                os._exit(3)
            state["printed"] = True                 # from here on the main thread prints
        hard_exit = not ne.get("ok", False)          # peers may be gone or stuck: do not wait for them in destroy_process_group
        if rank == 0:
            out["native_exchange"] = ne

    if rank == 0:
        sample = args.cpu_sample
        if sample < 0:
            glyph = WORKLOADS[workload][1]
            sample = {"point": 6_000_000, "gauss": 0, "line": 4_000_000}[glyph]       # a few seconds per thread setting
            if glyph == "gauss":
                sigma = WORKLOADS[workload][2]
                sample = max(200_000, int(1e9 / (2 * min(3 * sigma, 64) + 1) ** 2 / 4))
        if world == 1 and sample > 0:
            out["cpu_baseline"] = cpu_baseline(workload, G, min(sample, n), seed=42)
        print(json.dumps(out), flush=True)

    if world > 1:
        if hard_exit:
            # the two transports disagreed (or the check raised): the line above says so, and so does the exit code -- the
            # launcher (torchrun, self_launch) relays it
            sys.stdout.flush()
            os._exit(3)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
