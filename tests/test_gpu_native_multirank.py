"""The library's OWN multi-rank code with more than one rank -- pcr_hip_comm_halo_reduce (agreement, grouped send / recv to
rank +- 1, merge, apron reset), pcr_hip_comm_agree_max_i32, _allreduce_max_u32, _alltoall_counts / _alltoallv, _gatherv, and
pcr::ShardedPipeline (C++) on top: ingest, ingest_unrouted, exchange, finalize (again and again), gather, save_state --
THREE RANKS ON ONE GPU.  Real RCCL admits one rank per device, so the ten RCCL entry points csrc/comm.hip resolves by dlopen
are answered here by a test double (tests/native/fake_rccl.cpp: files as mailboxes, payloads staged through host memory;
PCR_HIP_RCCL points the library at it).  Everything but RCCL itself is the production code; RCCL itself is what the driver's
8-GPU run executes first.  Results against the unsharded oracle.  Two tests: the C++ pipeline (ranks that
never import torch) and pcr.distributed.ShardedPipeline(comm="native") (torch.distributed / gloo carries the bootstrap id; an
explicit PCR_HIP_RCCL wins over the RCCL torch has loaded)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G_W, G_H, WORLD = 160, 120, 3

_RANK = r"""
import os, sys, time
import numpy as np
HERE, rank, world, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
import pcr
assert "torch" not in sys.modules
from test_gpu_native_multirank import make_cfg, inputs, to_cloud
ident_path = os.path.join(out_dir, "id.bin")
if rank == 0:
    ident = pcr.NativeShardedPipeline.make_id()
    with open(ident_path + ".part", "wb") as f:
        f.write(ident)
    os.rename(ident_path + ".part", ident_path)
else:
    t0 = time.time()
    while not os.path.exists(ident_path):
        assert time.time() - t0 < 120, "no id from rank 0"
        time.sleep(0.01)
    ident = open(ident_path, "rb").read()
cfg = make_cfg(pcr)
cfg.output_path = os.path.join(out_dir, "whole.tif")
sp = pcr.NativeShardedPipeline.create(cfg, ident, rank, world, 0)
assert sp is not None, pcr.NativeShardedPipeline.create_error()
assert not sp.tiles_local() and sp.halo_rows() == 8
out = {"own": np.array([sp.row_begin(), sp.row_end()])}
nb = 6
# round 0: every rank is handed the WHOLE cloud (the engine keeps the points whose centre row it owns)
sp.ingest(to_cloud(pcr, inputs(0)).to_device())
sp.finalize()
for b in range(nb):
    out[f"r0b{b}"] = np.array(sp.result().band_array(b))
sp.finalize()                                   # nothing new: nothing may change (the apron rows that were sent are identity again)
for b in range(nb):
    out[f"r0again{b}"] = np.array(sp.result().band_array(b))
# round 1: an UNROUTED cloud -- rank r is handed the points i % world == r, wherever they lie
x, y, ch = inputs(1)
sel = np.arange(len(x)) % world == rank
got = sp.ingest_unrouted(to_cloud(pcr, (x[sel], y[sel], {k: v[sel] for k, v in ch.items()})))
out["received"] = np.array(got)
sp.finalize()
for b in range(nb):
    out[f"r1b{b}"] = np.array(sp.result().band_array(b))
whole = sp.gather(world - 1)
assert (whole is not None) == (rank == world - 1)
if whole is not None:
    for b in range(nb):
        out[f"whole{b}"] = np.array(whole.band_array(b))
sp.save_state(os.path.join(out_dir, "ck"))      # blocks cut tiles: the planes' owned rows travel to rank 0, which writes the tiles
out["bytes_sent"] = np.array(sp.bytes_sent())
np.savez(os.path.join(out_dir, f"n{rank}.npz"), **out)
del sp
print("rank", rank, "ok")
"""


def make_cfg(pcr):
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G_W), float(G_H))
    cfg.grid.tile_width, cfg.grid.tile_height = 64, 64           # 40-row blocks cut the 64-row tiles
    cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU
    cfg.cuda_device_id = 0
    g = pcr.gaussian_splat_spec("value", default_sigma=1.5, max_radius_cells=5.0)
    gs = pcr.gaussian_splat_spec("value", default_sigma=1.5, max_radius_cells=5.0)
    gs.type = pcr.ReductionType.Sum
    ln = pcr.line_splat_spec("value", direction_channel="dir", half_length_channel="hl", default_half_length=1.0, max_radius_cells=7.0)
    lc = pcr.line_splat_spec("value", direction_channel="dir", half_length_channel="hl", default_half_length=1.0, max_radius_cells=7.0)
    lc.type = pcr.ReductionType.Count
    ps = pcr.ReductionSpec()
    ps.value_channel, ps.type = "value", pcr.ReductionType.Sum
    pm = pcr.ReductionSpec()
    pm.value_channel, pm.type = "value", pcr.ReductionType.Max
    cfg.reductions = [g, gs, ln, lc, ps, pm]
    return cfg


def inputs(k):
    rng = np.random.default_rng(900 + k)
    n = 25000
    x, y = rng.uniform(-2, G_W + 2, n), rng.uniform(-2, G_H + 2, n)
    ch = {"value": rng.uniform(0, 1, n).astype(np.float32), "dir": rng.uniform(0, np.pi, n).astype(np.float32),
          "hl": rng.uniform(-5, 5, n).astype(np.float32)}
    return x, y, ch


def to_cloud(pcr, data):
    x, y, ch = data
    c = pcr.PointCloud.create(max(len(x), 1))
    c.set_x_array(x)
    c.set_y_array(y)
    c.resize(len(x))
    for name, arr in ch.items():
        c.add_channel(name, pcr.DataType.Float32)
        c.set_channel_array_f32(name, arr)
    return c


def run_ranks(tmp_path, code, extra_env=None):
    gxx = shutil.which("g++")
    if not gxx or not os.path.isdir("/opt/rocm/include/rccl"):
        pytest.skip("g++ or the RCCL headers are not available")
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.run([gxx, "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(HERE, "native", "fake_rccl.cpp"), "-o", fake, "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    env = dict(os.environ, PCR_HIP_RCCL=fake, PCR_FAKE_RCCL_DIR=str(tmp_path), PCR_REQUIRE_GPU_ENGINE="1", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, "-c", code, HERE, str(r), str(WORLD), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(WORLD)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung in the native collectives")
        outs.append((p.returncode, so, se))
    for r, (rc, so, se) in enumerate(outs):
        assert rc == 0, f"rank {r}: " + so[-1500:] + se[-3000:]


def verify(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pcr
    import pcr_oracle_py as O
    parts = [np.load(tmp_path / f"n{r}.npz") for r in range(WORLD)]
    assert [p["own"].tolist() for p in parts] == [[0, 40], [40, 80], [80, 120]]
    assert all(int(p["bytes_sent"]) > 0 for p in parts)                       # every rank has a neighbour and an apron
    og = O.make_grid((0, 0, G_W, G_H), tile=(64, 64))
    gg = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.5, sigma_y=1.5, max_radius=5.0)
    lg = O.make_glyph(O.GLYPH_LINE, half_length=1.0, max_radius=7.0)

    def want(x, y, ch):
        v = ch["value"]
        lk = dict(direction=ch["dir"], half_length=ch["hl"])
        return [O.run(og, O.WEIGHTED_AVERAGE, x, y, v, glyph=gg), O.run(og, O.SUM, x, y, v, glyph=gg),
                O.run(og, O.WEIGHTED_AVERAGE, x, y, v, glyph=lg, **lk), O.run(og, O.COUNT, x, y, v, glyph=lg, **lk),
                O.run(og, O.SUM, x, y, v, wide=True), O.run(og, O.MAX, x, y, v)]

    tol = [(1e-4, 1e-6), (1e-4, 1e-5), (1e-4, 1e-6), (0, 0), (1e-5, 1e-6), (0, 0)]

    def check(tag, w, what):
        for b, (rt, at) in enumerate(tol):
            got = np.vstack([p[f"{tag}{b}"] for p in parts])
            assert np.array_equal(np.isnan(got), np.isnan(w[b])), f"{what}, band {b}: NaN mask"
            m = ~np.isnan(w[b])
            assert (np.abs(got[m] - w[b][m]) <= at + rt * np.abs(w[b][m])).all(), f"{what}, band {b}"

    x0, y0, c0 = inputs(0)
    x1, y1, c1 = inputs(1)
    w0 = want(x0, y0, c0)
    check("r0b", w0, "first finalize")
    for b in range(6):                                        # a finalize with nothing new: the same bits
        for p in parts:
            assert np.array_equal(p[f"r0again{b}"], p[f"r0b{b}"], equal_nan=True), f"second finalize changed band {b}"
    X, Y = np.concatenate([x0, x1]), np.concatenate([y0, y1])
    C = {k: np.concatenate([c0[k], c1[k]]) for k in c0}
    w1 = want(X, Y, C)
    check("r1b", w1, "after the unrouted cloud")
    inside = (x1 >= 0) & (x1 <= G_W) & (y1 >= 0) & (y1 <= G_H)
    assert sum(int(p["received"]) for p in parts) == int(inside.sum())        # every point went to exactly one owner
    # gather: ONE grid on the last rank = the stacked strips; the ONE GeoTIFF rank 0 wrote
    for b in range(6):
        strips = np.vstack([p[f"r1b{b}"] for p in parts])
        assert np.array_equal(parts[WORLD - 1][f"whole{b}"], strips, equal_nan=True), f"gather, band {b}"
        assert np.array_equal(np.array(pcr.read_geotiff_band(str(tmp_path / "whole.tif"), b)), strips, equal_nan=True), f"GeoTIFF, band {b}"
    # the checkpoint rank 0 wrote from the gathered planes resumes in ONE unsharded pipeline: finalize gives the same bands
    cfg = make_cfg(pcr)
    cfg.state_dir, cfg.resume = str(tmp_path / "ck"), True
    one = pcr.Pipeline.create(cfg)
    assert one is not None, pcr.pipeline_create_error()
    one.finalize()
    for b, (rt, at) in enumerate(tol):
        got, w = np.array(one.result().band_array(b)), w1[b]
        assert np.array_equal(np.isnan(got), np.isnan(w)), f"resumed checkpoint, band {b}: NaN mask"
        m = ~np.isnan(w)
        assert (np.abs(got[m] - w[m]) <= at + rt * np.abs(w[m])).all(), f"resumed checkpoint, band {b}"


def test_three_ranks_on_one_gpu_through_the_native_collectives(tmp_path):
    """pcr::ShardedPipeline (C++; the ranks never import torch)."""
    run_ranks(tmp_path, _RANK)
    verify(tmp_path)


_PY_RANK = r"""
import os, sys
import numpy as np
HERE, rank, world, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
import torch
import torch.distributed as dist
import pcr
from pcr.distributed import ShardedPipeline
from test_gpu_native_multirank import make_cfg, inputs, to_cloud
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)          # carries the bootstrap id, nothing else
try:
    cfg = make_cfg(pcr)
    cfg.output_path = os.path.join(out_dir, "whole.tif")
    sp = ShardedPipeline(cfg, rank, world, device_id=0, comm="native")  # PCR_HIP_RCCL names the test double: it wins over torch's RCCL
    assert not sp.tiles_local and sp.halo == 8
    out = {"own": np.array(sp.own)}
    nb = 6
    sp.ingest(to_cloud(pcr, inputs(0)).to_device())
    sp.finalize()
    for b in range(nb):
        out[f"r0b{b}"] = np.array(sp.result().band_array(b))
    sp.finalize()
    for b in range(nb):
        out[f"r0again{b}"] = np.array(sp.result().band_array(b))
    x, y, ch = inputs(1)
    sel = np.arange(len(x)) % world == rank
    out["received"] = np.array(sp.ingest_unrouted(to_cloud(pcr, (x[sel], y[sel], {k: v[sel] for k, v in ch.items()}))))
    sp.finalize()
    for b in range(nb):
        out[f"r1b{b}"] = np.array(sp.result().band_array(b))
    whole = sp.gather(world - 1)
    assert (whole is not None) == (rank == world - 1)
    if whole is not None:
        for b in range(nb):
            out[f"whole{b}"] = np.array(whole.band_array(b))
    sp.save_state(os.path.join(out_dir, "ck"))
    import ctypes as C
    from pcr import _cabi as A
    sent = C.c_uint64()
    A.check(A.lib().pcr_hip_comm_stats(sp._comm, None, C.byref(sent)))
    out["bytes_sent"] = np.array(sent.value)
    np.savez(os.path.join(out_dir, f"n{rank}.npz"), **out)
    sp.close()
finally:
    dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_python_sharded_pipeline_with_native_comm_three_ranks(tmp_path):
    """pcr.distributed.ShardedPipeline(comm="native"): _exchange_native, the native route_cloud, the native _gather_rows and
    the gathered checkpoint, three ranks on one GPU (torch.distributed / gloo only carries the 128-byte id)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    run_ranks(tmp_path, _PY_RANK.replace('dist.init_process_group("gloo", rank=rank, world_size=world)',
                                         f'dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)'))
    verify(tmp_path)
