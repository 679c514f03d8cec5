"""pcr_hip_comm_* (include/pcr_hip.h): the native exchange step over RCCL.

What a one-GPU box can check: RCCL loads through the library, a world-1 communicator bootstraps from a unique id, the
exchange entry points validate their arguments, the C++ ShardedPipeline (pcr.NativeShardedPipeline) gives the unsharded
result at world 1, and what RCCL says when two ranks share ONE device (it refuses duplicate devices: then the two-rank
native path stays unmeasured here and the test records that instead of failing -- the gloo-staged exchange covers the
logic, tests/test_gpu_sharded_two_ranks.py)."""
import ctypes as C
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# RCCL is only ever loaded in CHILD processes here: the pytest process itself maps ROCm's HIP runtime (tests that bind the
# C-ABI directly) and later torch's libraries (tests that import torch); a third party's RCCL in between would be a second
# copy of rocm_smi / roctx in one process.  A child that crashes at exit fails its test through the exit code.
_WORLD_ONE = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from conftest import load_cabi
A = load_cabi()
L = A.lib()
assert L.pcr_hip_comm_available() == 1
ident = (C.c_uint8 * 128)()
A.check(L.pcr_hip_comm_unique_id(ident))
assert any(ident)
comm = C.c_void_p()
A.check(L.pcr_hip_comm_create(C.byref(comm), ident, 0, 1, 0))
rank, world = C.c_int(-1), C.c_int(-1)
A.check(L.pcr_hip_comm_rank(comm, C.byref(rank), C.byref(world)))
assert (rank.value, world.value) == (0, 1)
# world 1: both steps are no-ops, but the arguments are still validated
buf = A.DeviceBuffer.from_numpy(np.ones((8, 16), dtype=np.float32))
planes = (A.HaloPlane * 1)()
planes[0].d_plane, planes[0].kind = buf.ptr.value, A.PLANE_SUM
A.check(L.pcr_hip_comm_halo_reduce(comm, planes, 1, 16, 0, 8, 2, 6, 2, None))
for bad in ((6, 12),):                                   # owned rows outside the window
    try:
        A.check(L.pcr_hip_comm_halo_reduce(comm, planes, 1, 16, 0, 8, bad[0], bad[1], 2, None))
        raise SystemExit("halo_reduce accepted owned rows outside the state window")
    except A.PcrHipError:
        pass
try:
    A.check(L.pcr_hip_comm_create(C.byref(C.c_void_p()), ident, 3, 2, 0))     # rank outside [0, world)
    raise SystemExit("comm_create accepted rank 3 of 2")
except A.PcrHipError:
    pass
words = A.DeviceBuffer.from_numpy(np.arange(4, dtype=np.uint32))
A.check(L.pcr_hip_comm_allreduce_max_u32(comm, words.ptr, 4, None))          # world 1: identity
assert (words.to_numpy(np.uint32, 4) == np.arange(4, dtype=np.uint32)).all()
# the variable-size transfers at world 1: a rank's own group is a device-to-device copy, the agreement is still passed
x = np.arange(1000, dtype=np.float64)
v = np.arange(1000, dtype=np.float32) * 0.5
sx, sv = A.DeviceBuffer.from_numpy(x), A.DeviceBuffer.from_numpy(v)
rx, rv = A.DeviceBuffer.from_numpy(np.zeros(1200, np.float64)), A.DeviceBuffer.from_numpy(np.zeros(1200, np.float32))
srcs, dsts = (C.c_void_p * 2)(sx.ptr.value, sv.ptr.value), (C.c_void_p * 2)(rx.ptr.value, rv.ptr.value)
elems = (C.c_int32 * 2)(8, 4)
send, recv = (C.c_uint64 * 64)(1000), (C.c_uint64 * 64)()
A.check(L.pcr_hip_comm_alltoall_counts(comm, send, recv, None))
assert recv[0] == 1000
recv[0] = 0
A.check(L.pcr_hip_comm_alltoallv(comm, 2, srcs, dsts, elems, send, 1200, recv, None))
A.check(L.pcr_hip_stream_synchronize(None))
assert recv[0] == 1000 and (rx.to_numpy(np.float64, 1000) == x).all() and (rv.to_numpy(np.float32, 1000) == v).all()
for what, args in (("no room", (2, srcs, dsts, elems, send, 999, recv)), ("9 arrays", (9, srcs, dsts, elems, send, 1200, recv))):
    try:
        A.check(L.pcr_hip_comm_alltoallv(comm, *args, None))
        raise SystemExit("alltoallv accepted: " + what)
    except A.PcrHipError:
        pass
rx2 = A.DeviceBuffer.from_numpy(np.zeros(1000, np.float64))
A.check(L.pcr_hip_comm_gatherv(comm, 1, (C.c_void_p * 1)(sx.ptr.value), (C.c_void_p * 1)(rx2.ptr.value), (C.c_int32 * 1)(8),
                               1000, 1000, recv, 0, None))
A.check(L.pcr_hip_stream_synchronize(None))
assert (rx2.to_numpy(np.float64, 1000) == x).all()
try:
    A.check(L.pcr_hip_comm_gatherv(comm, 1, (C.c_void_p * 1)(sx.ptr.value), (C.c_void_p * 1)(rx2.ptr.value), (C.c_int32 * 1)(8),
                                   1000, 1000, recv, 3, None))
    raise SystemExit("gatherv accepted root 3 of 1")
except A.PcrHipError:
    pass
A.check(L.pcr_hip_comm_destroy(comm))
print("world-one comm ok")
"""


def _run_child(code, *args, timeout=240):
    import subprocess
    out = subprocess.run([sys.executable, "-c", code, HERE, *args], capture_output=True, text=True, timeout=timeout,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return out.stdout


def test_rccl_loads_and_world_one_communicator():
    assert "world-one comm ok" in _run_child(_WORLD_ONE)


_SHARDED_WORLD_ONE = r"""
import os, sys
import numpy as np
HERE = sys.argv[1]
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "pointcloud-raster_amd", "python")):
    sys.path.insert(0, p)
import pcr
import pcr_oracle_py as O
G, n = 200, 40000
rng = np.random.default_rng(5)
x, y = rng.uniform(1, G - 1, n), rng.uniform(1, G - 1, n)
v = rng.uniform(0, 1, n).astype(np.float32)
cfg = pcr.PipelineConfig()
cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
cfg.grid.tile_width, cfg.grid.tile_height = 64, 64
cfg.grid.compute_dimensions()
cfg.exec_mode = pcr.ExecutionMode.GPU
cnt = pcr.ReductionSpec()
cnt.value_channel, cnt.type = "value", pcr.ReductionType.Count
cfg.reductions = [pcr.gaussian_splat_spec("value", default_sigma=1.0, max_radius_cells=4.0), cnt]
ident = pcr.NativeShardedPipeline.make_id()
assert len(ident) == 128
sp = pcr.NativeShardedPipeline.create(cfg, ident, 0, 1, 0)
assert sp is not None, pcr.NativeShardedPipeline.create_error()
assert (sp.row_begin(), sp.row_end()) == (0, G) and pcr.NativeShardedPipeline.row_block(1, 4, 1000) == (250, 500)
cloud = pcr.PointCloud.create(n)
cloud.set_x_array(x)
cloud.set_y_array(y)
cloud.add_channel("value", pcr.DataType.Float32)
cloud.set_channel_array_f32("value", v)
sp.ingest(cloud.to_device())
sp.finalize()
res = sp.result()
og = O.make_grid((0.0, 0.0, float(G), float(G)), tile=(64, 64))
assert np.array_equal(np.array(res.band_array(1)), O.run(og, O.COUNT, x, y, v), equal_nan=True)
gl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=1.0, sigma_y=1.0, max_radius=4.0)
exact = O.run(og, O.WEIGHTED_AVERAGE, x, y, v, glyph=gl, wide=True).astype(np.float64)
got = np.array(res.band_array(0)).astype(np.float64)
m = ~np.isnan(got) & ~np.isnan(exact)
assert (np.isnan(got) != np.isnan(exact)).sum() <= 4
assert (np.abs(got[m] - exact[m]) <= 1e-4 * np.maximum(1e-3, np.abs(exact[m]))).all()
whole = sp.gather(0)
assert whole is not None and np.array_equal(np.array(whole.band_array(1)), np.array(res.band_array(1)), equal_nan=True)
del sp
# an UNROUTED cloud through the C++ pipeline: partitioned by owner on the device (points outside the grid dropped there), ingested
x2 = np.concatenate([x, rng.uniform(-50, -1, 500)])             # 500 points outside the grid
y2 = np.concatenate([y, rng.uniform(1, G - 1, 500)])
v2 = np.concatenate([v, np.ones(500, np.float32)])
cfg.output_path = os.path.join(sys.argv[2], "one.tif")
sp = pcr.NativeShardedPipeline.create(cfg, ident, 0, 1, 0)
cloud = pcr.PointCloud.create(n + 500)
cloud.set_x_array(x2)
cloud.set_y_array(y2)
cloud.add_channel("value", pcr.DataType.Float32)
cloud.set_channel_array_f32("value", v2)
assert sp.ingest_unrouted(cloud) == n                           # a host cloud: staged, partitioned, the 500 strays dropped
sp.finalize()
assert np.array_equal(np.array(sp.result().band_array(1)), O.run(og, O.COUNT, x, y, v), equal_nan=True)
assert np.array_equal(np.array(pcr.read_geotiff_band(cfg.output_path, 1)), np.array(sp.result().band_array(1)), equal_nan=True)
del sp
print("sharded world-one ok")
"""


def test_native_sharded_pipeline_world_one_matches_oracle(tmp_path):
    """The C++ ShardedPipeline (pcr.NativeShardedPipeline) at world 1: id from RCCL, no exchange, the unsharded result."""
    assert "sharded world-one ok" in _run_child(_SHARDED_WORLD_ONE, str(tmp_path))


_PY_NATIVE_ROUTE = r"""
import ctypes as C, os, sys
import numpy as np
import torch                                   # before pcr: one shared HIP runtime
HERE = sys.argv[1]
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
import pcr
from pcr import _cabi as A
from pcr.distributed import partition_cloud, _alltoallv_native
L = A.lib()
ident = (C.c_uint8 * 128)()
A.check(L.pcr_hip_comm_unique_id(ident))
comm = C.c_void_p()
A.check(L.pcr_hip_comm_create(C.byref(comm), ident, 0, 1, 0))
G, n = 300, 50000
rng = np.random.default_rng(12)
x, y = rng.uniform(-20, G + 20, n), rng.uniform(-20, G + 20, n)      # some points outside the grid: dropped at the partition
v = rng.uniform(0, 1, n).astype(np.float32)
cloud = pcr.PointCloud.create(n)
cloud.set_x_array(x)
cloud.set_y_array(y)
cloud.add_channel("value", pcr.DataType.Float32)
cloud.set_channel_array_f32("value", v)
dev = cloud.to_device()
gc = pcr.GridConfig()
gc.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
gc.compute_dimensions()
counts, grouped = partition_cloud(dev, gc, [(0, G)])
inside = (x >= 0) & (x <= G) & (y >= 0) & (y <= G)
assert counts == [int(inside.sum())]
mine = _alltoallv_native(comm, dev, counts, grouped, 1, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert mine.count() == counts[0] and mine.location() == pcr.MemoryLocation.Device
host = mine.to_host()
got = np.stack([np.array(host.x_array()), np.array(host.y_array()), np.array(host.channel_array_f32("value")).astype(np.float64)])
want = np.stack([x[inside], y[inside], v[inside].astype(np.float64)])
assert np.array_equal(got[:, np.lexsort(got)], want[:, np.lexsort(want)])          # the same points, whatever their order
A.check(L.pcr_hip_comm_destroy(comm))
print("python native route ok")
"""


def test_python_route_through_the_native_alltoallv_world_one():
    """route_cloud(comm=...) -- the Python side of pcr_hip_comm_alltoall_counts / _alltoallv -- at world 1: the partition's groups
    land in a device cloud through the library's own transfer (two ranks need two GPUs: the driver's run)."""
    assert "python native route ok" in _run_child(_PY_NATIVE_ROUTE)


def _two_rank_worker(rank, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    import pcr
    from pcr.distributed import ShardedPipeline
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    msg = "ok"
    try:
        cfg = pcr.PipelineConfig()
        cfg.grid.bounds = pcr.BBox(0.0, 0.0, 160.0, 120.0)
        cfg.grid.compute_dimensions()
        cfg.exec_mode = pcr.ExecutionMode.GPU
        cfg.cuda_device_id = 0
        cfg.reductions = [pcr.gaussian_splat_spec("value", default_sigma=2.0, max_radius_cells=6.0)]
        try:
            sp = ShardedPipeline(cfg, rank, 2, device_id=0, comm="native")
            sp.close()
        except RuntimeError as exc:                   # RCCL: two ranks on one device
            msg = "refused: " + str(exc)[:200]
    finally:
        with open(os.path.join(out_dir, f"native_r{rank}.txt"), "w") as f:
            f.write(msg)
        dist.destroy_process_group()


def test_two_ranks_on_one_device_native_is_created_or_cleanly_refused(tmp_path):
    """RCCL allows one rank per device.  Either the two-rank native communicator comes up on this box, or both ranks get a
    RuntimeError naming RCCL -- never a hang or a crash."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_two_rank_worker, args=(r, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        if p.is_alive():
            p.kill()
            pytest.fail("native communicator bootstrap hung with two ranks on one device")
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]      # a crash at exit counts
    texts = [open(os.path.join(str(tmp_path), f"native_r{r}.txt")).read() for r in range(2)]
    assert all(t == "ok" or t.startswith("refused: ") for t in texts), texts
    print("two ranks on one device, native comm:", texts)
