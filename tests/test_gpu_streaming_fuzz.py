"""Randomised differential test of the ways a cloud reaches the pipeline and a result leaves it: the
grids, reductions, filters and clouds of tests/test_gpu_out_of_core_fuzz.py, every cloud through ONE of

  host      Pipeline.ingest of a pageable host cloud (staged)
  device    ... of a device cloud
  pinned    Pipeline.ingest_async of a page-locked cloud (nothing waited for until finalize)
  file      write_point_cloud (PCRP) -> Pipeline.ingest_file in chunks of a random size (double-buffered, the last chunk ragged)
  readdev   write_point_cloud -> read_point_cloud(..., Device) -> ingest

(not CSV: as upstream, the CSV reader makes every extra column a Float64 channel and the pipeline takes Float32 value channels
only -- `pipeline: value channel must be Float32`, the reference's own message), with the result on the host or on the device,
written as ONE GeoTIFF at output_path and read back.  Yardstick: the same library fed plain host clouds (assert_bands_match: bit
for bit for Max / Min / Count of Points and Lines, fp32 re-association for sums -- chunks change the grouping of a Point sum).
The GeoTIFF must hold the result's bands bit for bit.  PCR_STREAM_FUZZ_SEEDS=a:b
soaks a range."""
import os

import numpy as np
import pytest

import pcr
import test_gpu_out_of_core_fuzz as F
from test_gpu_pipeline_api import cloud_from

pytestmark = pytest.mark.gpu
CHANNELS = ("a", "b", "cls", "dir")


def seeds():
    env = os.environ.get("PCR_STREAM_FUZZ_SEEDS")
    if env:
        a, b = env.split(":")
        return list(range(int(a), int(b)))
    return list(range(12))


def pinned(c):
    n = len(c["x"])
    pc = pcr.PointCloud.create(max(n, 1), pcr.MemoryLocation.HostPinned)
    pc.set_x_array(c["x"])
    pc.set_y_array(c["y"])
    pc.resize(n)
    for name in CHANNELS:
        pc.add_channel(name, pcr.DataType.Float32)
        if n:
            pc.set_channel_array_f32(name, c[name])
    return pc


@pytest.mark.parametrize("seed", seeds())
def test_every_way_in_and_out_equals_plain_ingest(seed, tmp_path):
    og, specs, filt, clouds, _ = F.build(seed)
    rng = np.random.default_rng(55000 + seed)
    plain = pcr.Pipeline.create(F.config(og, specs, filt))
    assert plain is not None, pcr.pipeline_create_error()
    for c in clouds:
        plain.ingest(cloud_from(c["x"], c["y"], {k: c[k] for k in CHANNELS}, "host"))
    plain.finalize()
    want = [np.array(plain.result().band_array(i)) for i in range(len(specs))]

    out = str(tmp_path / "out.tif")
    on_device = bool(rng.uniform() < 0.5)
    kw = dict(output_path=out)
    if on_device:
        kw["result_location"] = pcr.MemoryLocation.Device
    pipe = pcr.Pipeline.create(F.config(og, specs, filt, **kw))
    assert pipe is not None, pcr.pipeline_create_error()
    keep_alive, ways = [], []
    for k, c in enumerate(clouds):
        n = len(c["x"])
        way = str(rng.choice(["host", "device", "pinned", "file", "readdev"]))
        ways.append(way)
        host = cloud_from(c["x"], c["y"], {name: c[name] for name in CHANNELS}, "host")
        if way == "host":
            pipe.ingest(host)
        elif way == "device":
            pipe.ingest(host.to_device())
        elif way == "pinned":
            pc = pinned(c)
            keep_alive.append(pc)                             # the caller keeps page-locked clouds alive until the pipeline has synchronized
            pipe.ingest_async(pc)
        elif way == "file":
            path = str(tmp_path / f"c{k}.pcrp")
            pcr.write_point_cloud(path, host)
            chunk = int(rng.integers(max(1, n // 9), n + 2))
            assert pipe.ingest_file(path, chunk_points=chunk) == n, f"seed {seed}: ingest_file({way}, chunk {chunk}) of {n} points"
        else:
            path = str(tmp_path / f"c{k}.pcrp")
            pcr.write_point_cloud(path, host)
            dev = pcr.read_point_cloud(path, pcr.PointCloudFormat.Auto, pcr.MemoryLocation.Device)
            assert dev.location() == pcr.MemoryLocation.Device and dev.count() == n
            pipe.ingest(dev)
    pipe.finalize()
    desc = f"seed {seed} (ways={ways}, result on {'device' if on_device else 'host'})"
    assert pipe.stats().points_processed == plain.stats().points_processed, f"{desc}: points_processed"
    res = pipe.result().to_host() if on_device else pipe.result()
    got = [np.array(res.band_array(i)) for i in range(len(specs))]
    F.assert_bands_match(desc, og, specs, filt, clouds, got, want)
    w, h, nb, _crs, bounds = pcr.read_geotiff_info(out)
    assert (w, h, nb) == (og.width, og.height, len(specs)), desc
    assert (bounds.min_x, bounds.min_y, bounds.max_x, bounds.max_y) == (og.min_x, og.min_y, og.max_x, og.max_y), desc
    for i in range(len(specs)):
        assert np.array_equal(np.array(pcr.read_geotiff_band(out, i)), got[i], equal_nan=True), f"{desc}: GeoTIFF band {i}"
    assert pcr.read_geotiff_band_names(out) == [f"band{i}" for i in range(len(specs))], desc
