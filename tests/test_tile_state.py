"""`.pcrt` tile-state checkpoints: byte-compatible with the reference's
src/io/tile_state_io.cpp in both directions, and Pipeline.save_state / load_state / resume."""
import ctypes as C
import os
import struct
import sys

import numpy as np
import pytest

import pcr
import pcr_oracle_py as O
from conftest import GOLDEN, assert_band_close

sys.path.insert(0, GOLDEN)
from make_ref_tile_state import fixture_state   # noqa: E402

FIXTURE = os.path.join(GOLDEN, "ref_tile_0001_0002.pcrt")     # written by the reference's own writer


def test_reader_reads_reference_written_file():
    row, col, state, rtype = pcr.read_tile_state(FIXTURE)
    assert (row, col) == (1, 2) and rtype == pcr.ReductionType.Average
    assert state.shape == (2, 3, 5) and np.array_equal(state, fixture_state())


def test_writer_is_byte_identical_to_reference(tmp_path):
    p = str(tmp_path / "mine.pcrt")
    pcr.write_tile_state(p, 1, 2, fixture_state(), pcr.ReductionType.Average)
    mine, ref = open(p, "rb").read(), open(FIXTURE, "rb").read()
    assert mine == ref
    # header layout of include/pcr/io/tile_state_io.h:10-25
    magic, ver, trow, tcol, cols, rows, k, red = struct.unpack("<4sIiiiiiB", mine[:29])
    assert (magic, ver, trow, tcol, cols, rows, k, red) == (b"PCRT", 1, 1, 2, 5, 3, 2, 3)
    assert mine[29:36] == b"\0" * 7 and len(mine) == 36 + 2 * 3 * 5 * 4


def test_filename_and_errors(tmp_path):
    assert pcr.tile_state_filename("/tmp/x", 3, 12) == "/tmp/x/tile_0003_0012.pcrt"
    assert pcr.tile_state_filename("/tmp/x/", 0, 0) == "/tmp/x/tile_0000_0000.pcrt"
    with pytest.raises(RuntimeError, match="file not found"):
        pcr.read_tile_state(str(tmp_path / "nope.pcrt"))
    bad = tmp_path / "bad.pcrt"
    bad.write_bytes(b"NOPE" + b"\0" * 60)
    with pytest.raises(RuntimeError, match="invalid magic number"):
        pcr.read_tile_state(str(bad))
    trunc = tmp_path / "trunc.pcrt"
    trunc.write_bytes(open(FIXTURE, "rb").read()[:80])
    with pytest.raises(RuntimeError, match="incomplete state data"):
        pcr.read_tile_state(str(trunc))
    v2 = bytearray(open(FIXTURE, "rb").read())
    v2[4] = 2
    (tmp_path / "v2.pcrt").write_bytes(bytes(v2))
    with pytest.raises(RuntimeError, match="unsupported version 2"):
        pcr.read_tile_state(str(tmp_path / "v2.pcrt"))


@pytest.mark.skipif(O.ref_lib() is None, reason="oracle/_ref not built (reference tree absent)")
def test_live_reference_round_trips(tmp_path):
    R = O.ref_lib()
    R.pcr_ref_write_tile_state.argtypes = [C.c_char_p] + [C.c_int] * 6 + [C.c_void_p]
    R.pcr_ref_read_tile_state.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 6 + [C.c_void_p]
    rng = np.random.default_rng(2)
    st = rng.normal(size=(1, 7, 9)).astype(np.float32)
    mine = str(tmp_path / "mine.pcrt")
    pcr.write_tile_state(mine, 4, 5, st, pcr.ReductionType.Max)
    v = [C.c_int(0) for _ in range(6)]
    out = np.zeros_like(st)
    assert R.pcr_ref_read_tile_state(mine.encode(), *[C.byref(a) for a in v], out.ctypes.data) == 0
    assert [a.value for a in v] == [4, 5, 9, 7, 1, 1] and np.array_equal(out, st)      # the reference reads ours
    theirs = str(tmp_path / "theirs.pcrt")
    assert R.pcr_ref_write_tile_state(theirs.encode(), 4, 5, 9, 7, 1, 1, st.ctypes.data) == 0
    assert open(theirs, "rb").read() == open(mine, "rb").read()                       # and writes the same bytes
    buf = C.create_string_buffer(256)
    R.pcr_ref_tile_state_filename(b"/a/b", 12, 345, buf, 256)
    assert buf.value.decode() == pcr.tile_state_filename("/a/b", 12, 345)


def _cfg(reductions, state_dir="", resume=False, tile=5, G=10):
    cfg = pcr.PipelineConfig()
    cfg.grid.bounds = pcr.BBox(0.0, 0.0, float(G), float(G))
    cfg.grid.tile_width = cfg.grid.tile_height = tile
    cfg.grid.compute_dimensions()
    cfg.exec_mode = pcr.ExecutionMode.GPU
    cfg.reductions = reductions
    cfg.state_dir = state_dir
    cfg.resume = resume
    return cfg


def _spec(t):
    r = pcr.ReductionSpec()
    r.value_channel, r.type = "value", t
    return r


def _cloud(x, y, v):
    c = pcr.PointCloud.create(len(x))
    c.set_x_array(np.asarray(x, dtype=np.float64))
    c.set_y_array(np.asarray(y, dtype=np.float64))
    c.add_channel("value", pcr.DataType.Float32)
    c.set_channel_array_f32("value", np.asarray(v, dtype=np.float32))
    return c


@pytest.mark.gpu
def test_pipeline_checkpoint_resume_equals_uninterrupted(tmp_path):
    T = pcr.ReductionType
    rng = np.random.default_rng(8)
    G = 64
    x1, y1, v1 = rng.uniform(0, 30, 4000), rng.uniform(0, G, 4000), rng.uniform(0, 1, 4000)   # left tiles only
    x2, y2, v2 = rng.uniform(0, G, 6000), rng.uniform(0, G, 6000), rng.uniform(0, 1, 6000)
    reds = lambda: [_spec(T.Average), _spec(T.Count), _spec(T.Max)]     # noqa: E731
    d = str(tmp_path / "ckpt")
    a = pcr.Pipeline.create(_cfg(reds(), tile=16, G=G))
    a.ingest(_cloud(x1, y1, v1))
    a.save_state(d)
    # only touched tiles have files (columns 0-1 of the 4x4 tiling), one directory per reduction
    names = sorted(os.listdir(os.path.join(d, "reduction_0")))
    assert len(names) == 8 and names[0] == "tile_0000_0000.pcrt" and all(n[10:14] in ("0000", "0001") for n in names)
    row, col, st, rt = pcr.read_tile_state(os.path.join(d, "reduction_0", "tile_0002_0001.pcrt"))
    assert st.shape == (2, 16, 16) and rt == T.Average and (row, col) == (2, 1)
    b = pcr.Pipeline.create(_cfg(reds(), state_dir=d, resume=True, tile=16, G=G))     # resume at create
    b.ingest(_cloud(x2, y2, v2))
    b.finalize()
    c = pcr.Pipeline.create(_cfg(reds(), tile=16, G=G))
    c.ingest(_cloud(x1, y1, v1))
    c.ingest(_cloud(x2, y2, v2))
    c.finalize()
    for band, (rt_, at_) in enumerate([(1e-6, 1e-7), (0, 0), (0, 0)]):
        assert_band_close(np.array(b.result().band_array(band)), np.array(c.result().band_array(band)),
                          rtol=rt_, atol=at_, what=f"band {band}")
    # resume without further ingest: untouched tiles stay NaN, touched ones come back
    e = pcr.Pipeline.create(_cfg(reds(), tile=16, G=G))
    e.load_state(d)
    e.finalize()
    cnt = np.array(e.result().band_array(1))
    assert np.nansum(cnt) == 4000 and np.isnan(cnt[:, 32:]).all()


@pytest.mark.gpu
def test_pipeline_loads_reference_written_tile(tmp_path):
    """A state file produced by the reference's writer for tile (1, 0) of a 10x10 / 5x5 grid."""
    d = tmp_path / "ref_state"
    d.mkdir()
    st = np.zeros((2, 5, 5), dtype=np.float32)
    st[0] = np.arange(25, dtype=np.float32).reshape(5, 5) * 3.0       # sums
    st[1] = 3.0                                                       # counts
    st[1, 4, 4] = 0.0                                                 # one empty cell
    R = O.ref_lib()
    path = pcr.tile_state_filename(str(d), 1, 0)
    if R is not None:
        R.pcr_ref_write_tile_state.argtypes = [C.c_char_p] + [C.c_int] * 6 + [C.c_void_p]
        assert R.pcr_ref_write_tile_state(path.encode(), 1, 0, 5, 5, 2, 3, st.ctypes.data) == 0
    else:                                      # GPU box: same bytes from our byte-identical writer
        pcr.write_tile_state(path, 1, 0, st, pcr.ReductionType.Average)
    p = pcr.Pipeline.create(_cfg([_spec(pcr.ReductionType.Average)], state_dir=str(d), resume=True))
    p.finalize()
    band = np.array(p.result().band_array(0))
    want = np.full((10, 10), np.nan, dtype=np.float32)
    want[5:10, 0:5] = np.arange(25, dtype=np.float32).reshape(5, 5)
    want[9, 4] = np.nan
    assert_band_close(band, want, what="reference-written tile")
    # a file for another reduction type in the same place is ignored, not misread
    pcr.write_tile_state(pcr.tile_state_filename(str(d), 0, 0), 0, 0, np.ones((1, 5, 5), np.float32), pcr.ReductionType.Sum)
    q = pcr.Pipeline.create(_cfg([_spec(pcr.ReductionType.Average)], state_dir=str(d), resume=True))
    q.finalize()
    assert np.isnan(np.array(q.result().band_array(0))[0:5, 0:5]).all()


@pytest.mark.parametrize("engine", ["host", pytest.param("hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("damage", ["truncated", "bad_magic", "wrong_shape", "empty"])
def test_damaged_tile_files_are_treated_as_absent_on_resume(tmp_path, damage, engine):
    """The reference's tile manager re-initialises a tile whose file does not read back (corrupt, truncated, of another shape:
    src/engine/tile_manager.cpp:272-320) instead of failing the run; so does resume here -- the damaged tile starts from
    identity, its neighbours come back.  On both engines (one load code: host/src/pipeline_common.cpp)."""
    d = str(tmp_path)
    good = np.arange(50, dtype=np.float32).reshape(2, 5, 5) + 1.0            # {sum, count} of an Average tile, all cells counted
    pcr.write_tile_state(pcr.tile_state_filename(d, 0, 0), 0, 0, good, pcr.ReductionType.Average)
    bad = pcr.tile_state_filename(d, 1, 1)
    pcr.write_tile_state(bad, 1, 1, good, pcr.ReductionType.Average)
    data = open(bad, "rb").read()
    if damage == "truncated":
        open(bad, "wb").write(data[:60])
    elif damage == "bad_magic":
        open(bad, "wb").write(b"XXXX" + data[4:])
    elif damage == "wrong_shape":
        pcr.write_tile_state(bad, 1, 1, np.ones((2, 4, 5), np.float32), pcr.ReductionType.Average)     # a 5 x 4 tile where the grid has 5 x 5
    else:
        open(bad, "wb").close()
    cfg = _cfg([_spec(pcr.ReductionType.Average)], state_dir=d, resume=True)
    cfg.exec_mode = pcr.ExecutionMode.CPU if engine == "host" else pcr.ExecutionMode.GPU
    p = pcr.Pipeline.create(cfg)
    assert p is not None, pcr.pipeline_create_error()
    assert p.engine() == engine
    p.ingest(_cloud([7.5], [2.5], [3.0]))                                     # one point into the damaged tile: cell (row 7, col 7)
    p.finalize()
    band = np.array(p.result().band_array(0))
    assert np.allclose(band[0:5, 0:5], good[0] / good[1])                     # the intact tile came back
    want = np.full((5, 5), np.nan, np.float32)
    want[2, 2] = 3.0
    assert np.array_equal(band[5:10, 5:10], want, equal_nan=True)             # the damaged one started from identity
