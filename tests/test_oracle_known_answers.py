"""Pins the CPU oracle against the reference's own known-answer tests
(tests/golden/reference_known_answers.json; every case cites its reference file:line)."""
import ctypes as C

import numpy as np
import pytest

import pcr_oracle_py as O
from conftest import assert_band_close, grid_from_json

RT = {"Sum": O.SUM, "Max": O.MAX, "Min": O.MIN, "Average": O.AVERAGE,
      "WeightedAverage": O.WEIGHTED_AVERAGE, "Count": O.COUNT}


def test_pipeline_known_answers(known_answers, denan):
    for case in known_answers["pipeline"]:
        g = grid_from_json(O, case["grid"])
        for b, red in enumerate(case["reductions"]):
            r = O.Reduction(g, RT[red["type"]])
            for cl in case["clouds"]:
                r.ingest(cl["x"], cl["y"], cl["value"])
            got = r.finalize()
            want = np.array(denan(case["expected"][b]), dtype=np.float32).reshape(g.height, g.width)
            assert_band_close(got, want, what=f'{case["name"]}[{red["type"]}] ({case["source"]})')


def test_world_to_cell(known_answers):
    for c in known_answers["world_to_cell"]:
        g = grid_from_json(O, c["grid"])
        col, row, ok = O.world_to_cell(g, float(c["wx"]), float(c["wy"]))
        assert ok == c["valid"], c["source"]
        if ok:
            assert (col, row) == (c["col"], c["row"]), c["source"]


def test_compute_dimensions(known_answers):
    for c in known_answers["compute_dimensions"]:
        gj = c["grid"]
        g = O.Grid(*[float(b) for b in gj["bounds"]], gj["cell"][0], gj["cell"][1], 0, 0, *gj["tile"])
        tx, ty = C.c_int32(0), C.c_int32(0)
        O.lib().pcro_compute_dimensions(C.byref(g), C.byref(tx), C.byref(ty))
        assert (g.width, g.height, tx.value, ty.value) == (c["width"], c["height"], c["tiles_x"], c["tiles_y"]), c["source"]


def test_tile_cell_range(known_answers):
    for c in known_answers["tile_cell_range"]:
        g = grid_from_json(O, c["grid"])
        assert list(O.tile_cell_range(g, c["tile_row"], c["tile_col"])) == c["expect"], c["source"]


def test_router_assignment(known_answers):
    for c in known_answers["router"]:
        g = grid_from_json(O, c["grid"])
        tiles_x = (g.width + g.tile_width - 1) // g.tile_width
        for i, (x, y) in enumerate(zip(c["x"], c["y"])):
            col, row, ok = O.world_to_cell(g, x, y)
            assert ok
            assert row * g.width + col == c["cell"][i], c["source"]
            assert (row // g.tile_height) * tiles_x + col // g.tile_width == c["tile"][i], c["source"]
            if "local" in c:
                c0, r0, cw, ch = O.tile_cell_range(g, row // g.tile_height, col // g.tile_width)
                assert (row - r0) * cw + (col - c0) == c["local"][i], c["source"]


def _state_run(lib_accum, lib_init, lib_final, rtype, k, case, denan):
    n = case["tile_cells"]
    state = np.zeros(k * n, dtype=np.float32)
    assert lib_init(rtype, state.ctypes.data, n) == 0
    batches = case.get("batches") or [{"cells": case["cells"], "values": case["values"]}]
    for b in batches:
        ci = np.array(b["cells"], dtype=np.uint32)
        v = np.array(b["values"], dtype=np.float32)
        assert lib_accum(rtype, ci.ctypes.data, v.ctypes.data, state.ctypes.data, len(ci), n) == 0
    if "state" in case:
        np.testing.assert_array_equal(state, np.array(case["state"], dtype=np.float32), err_msg=case["source"])
    if "final" in case:
        out = np.zeros(n, dtype=np.float32)
        assert lib_final(rtype, state.ctypes.data, out.ctypes.data, n) == 0
        assert_band_close(out, np.array(denan(case["final"]), dtype=np.float32), what=case["source"])


def test_state_ops(known_answers, denan):
    L = O.lib()
    for case in known_answers["state_ops"]:
        rt = RT[case["type"]]
        _state_run(L.pcro_accumulate, L.pcro_init_state, L.pcro_finalize_state, rt,
                   L.pcro_state_floats(rt), case, denan)


def test_state_ops_out_of_range_index():
    # tests/cpp/test_reduction_ops.cpp:397 -- cell index >= tile_cells is InvalidArgument
    L = O.lib()
    state = np.zeros(4, dtype=np.float32)
    ci = np.array([7], dtype=np.uint32)
    v = np.array([1.0], dtype=np.float32)
    assert L.pcro_accumulate(O.SUM, ci.ctypes.data, v.ctypes.data, state.ctypes.data, 1, 4) == 1
    assert b"out of range" in L.pcro_last_error()


def test_unregistered_reduction_rejected():
    # src/ops/reduction_registry.cpp:173-184: Median/Percentile/MostRecent/... are not registered
    assert O.lib().pcro_state_floats(6) == 0
    g = O.make_grid((0, 0, 4, 4))
    with pytest.raises(O.OracleError):
        O.Reduction(g, 8)


def test_glyph_quirk_probes(known_answers):
    for c in known_answers["glyph"]:
        g = grid_from_json(O, c["grid"])
        s = c["spec"]
        if s["glyph"] == "Gaussian":
            gl = O.make_glyph(O.GLYPH_GAUSSIAN, sigma_x=s["sigma"], sigma_y=s["sigma"], max_radius=s["max_radius"])
        else:
            gl = O.make_glyph(O.GLYPH_LINE, direction=s["direction"], half_length=s["half_length"],
                              max_radius=s["max_radius"])
        band = O.run(g, RT[s["type"]], c["x"], c["y"], c["value"], glyph=gl)
        if "probes" in c:
            for row, col, val in c["probes"]:
                assert band[row, col] == pytest.approx(val, rel=c["rtol"]), (c["name"], row, col)
        if "nan_cols_from" in c:
            assert np.isnan(band[:, c["nan_cols_from"]:]).all(), c["name"]
            assert np.isnan(band[c["nan_rows_from"]:, :]).all(), c["name"]
            assert np.isfinite(band[:c["nan_rows_from"], :c["nan_cols_from"]]).any(), c["name"]
        if "cells_set" in c:
            got = sorted([int(r), int(cc)] for r, cc in np.argwhere(~np.isnan(band)))
            assert got == sorted(c["cells_set"]), c["name"]


def test_glyph_rejects_min_max():
    # pipeline.cpp:500-508 -> NotImplemented
    g = O.make_grid((0, 0, 8, 8))
    r = O.Reduction(g, O.MAX, O.make_glyph(O.GLYPH_GAUSSIAN))
    with pytest.raises(O.OracleError) as e:
        r.ingest([1.0], [1.0], [1.0])
    assert e.value.code == 6 and "glyph splatting only supports" in str(e.value)


def test_empty_cloud_is_noop():
    # pipeline.cpp:284-287
    g = O.make_grid((0, 0, 4, 4))
    r = O.Reduction(g, O.SUM)
    r.ingest(np.zeros(0), np.zeros(0), np.zeros(0, dtype=np.float32))
    assert np.isnan(r.finalize()).all()


def test_router_batches(known_answers):
    """tests/cpp/test_tile_router.cpp:86-200: valid mask, order after the sort, one batch per tile with tile-local
    cell indices -- on the oracle's routing and on the reference-shaped CPU pipeline (oracle/pcr_cpu_pipeline.cpp)."""
    for c in known_answers["router_batches"]:
        g = grid_from_json(O, c["grid"])
        tiles_x = (g.width + g.tile_width - 1) // g.tile_width
        routed = [O.world_to_cell(g, x, y) for x, y in zip(c["x"], c["y"])]
        if "valid_mask" in c:
            assert [int(ok) for _, _, ok in routed] == c["valid_mask"], c["source"]
            continue
        keys = sorted(((row // g.tile_height) * tiles_x + col // g.tile_width, row * g.width + col)
                      for col, row, ok in routed if ok)
        batches = {}
        for tile, cell in keys:                                  # already "tile ascending, cell ascending inside a tile"
            row, col = divmod(cell, g.width)
            c0, r0, cw, ch = O.tile_cell_range(g, row // g.tile_height, col // g.tile_width)
            batches.setdefault(tile, []).append((row - r0) * cw + (col - c0))
        assert len(batches) == c["num_batches"], c["source"]
        for local in batches.values():
            assert len(local) == c["points_per_batch"] and max(local) < c["local_index_below"], c["source"]
            assert local == sorted(local)
        # the CPU pipeline runs those very stages: one point per cell, Count must be 1 everywhere
        v = np.ones(len(c["x"]), dtype=np.float32)
        band, _ = O.cpu_pipeline_run(g, O.COUNT, np.array(c["x"]), np.array(c["y"]), v, threads=1)
        assert (band == 1.0).all(), c["source"]
