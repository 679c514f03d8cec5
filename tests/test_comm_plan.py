"""The agreement step of the native halo reduce (pcr_hip_comm_halo_plan, include/pcr_hip.h) on CPU.

pcr_hip_comm_halo_reduce first all-gathers what every rank brings and then judges the gathered records with this pure
function; what is tested here is that the judgement is the SAME on every rank -- every rank posts matching sends and
receives, or every rank refuses -- for the geometries that used to leave one rank returning alone while its neighbours
waited in ncclRecv (blocks that differ by one unit around the halo, an empty last block, asymmetric windows).
The reference is single-device (include/pcr/engine/pipeline.h:68): no reference test to mirror.
"""
import ctypes as C

import pytest

from conftest import load_cabi


def geoms(A, blocks, H, halo, width=64, nplanes=2, kinds=0x21, edit=None):
    """Records of a row-block sharding as the pipeline sizes it: state window = block +- halo, clipped to the grid."""
    arr = (A.HaloGeom * len(blocks))()
    for r, (b0, b1) in enumerate(blocks):
        s0, s1 = max(0, b0 - halo), min(H, b1 + halo)
        g = arr[r]
        g.width, g.state_row0, g.state_rows, g.own_row0, g.own_row1 = width, s0, s1 - s0, b0, b1
        g.halo, g.nplanes, g.kinds, g.valid = halo, nplanes, kinds, 1
    if edit:
        edit(arr)
    return arr


def plan_all(A, arr):
    """(rc, message, (send_up, send_dn, recv_up, recv_dn)) per rank."""
    L = A.lib()
    out = []
    for r in range(len(arr)):
        v = [C.c_int(-1) for _ in range(4)]
        rc = L.pcr_hip_comm_halo_plan(arr, len(arr), r, *[C.byref(x) for x in v])
        out.append((rc, L.pcr_hip_last_error().decode(), tuple(x.value for x in v)))
    return out


def row_block(rank, world, height, align=1):
    units = (height + align - 1) // align
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, height), min(u1 * align, height)


def test_balanced_blocks_every_send_has_its_receive():
    A = load_cabi()
    for world, H, halo in ((2, 64, 4), (3, 100, 7), (8, 16384, 4), (5, 103, 20)):
        blocks = [row_block(r, world, H) for r in range(world)]
        res = plan_all(A, geoms(A, blocks, H, halo))
        assert all(rc == 0 for rc, _, _ in res), res
        for r in range(world):
            su, sd, ru, rd = res[r][2]
            assert su == (min(halo, blocks[r][0]) if r > 0 else 0)
            assert sd == (min(halo, H - blocks[r][1]) if r < world - 1 else 0)
            if r > 0:
                assert ru == res[r - 1][2][1] and su == res[r - 1][2][3]          # my receive = its send, my send = its receive
            if r < world - 1:
                assert rd == res[r + 1][2][0] and sd == res[r + 1][2][2]


def test_blocks_that_differ_by_one_row_around_the_halo_are_refused_by_every_rank():
    """H = 1005 over 100 ranks: blocks of 11 and 10 rows, halo 11.  The 10-row ranks used to return alone."""
    A = load_cabi()
    world, H, halo = 100, 1005, 11
    blocks = [row_block(r, world, H) for r in range(world)]
    assert {b1 - b0 for b0, b1 in blocks} == {10, 11}
    res = plan_all(A, geoms(A, blocks, H, halo))
    assert all(rc == 1 for rc, _, _ in res)
    assert len({msg for _, msg, _ in res}) == 1 and "shorter than" in res[0][1]
    # one row less of halo and everybody goes ahead
    assert all(rc == 0 for rc, _, _ in plan_all(A, geoms(A, blocks, H, 10)))


def test_an_empty_last_block_is_refused_by_every_rank():
    A = load_cabi()
    world, H = 4, 96
    blocks = [row_block(r, world, H, align=32) for r in range(world)]          # 3 units of 32 rows over 4 ranks
    assert blocks[-1][0] == blocks[-1][1]
    res = plan_all(A, geoms(A, blocks, H, 4))
    assert all(rc == 1 for rc, _, _ in res) and all("owns no rows" in msg for _, msg, _ in res)


def test_mismatched_windows_both_neighbours_refuse():
    """Rank 1 keeps a taller state window than its halo allows / a different halo / width: BOTH ranks of the edge (and
    every other rank) must refuse -- the old code posted a receive of min(halo, rows) on one side and a send of a
    different size on the other."""
    A = load_cabi()
    blocks = [(0, 32), (32, 64)]

    def taller(arr):
        arr[1].state_row0, arr[1].state_rows = 32 - 9, 32 + 9                 # 9 apron rows with a halo of 4

    def other_halo(arr):
        arr[1].halo = 5

    def other_width(arr):
        arr[0].width = 128

    def other_planes(arr):
        arr[1].nplanes, arr[1].kinds = 1, 0x1

    def gap(arr):
        arr[1].own_row0 = 33

    def poisoned(arr):
        arr[0].valid = 0

    for edit, word in ((taller, "more than the halo"), (other_halo, "where rank 0 brings"), (other_width, "where rank 0 brings"),
                       (other_planes, "where rank 0 brings"), (gap, "not contiguous"), (poisoned, "invalid arguments")):
        res = plan_all(A, geoms(A, blocks, 64, 4, edit=edit))
        assert [rc for rc, _, _ in res] == [1, 1], (edit.__name__, res)
        assert all(word in msg for _, msg, _ in res), (edit.__name__, res)
        assert all(v == (0, 0, 0, 0) for _, _, v in res)


def test_asymmetric_but_consistent_windows_are_planned_from_what_the_sender_holds():
    """A sender whose apron is SHORTER than the halo (its window is clipped) is fine: the receive is sized from the sender's
    record, not from min(halo, rows)."""
    A = load_cabi()

    def clipped(arr):
        arr[0].state_rows = 32 + 2                 # rank 0 keeps only 2 of the 4 rows below its block

    res = plan_all(A, geoms(A, [(0, 32), (32, 64)], 64, 4, edit=clipped))
    assert [rc for rc, _, _ in res] == [0, 0]
    assert res[0][2] == (0, 2, 0, 4) and res[1][2] == (4, 0, 2, 0)


def test_nothing_to_move_is_agreed_not_assumed():
    A = load_cabi()
    blocks = [(0, 32), (32, 64)]
    for halo, nplanes in ((0, 2), (4, 0)):
        res = plan_all(A, geoms(A, blocks, 64, halo, nplanes=nplanes, kinds=0x21 if nplanes else 0))
        assert all(rc == 0 and v == (0, 0, 0, 0) for rc, _, v in res)


def test_argument_checks():
    A = load_cabi()
    L = A.lib()
    arr = geoms(A, [(0, 32), (32, 64)], 64, 4)
    v = [C.c_int(0) for _ in range(4)]
    assert L.pcr_hip_comm_halo_plan(arr, 2, 2, *[C.byref(x) for x in v]) == 1
    assert L.pcr_hip_comm_halo_plan(None, 2, 0, *[C.byref(x) for x in v]) == 1
    word = C.c_int32(3)
    assert L.pcr_hip_comm_agree_max_i32(None, C.byref(word), None) == 1


def test_two_copies_of_rccl_in_one_process_are_refused(tmp_path):
    """Round 3 ended a test process with `double free or corruption` because a second RCCL (ROCm's) had been loaded beside
    torch's bundled one (each brings its own rocm_smi / roctx statics).  The library now looks at the process's mapping
    table before it resolves RCCL: two copies -> pcr_hip_comm_available() == 0 and every comm entry point says why.
    Run in a child process that leaves with os._exit (the two copies must not run their exit handlers)."""
    import os
    import subprocess
    import sys
    import torch
    a = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    b = "/opt/rocm/lib/librccl.so.1"
    if not (os.path.exists(a) and os.path.exists(b)) or os.path.realpath(a) == os.path.realpath(b):
        pytest.skip("needs torch's bundled RCCL and ROCm's as two different files")
    A = load_cabi()
    code = f"""
import ctypes, os, sys
ctypes.CDLL({a!r}, mode=ctypes.RTLD_GLOBAL)
ctypes.CDLL({b!r}, mode=ctypes.RTLD_GLOBAL)
L = ctypes.CDLL({A.LIB_PATH!r})
L.pcr_hip_last_error.restype = ctypes.c_char_p
ok = L.pcr_hip_comm_available()
buf = (ctypes.c_uint8 * 128)()
rc = L.pcr_hip_comm_unique_id(buf)
sys.stdout.write(f"{{ok}} {{rc}} {{L.pcr_hip_last_error().decode()}}\\n")
sys.stdout.flush()
os._exit(0)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    ok, rc, msg = out.stdout.strip().split(" ", 2)
    assert ok == "0" and rc != "0" and "two copies of RCCL" in msg, out.stdout


# ---- the variable-size transfers (pcr_hip_comm_alltoallv / _gatherv): the same discipline, the same kind of test ----

def xfer_records(A, matrix, elem=(8, 8, 4), root=-1, capacity=None, edit=None):
    """Records of `world` ranks for the send-count matrix[src][dst]; capacity defaults to exactly what each rank receives."""
    world = len(matrix)
    arr = (A.XferGeom * world)()
    for r in range(world):
        g = arr[r]
        g.narrays, g.root, g.valid = len(elem), root, 1
        for a, e in enumerate(elem):
            g.elem_bytes[a] = e
        for p in range(world):
            g.send_counts[p] = matrix[r][p]
        g.recv_capacity = sum(matrix[p][r] for p in range(world)) if capacity is None else capacity[r]
    if edit:
        edit(arr)
    return arr


def xfer_plan_all(A, arr):
    L = A.lib()
    world = len(arr)
    out = []
    for r in range(world):
        so, rc_, ro = (C.c_uint64 * (world + 1))(), (C.c_uint64 * world)(), (C.c_uint64 * (world + 1))()
        rc = L.pcr_hip_comm_xfer_plan(arr, world, r, so, rc_, ro)
        out.append((rc, L.pcr_hip_last_error().decode(), list(so), list(rc_), list(ro)))
    return out


def test_alltoallv_plan_every_send_has_its_receive():
    A = load_cabi()
    import random
    rnd = random.Random(5)
    for world in (1, 2, 3, 8):
        m = [[rnd.randrange(0, 1000) if rnd.random() < 0.8 else 0 for _ in range(world)] for _ in range(world)]
        res = xfer_plan_all(A, xfer_records(A, m))
        assert all(rc == 0 for rc, *_ in res), res
        for r in range(world):
            _, _, so, rcnt, ro = res[r]
            assert rcnt == [m[p][r] for p in range(world)]                       # what I receive from p is what p sends me
            assert so == [sum(m[r][:p]) for p in range(world + 1)]               # my groups lie in rank order
            assert ro == [sum(rcnt[:p]) for p in range(world + 1)]


def test_alltoallv_plan_refuses_on_every_rank_or_on_none():
    A = load_cabi()
    m = [[5, 7, 0], [1, 0, 9], [4, 4, 4]]
    cases = {
        "a rank without room": dict(capacity=[10, 11, 12]),                       # rank 2 receives 13
        "element sizes differ": dict(edit=lambda a: a[1].elem_bytes.__setitem__(2, 8)),
        "array counts differ": dict(edit=lambda a: setattr(a[2], "narrays", 2)),
        "one rank gathers, the others exchange": dict(edit=lambda a: setattr(a[0], "root", 0)),
        "a rank with invalid arguments": dict(edit=lambda a: setattr(a[1], "valid", 0)),
    }
    for what, kw in cases.items():
        res = xfer_plan_all(A, xfer_records(A, m, **kw))
        assert all(rc == 1 for rc, *_ in res), (what, res)
        assert len({msg for _, msg, *_ in res}) == 1, (what, res)               # the same message everywhere


def test_gatherv_plan_strips_land_in_rank_order_on_the_root():
    A = load_cabi()
    world, root = 4, 2
    strips = [2048 * 64, 2048 * 64, 2000 * 64, 100 * 64]
    m = [[strips[r] if p == root else 0 for p in range(world)] for r in range(world)]
    res = xfer_plan_all(A, xfer_records(A, m, elem=(4, 4), root=root, capacity=[0, 0, sum(strips), 0]))
    assert all(rc == 0 for rc, *_ in res), res
    assert res[root][3] == strips and res[root][4] == [sum(strips[:p]) for p in range(world + 1)]
    for r in range(world):
        if r != root:
            assert res[r][3] == [0] * world
    # somebody sends to a rank that is not the root / the root has too little room: everyone refuses
    bad = xfer_plan_all(A, xfer_records(A, m, elem=(4, 4), root=root, capacity=[0, 0, sum(strips), 0],
                                        edit=lambda a: a[1].send_counts.__setitem__(0, 3)))
    assert all(rc == 1 and "not the root" in msg for rc, msg, *_ in bad), bad
    small = xfer_plan_all(A, xfer_records(A, m, elem=(4, 4), root=root, capacity=[0, 0, sum(strips) - 1, 0]))
    assert all(rc == 1 and "has room for" in msg for rc, msg, *_ in small), small
