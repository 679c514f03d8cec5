"""The N > 1 default of bench.py IS the north-star configuration (BASELINE.json configs[4]: fixed grid, strong scaling,
row blocks that cut the reference tiles, live halo exchange): rehearsed here with two ranks on one GPU over gloo on a
down-scaled grid -- the same code path the driver launches with RCCL on 8 GPUs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, launcher=True, ranks=2, extra_env=None):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--backend", "gloo", "--same-device"] + extra
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail          # the driver's form: bench.py starts its own ranks as a child process
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_default_multi_gpu_bench_is_c5_strong_scaling_with_live_exchange():
    r = _run(["--grid", "6144", "--points", "3000000", "--unrouted"])      # 2 x 3072 rows: cuts the 4096-row tiles
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    c = r["config"]
    assert c["workload"].startswith("C5_point") and c["grid"] == "6144x6144" and c["rows_per_gpu"] == 3072
    assert c["points_total"] == 6_000_000 and c["tiles_local"] is False
    assert c["collectives_per_step"] == {"p2p_messages": 0, "all_reduces": 1}     # Point: touched-tile union only
    assert c["exchange_ms"] > 0
    g1 = r["per_glyph"]["gauss1"]
    assert g1["exchange"]["halo_rows"] == 4 and g1["exchange"]["collectives_per_step"]["p2p_messages"] == 4
    assert g1["exchange"]["halo_bytes_sent_per_step"] == 2 * 4 * 6144 * 4
    assert g1["Mpts/s"] > 0 and r["value"] > 0
    assert r["unrouted"]["Mpts/s"] > 0
    assert r["one_gpu_same_problem"]["Mpts/s"] > 0 and r["speedup_vs_one_gpu"] > 0
    assert "roofline" in r and r["roofline"]["traffic"] is None or isinstance(r["roofline"]["traffic"], int)
    _check_selfcheck(r)


def _check_selfcheck(r, ranks=2):
    sc = r["selfcheck"]
    assert sc["ok"] is True, sc
    assert sc["points_valid_all_ranks"] == sc["points_total"] == r["config"]["points_total"]
    assert sc["gauss_planes"] == 2 and sc["gauss_max_rel_diff"] <= 2e-6
    assert all(v > 0 for v in sc["gauss_halo_rows_sum_before_exchange"])        # the halo rows really carried weight
    # finalized again, then ingested again: nothing counted twice
    assert sc["refinalize_owned_rows_unchanged"] is True and all(v == 0 for v in sc["refinalize_halo_rows_sum"])
    assert sc["second_ingest_max_rel_diff_vs_twice"] <= 4e-6
    assert r["config"]["world_size"] == ranks and r["config"]["backend"] == "gloo"


def test_plain_python_bench_gpus_2_launches_its_own_ranks_and_checks_the_exchange():
    """`python bench.py --gpus 2 ...` with no launcher around it (the driver's form)."""
    r = _run(["--grid", "6144", "--points", "3000000"], launcher=False)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["value"] > 0
    _check_selfcheck(r)


def test_weak_flag_keeps_round_one_shape_without_collectives():
    r = _run(["--weak", "--grid", "1024", "--points", "1000000"])
    assert r["scaling"] == "weak" and r["config"]["grid"] == "1024x2048"
    assert r["config"]["tiles_local"] is False or r["config"]["collectives_per_step"]["p2p_messages"] == 0


def test_the_eight_rank_geometry_with_the_four_ranks_a_one_gpu_box_admits():
    """BASELINE configs[4] at N = 8: 16384 columns, 2048-row blocks that cut the 4096-row reference tiles, halo 4, inner ranks
    with two neighbours.  A one-GPU box admits SIX processes on its card, this test runner is one of them and a process that
    is still going away counts too, so the same blocks are instantiated as FOUR ranks on a 16384 x 8192 grid (blocks 0|1 and
    2|3 share a tile row and exchange; 1|2 meet on a tile boundary -- exactly the alternation the 8-rank run has; ranks 1 and
    2 have two neighbours), through the driver's launcher path: the run must validate itself."""
    r = _run(["--grid", "16384", "--height", "8192", "--points", "1500000"], launcher=False, ranks=4)
    assert r["n_gpus"] == 4 and r["scaling"] == "strong"
    c = r["config"]
    assert c["grid"] == "16384x8192" and c["rows_per_gpu"] == 2048 and c["tiles_local"] is False
    assert c["collectives_per_step"] == {"p2p_messages": 0, "all_reduces": 1}
    g1 = r["per_glyph"]["gauss1"]
    assert g1["exchange"]["halo_rows"] == 4
    assert g1["exchange"]["collectives_per_step"]["p2p_messages"] == 4          # rank 0: one neighbour, two planes, send + receive
    assert g1["exchange"]["halo_bytes_sent_per_step"] == 2 * 4 * 16384 * 4
    _check_selfcheck(r, ranks=4)


def test_bench_times_the_native_exchange_and_compares_the_two_transports_on_the_test_double(tmp_path):
    """`bench.py --comm native` with three ranks on one GPU: the library's own exchange is the TIMED transport, the selfcheck
    runs through it, and `native_exchange` compares it with the torch transport bit for bit -- the part of the driver's N > 1
    run that needs more than one rank, on the test double for RCCL (tests/native/fake_rccl.cpp; real RCCL admits one rank per
    device).  2 048-row blocks on a 6 144-row grid cut the 4 096-row tiles: every step exchanges."""
    import shutil
    gxx = shutil.which("g++")
    if not gxx or not os.path.isdir("/opt/rocm/include/rccl"):
        pytest.skip("g++ or the RCCL headers are not available")
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.run([gxx, "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"), "-o", fake, "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    r = _run(["--grid", "6144", "--points", "2000000", "--comm", "native"], ranks=3,
             extra_env=dict(PCR_HIP_RCCL=fake, PCR_FAKE_RCCL_DIR=str(tmp_path)))
    assert r["n_gpus"] == 3 and r["config"]["rows_per_gpu"] == 2048 and r["config"]["tiles_local"] is False
    sc = r["selfcheck"]
    assert sc["ok"] is True and sc["comm"] == "native", sc
    assert sc["refinalize_owned_rows_unchanged"] is True and all(v == 0 for v in sc["refinalize_halo_rows_sum"])
    ne = r["native_exchange"]
    assert ne["ok"] is True and ne["ranks_bit_identical"] == 3, ne
    g1 = r["per_glyph"]["gauss1"]
    assert g1["exchange"]["comm"] == "native" and g1["exchange"]["halo_rows"] == 4
