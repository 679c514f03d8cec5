"""Randomised differential test of the row-block SHARDED pipelines against one unsharded pipeline of this library: the grids,
reductions, filters and clouds of tests/test_gpu_out_of_core_fuzz.py, cut into two or three row blocks that do or do not fall
on reference-tile rows; every round a rank is handed either the WHOLE cloud (it keeps the points whose centre row it owns) or
an arbitrary third of it (ingest_unrouted: partition on the device, all-to-all to the owners); finalize after some rounds, twice
after some (state survives finalize, src/engine/pipeline.cpp:1344-1364: the exchange must be idempotent); half of the time a
checkpoint in the middle, resumed by NEW shards; at the end the strips are gathered on a random rank.  Two transports:

  native   pcr::ShardedPipeline (C++) over the library's own pcr_hip_comm_* -- three ranks on one GPU on the test double for RCCL
           (tests/native/fake_rccl.cpp; real RCCL admits one rank per device), ranks that never import torch;
  torch    pcr.distributed.ShardedPipeline over torch.distributed / gloo, two ranks.

Every rank's strip of every finalize, stacked, must equal the unsharded pipeline's bands after the same rounds (bit for bit for
Max / Min / Count of Points and Lines, fp32 re-association for the sums: assert_bands_match).  Several seeds per launch of the
ranks; PCR_SHARD_FUZZ_SEEDS=a:b soaks a range."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def plan(seed, world):
    """What the ranks do with seed's clouds, the same on every rank and in the parent."""
    import test_gpu_out_of_core_fuzz as F
    og, specs, filt, clouds, _ = F.build(seed)
    rng = np.random.default_rng(99000 + seed)
    rounds = [dict(mode=str(rng.choice(["whole", "unrouted"])), finalize=int(rng.choice([0, 1, 1, 2]))) for _ in clouds]
    rounds[-1]["finalize"] = max(1, rounds[-1]["finalize"])
    ck = int(rng.integers(0, len(clouds))) if rng.uniform() < 0.5 else None
    return og, specs, filt, clouds, rounds, ck, int(rng.integers(0, world))


def file_barrier(out_dir, tag, rank, world):
    import time
    open(os.path.join(out_dir, f"bar_{tag}_{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(out_dir, f"bar_{tag}_{r}")) for r in range(world)):
        assert time.time() - t0 < 300, f"barrier {tag}: a rank is missing"
        time.sleep(0.002)


def rank_main(kind, rank, world, out_dir, seed_lo, seed_hi, port):
    """One rank of the launch: every seed in [seed_lo, seed_hi) on fresh pipelines."""
    sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
    if kind == "torch":
        import torch                                   # before pcr: one shared HIP runtime
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import pcr
    import test_gpu_out_of_core_fuzz as F
    if kind == "native":
        assert "torch" not in sys.modules
    made = [0]

    def create(cfg):
        cfg.cuda_device_id = 0
        if kind == "native":
            # a communicator of its own id for every pipeline, as RCCL requires (rank 0 makes it, a file carries it)
            made[0] += 1
            path = os.path.join(out_dir, f"id{made[0]}.bin")
            if rank == 0:
                with open(path + ".part", "wb") as f:
                    f.write(pcr.NativeShardedPipeline.make_id())
                os.rename(path + ".part", path)
            file_barrier(out_dir, f"id{made[0]}", rank, world)
            sp = pcr.NativeShardedPipeline.create(cfg, open(path, "rb").read(), rank, world, 0)
            assert sp is not None, pcr.NativeShardedPipeline.create_error()
            return sp
        from pcr.distributed import ShardedPipeline
        return ShardedPipeline(cfg, rank, world, device_id=0)

    out = {}
    for seed in range(seed_lo, seed_hi):
        og, specs, filt, clouds, rounds, ck, dst = plan(seed, world)
        sp = create(F.config(og, specs, filt))
        for k, (c, r) in enumerate(zip(clouds, rounds)):
            if r["mode"] == "whole":
                sp.ingest(F.to_cloud(c))
            else:
                sel = np.arange(len(c["x"])) % world == rank
                part = {key: (val[sel] if isinstance(val, np.ndarray) else val) for key, val in c.items()}
                sp.ingest_unrouted(F.to_cloud(part))
            if ck == k:
                ckdir = os.path.join(out_dir, f"ck{seed}")
                sp.save_state(ckdir)
                file_barrier(out_dir, f"ck{seed}", rank, world)      # (cut tiles: rank 0 writes every tile)
                if kind == "torch":
                    sp.close()
                del sp
                sp = create(F.config(og, specs, filt, state_dir=ckdir, resume=True))
            for _ in range(r["finalize"]):
                sp.finalize()
            if r["finalize"]:
                res = sp.result()
                for b in range(len(specs)):
                    out[f"s{seed}k{k}b{b}"] = np.array(res.band_array(b))
        whole = sp.gather(dst)
        assert (whole is not None) == (rank == dst), f"seed {seed}: gather"
        if whole is not None:
            for b in range(len(specs)):
                out[f"s{seed}whole{b}"] = np.array(whole.band_array(b))
        if kind == "torch":
            sp.close()
        del sp
    np.savez(os.path.join(out_dir, f"f{rank}.npz"), **out)
    if kind == "torch":
        dist.destroy_process_group()
    print("rank", rank, "ok")


def launch(tmp_path, kind, world, seed_lo, seed_hi):
    env = dict(os.environ, PCR_REQUIRE_GPU_ENGINE="1")
    port = 0
    if kind == "native":
        gxx = shutil.which("g++")
        if not gxx or not os.path.isdir("/opt/rocm/include/rccl"):
            pytest.skip("g++ or the RCCL headers are not available")
        fake = str(tmp_path / "libfake_rccl.so")
        subprocess.run([gxx, "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        os.path.join(HERE, "native", "fake_rccl.cpp"), "-o", fake, "-L/opt/rocm/lib", "-lamdhip64"], check=True)
        env.update(PCR_HIP_RCCL=fake, PCR_FAKE_RCCL_DIR=str(tmp_path))
    else:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    code = (f"import sys; sys.path.insert(0, {HERE!r}); sys.path.insert(0, {os.path.join(ROOT, 'oracle')!r}); import test_gpu_sharded_fuzz as T; "
            f"T.rank_main({kind!r}, int(sys.argv[1]), {world}, {str(tmp_path)!r}, {seed_lo}, {seed_hi}, {port})")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"{kind}: a rank hung (seeds {seed_lo}..{seed_hi - 1})")
        outs.append((p.returncode, so, se))
    for r, (rc, so, se) in enumerate(outs):
        assert rc == 0, f"{kind} rank {r} (seeds {seed_lo}..{seed_hi - 1}): " + so[-1500:] + se[-3000:]


def verify(tmp_path, kind, world, seed_lo, seed_hi):
    import pcr
    import test_gpu_out_of_core_fuzz as F
    parts = [np.load(tmp_path / f"f{r}.npz") for r in range(world)]
    for seed in range(seed_lo, seed_hi):
        og, specs, filt, clouds, rounds, ck, dst = plan(seed, world)
        one = pcr.Pipeline.create(F.config(og, specs, filt))
        assert one is not None, pcr.pipeline_create_error()
        last = None
        for k, (c, r) in enumerate(zip(clouds, rounds)):
            one.ingest(F.to_cloud(c))
            if not r["finalize"]:
                continue
            one.finalize()
            want = [np.array(one.result().band_array(b)) for b in range(len(specs))]
            got = [np.vstack([p[f"s{seed}k{k}b{b}"] for p in parts]) for b in range(len(specs))]
            F.assert_bands_match(f"{kind} x{world}, seed {seed} (rounds={rounds}, checkpoint after {ck}), finalize after round {k}:",
                                 og, specs, filt, clouds[:k + 1], got, want)
            last = got
        for b in range(len(specs)):                             # the gathered grid IS the stacked strips of the last finalize
            assert np.array_equal(parts[dst][f"s{seed}whole{b}"], last[b], equal_nan=True), f"{kind}, seed {seed}: gather on rank {dst}, band {b}"


def seed_ranges(per_launch, default_hi):
    env = os.environ.get("PCR_SHARD_FUZZ_SEEDS")
    lo, hi = (int(v) for v in env.split(":")) if env else (0, default_hi)
    return [(a, min(a + per_launch, hi)) for a in range(lo, hi, per_launch)]


@pytest.mark.parametrize("seeds", seed_ranges(6, 12), ids=lambda r: f"seeds{r[0]}-{r[1] - 1}")
def test_three_native_shards_on_the_test_double_equal_one_pipeline(tmp_path, seeds):
    launch(tmp_path, "native", 3, *seeds)
    verify(tmp_path, "native", 3, *seeds)


@pytest.mark.parametrize("seeds", seed_ranges(6, 6), ids=lambda r: f"seeds{r[0]}-{r[1] - 1}")
def test_two_torch_shards_over_gloo_equal_one_pipeline(tmp_path, seeds):
    launch(tmp_path, "torch", 2, *seeds)
    verify(tmp_path, "torch", 2, *seeds)
