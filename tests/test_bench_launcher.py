"""bench.py as the driver starts it: `python bench.py --gpus N` with no launcher around it must start
`python -m torch.distributed.run` as a CHILD (never exec, never touch the GPU itself) and hand its exit code back."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_importing_bench_does_not_import_torch_or_pcr():
    before = set(sys.modules)
    mod = _load_bench()
    new = set(sys.modules) - before
    assert mod.torch is None and mod.pcr is None
    assert not any(m == "pcr" or m.startswith("pcr.") for m in new)


def test_self_launch_starts_torchrun_as_a_child_and_returns_its_code(monkeypatch):
    mod = _load_bench()
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    import subprocess
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert mod.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_takes_the_launcher_branch_only_without_world_size(monkeypatch):
    mod = _load_bench()
    calls = []
    monkeypatch.setattr(mod, "self_launch", lambda n: calls.append(n) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    try:
        mod.main()
    except SystemExit as e:
        assert e.code == 0
    assert calls == [8] and mod.torch is None        # decided before torch / pcr were imported
