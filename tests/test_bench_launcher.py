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

    class Child:
        pid = 4242

        def wait(self, timeout=None):
            seen["timeout"] = timeout
            return 7

    def fake_popen(cmd, env=None, **kw):
        seen["cmd"], seen["env"], seen["kw"] = cmd, env, kw
        return Child()

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert mod.self_launch(4, limit_s=99.0) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    # torchrun's own rendezvous on a port it binds itself (no pick-then-use race), loopback only
    assert "--nproc-per-node=4" in cmd and "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert "--master-port" not in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["kw"].get("start_new_session") is True and seen["timeout"] == 99.0


def test_self_launch_kills_the_child_group_at_the_wall_clock_limit_and_exits_non_zero(monkeypatch, tmp_path):
    """A real child that never finishes (a stand-in for ranks stuck in a collective): the launcher ends exactly the
    process group it started and returns 124."""
    import subprocess
    import time
    mod = _load_bench()
    sleeper = tmp_path / "sleeper.py"
    pidfile = tmp_path / "pid"
    sleeper.write_text("import os, sys, time\nopen(sys.argv[1], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    real_popen = subprocess.Popen

    def popen_sleeper(cmd, env=None, **kw):
        return real_popen([sys.executable, str(sleeper), str(pidfile)], env=env, **kw)

    monkeypatch.setattr(subprocess, "Popen", popen_sleeper)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    t0 = time.time()
    assert mod.self_launch(2, limit_s=2.0) == 124
    assert time.time() - t0 < 30
    pid = int(pidfile.read_text())
    time.sleep(0.2)
    try:
        os.kill(pid, 0)
        alive = True
    except ProcessLookupError:
        alive = False
    assert not alive, "the launcher left its child running"


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
