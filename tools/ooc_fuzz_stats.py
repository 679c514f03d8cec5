"""How often the out-of-core fuzz (tests/test_gpu_out_of_core_fuzz.py) really runs out of core, spills, checkpoints: run on the GPU box."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa: F401  (paths)
import pcr
import test_gpu_out_of_core_fuzz as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
banded = spilled = ck = filt_n = 0
for seed in range(n):
    og, specs, filt, clouds, knobs = F.build(seed)
    with tempfile.TemporaryDirectory() as d:
        p = pcr.Pipeline.create(F.config(og, specs, filt, gpu_memory_budget=max(knobs["gpu"], 1 << 12), host_cache_budget=knobs["host"], state_dir=d))
        b = p.out_of_core()
        files = 0
        for c in clouds:
            p.ingest(F.to_cloud(c))
        if b:
            for _, _, fs in os.walk(p.spill_dir()):
                files += len(fs)
        banded += b
        spilled += files > 0
        ck += knobs["checkpoint_after"] is not None
        filt_n += filt is not None
        del p
print(f"{n} seeds: out of core {banded}, with tile files on disk {spilled}, checkpointed {ck}, filtered {filt_n}")
