"""`import pcr` must not depend on whether torch was imported first: libpcr_hip.so and PyTorch-ROCm both need
`libamdhip64.so.7` and the loader keeps the first copy it sees -- pcr preloads torch's copy when torch is installed,
so both orders end with ONE HIP runtime in the process (pcr/__init__.py, _share_hip_runtime_with_torch)."""
import os
import subprocess
import sys

import pytest

from conftest import PKG_PY

CODE = """
import sys
sys.path.insert(0, {pkg!r})
{first}
{second}
import pcr
paths = pcr.hip_runtime_paths()
print("RUNTIMES", len(paths), paths)
"""


@pytest.mark.parametrize("first,second", [("import pcr", "import torch"), ("import torch", "import pcr")],
                         ids=["pcr-then-torch", "torch-then-pcr"])
def test_one_hip_runtime_whatever_the_import_order(first, second):
    out = subprocess.run([sys.executable, "-c", CODE.format(pkg=PKG_PY, first=first, second=second)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RUNTIMES")][0]
    assert line.split()[1] == "1", line
    import importlib.util
    torch_dir = os.path.dirname(importlib.util.find_spec("torch").origin)
    assert torch_dir in line, f"the shared runtime should be torch's bundled copy: {line}"
