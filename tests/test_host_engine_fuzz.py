"""The mixed-pipeline differential test of tests/test_gpu_pipeline_fuzz.py on the HOST engine (ExecutionMode.CPU): the same
random pipelines -- several ReductionSpecs of mixed glyphs and value channels, an optional FilterSpec, several ingests, grids with
odd cell sizes and tiles -- against the same oracle, to the same tolerances.  Runs in the CPU suite."""
import pytest

import pcr
from test_gpu_pipeline_fuzz import check_mixed_pipeline


@pytest.mark.parametrize("seed", range(40))
def test_mixed_pipeline_on_the_host_engine_matches_oracle(seed):
    check_mixed_pipeline(seed, pcr.ExecutionMode.CPU, "host")
