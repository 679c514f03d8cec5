"""Soak run of the mixed-pipeline differential test (tests/test_gpu_pipeline_fuzz.py) over seeds the suite does not contain
(run by hand on the GPU box: python tests/soak_pipeline_fuzz.py FIRST LAST).  Pipelines are fresh per seed, so every Point
group goes through the scatter that also stores its bands.  Uses the oracle: test infrastructure."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "pointcloud-raster_amd", "python"))
import test_gpu_pipeline_fuzz as P               # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    try:
        P.test_mixed_pipeline_matches_oracle(seed)
    except BaseException as e:                    # noqa: BLE001
        if type(e).__name__ != "Skipped":
            bad.append((seed, repr(e)[:300]))
            traceback.print_exc()
    if seed % 100 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done", first, last, "failures:", len(bad))
for b in bad:
    print(b)
sys.exit(1 if bad else 0)
