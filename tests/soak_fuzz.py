"""Soak run of the randomised differential test over seeds the suite does not contain (run by hand on the GPU box:
python tests/soak_fuzz.py FIRST LAST).  Uses the oracle: test infrastructure."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_cabi                    # noqa: E402
import test_gpu_fuzz as F                         # noqa: E402
import test_gpu_variants as V                     # noqa: E402

A = load_cabi()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    try:
        F.test_random_case_matches_oracle.__wrapped__(A, seed) if hasattr(F.test_random_case_matches_oracle, "__wrapped__") \
            else F.test_random_case_matches_oracle(A, seed)
    except BaseException as e:                    # noqa: BLE001
        bad.append(("fuzz", seed, repr(e)[:300]))
        traceback.print_exc()
    if seed % 4 == 0:
        try:
            V.test_random_large_gaussians_on_the_matrix_core_pass(A, seed)
        except BaseException as e:                # noqa: BLE001
            if type(e).__name__ != "Skipped":
                bad.append(("large_gauss", seed, repr(e)[:300]))
                traceback.print_exc()
    if seed % 100 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("done", first, last, "failures:", len(bad))
for b in bad:
    print(b)
sys.exit(1 if bad else 0)
