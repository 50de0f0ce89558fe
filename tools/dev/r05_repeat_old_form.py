"""dev: the 1500-launch bitwise repeat test (tests/test_gpu_fresh_process.py) on the score kernel form of rounds 4 - 5a (k_from_s = 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fresh_process as T
from pepflowww_amd import synth
try:
    T.test_fused_score_kernel_is_bitwise_stable_over_many_launches.__wrapped__ if False else None
    T.test_fused_score_kernel_is_bitwise_stable_over_many_launches(synth.seeded_state_dict(), False)
    print("0 launches differ")
except AssertionError as e:
    print(str(e).splitlines()[0])
