#!/usr/bin/env python3
"""Random shapes for the conditioned iteration in the cooperative launch (mode 4 and its three-launch form) against the oracle's
joint loss and gradients: calls the parity test of tests/test_hip_coop_step.py with random (M, N, d), small M included.
usage: python tools/fuzz_coop_conditioned.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_hip_coop_step import test_coop_conditioned_iteration_matches_oracle as check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
for case in range(n):
    M = int(rng.choice([2, 3, 4, 5, 6, 7, 9, 16, 17, 31, 33, 40, 63, 64, 65, 80, 100, 128]))
    d = int(rng.integers(2, 7))
    N = M + int(rng.integers(1, 40))      # (N > M: the oracle's equal-inputs shortcut must not trigger)
    check(M, N, d)
    print("case %d: M = %d, N = %d, d = %d  ok" % (case, M, N, d), flush=True)
print("campaign passed: %d cases" % n)
