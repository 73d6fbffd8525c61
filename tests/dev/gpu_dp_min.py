import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mtr_amd
from tests.oracle_binding import Oracle
rng = np.random.RandomState(1)
unit = np.array([0,1,2], np.uint8)
read = np.concatenate([rng.randint(0,4,30), np.tile(unit, 20), rng.randint(0,4,30)]).astype(np.uint8)
eng = mtr_amd.Engine(); orc = Oracle()
eng.upload([read])
for U, qs, qe in ((3, 20, 100), (3, 0, 119), (1, 5, 60), (17, 0, 119)):
    u = read[40:40+U].copy()
    print("task", U, qs, qe, flush=True)
    t0 = time.time()
    out = eng.test_wrap_dp([(0, qs, qe, u, 1, 1, 3)])
    print("  gpu   ", tuple(int(x) for x in out[0]), f"{time.time()-t0:.3f}s", flush=True)
    print("  oracle", orc.wrap_dp(read, qs, qe, u, 1, 1, 3), flush=True)
