import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.tools import fuzz_parity as F
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
rng = np.random.default_rng(2026)
c = F.make_case(rng, False, batches=[5, 8, 9, 16, 17, 24, 32, 33, 40, 56, 57, 64, 65, 80, 100, 112, 120], n_factor=25)
print(F.describe(0, c))
for env in ({}, {"SOBER_NYSTROM_SKIP": "1"}):
    os.environ.pop("SOBER_NYSTROM_SKIP", None)
    os.environ.update(env)
    os.environ["SOBER_NYSTROM_DEBUG"] = "1"
    print(env, F.check_case(c, dev, verbose=True, always_diagnose=True))
