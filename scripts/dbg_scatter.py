import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from raytracer_amd import abi
from raytracer_amd.tracer import scatter_rays
rng = np.random.default_rng(1)
n = 200000
normal = rng.normal(size=(n,3)).astype(np.float32)
in_dir = rng.normal(size=(n,3)).astype(np.float32)
pcgs = np.zeros((n,2), dtype=np.uint64)
for i in range(n):
    p = O.PCG(int(rng.integers(0, 2**63)), int(rng.integers(0, 2**63)))
    pcgs[i] = (p.state, p.inc)
for kind in (0, 1):
    d, adv = scatter_rays(kind, pcgs, in_dir, normal)
    want = np.zeros_like(d)
    for i in range(n):
        p = O.PCG(); p.st[0] = int(pcgs[i,0]); p.st[1] = int(pcgs[i,1])
        r = O.scatter_ray(kind, p, in_dir[i], (0,0,0), normal[i], 1)
        want[i] = r.direction[:]
        assert p.st[0] == adv[i,0]
    err = np.abs(d - want).max(axis=1)
    print("kind", kind, "exact", (err==0).mean(), "max", err.max(), "quantiles", np.quantile(err, [0.5,0.9,0.99,0.999,0.9999]))
