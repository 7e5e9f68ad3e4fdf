"""One-off: MITC3+ stiffness of many batch sizes (ragged last wave) -> .npz; run once with FEMSHELL_M3_SIMPLE=1 and once
without, then compare (tools/exp/m3_fuzz.py out.npz | tools/exp/m3_fuzz.py --cmp a.npz b.npz)."""
import sys
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    worst = 0.0
    for k in a.files:
        worst = max(worst, float(np.abs(a[k] - b[k]).max() / max(1e-300, np.abs(a[k]).max())))
        assert a[k].shape == b[k].shape
    print("sizes", len(a.files), "worst relative difference", worst)
    sys.exit(0 if worst < 1e-13 else 1)
import torch
sys.path.insert(0, '.')
import fem_shell_amd as fsa
from fem_shell_amd import batch
from tests import util
props = fsa.build_prop_table(util.standard_props())
pd = batch.to_device_props(props)
out = {}
rng = np.random.default_rng(3)
for E in list(range(1, 26)) + [59, 60, 61, 639, 640, 641, 10007]:
    xyz = rng.standard_normal((E, 3, 3)) + rng.uniform(-50, 50, (E, 1, 3))
    pid = rng.integers(0, len(props), E).astype(np.int32)
    o = batch.shell_batch(3, torch.from_numpy(xyz).cuda(), torch.from_numpy(pid).cuda(), pd, len(props), want_K=True)
    out[f"K{E}"] = o["K"].cpu().numpy()
    out[f"st{E}"] = o["status"].cpu().numpy().astype(np.float64)
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out))
