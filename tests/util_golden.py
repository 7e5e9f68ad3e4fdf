import os

import numpy as np

from fem_shell_amd.properties import PROP_DTYPE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def golden_props(g, prefix="prop_"):
    """fs_shell_prop rows as the REFERENCE evaluated them (prop_* arrays of a fixture)."""
    n = g[prefix + "h"].shape[0]
    t = np.zeros(n, dtype=PROP_DTYPE)
    for k in ("A", "B", "D", "Cs", "h", "m_t", "m_r", "k_drill", "C_raw", "flags", "nonlinear"):
        t[k] = g[prefix + k]
    if prefix + "rho" in g.files:        # fixtures written before the centrifugal prestress rows
        t["rho"] = g[prefix + "rho"]
    return t
