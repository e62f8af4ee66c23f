"""pytest config: `gpu` marker + repo root on sys.path.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI
symbol checks, gloo world_size-2 tests.  `-m gpu` runs on an MI355X: the HIP
kernels, through the C-ABI, against the oracle and the golden fixtures.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def rel_close(a, b, tol=1e-5):
    """|a-b| <= tol*max(1,|b|) — the tolerance form SURVEY.md section 8d prescribes."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    return float(err.max()) if err.size else 0.0


def bounded(name, err, tol):
    """assert err <= tol for one of the multi-step trace bounds that are looser than the contract's 1e-5 (each call site
    carries the derivation of its number).  With GYMRL_TOL_LEDGER=<file> every (name, observed, bound) is appended to that
    file, so the observed drift of a GPU run can be committed next to the bound (profiles/r04_trace_tolerances.json)."""
    err = float(err)
    path = os.environ.get("GYMRL_TOL_LEDGER")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps({"name": name, "observed": err, "bound": tol}) + "\n")
    assert err <= tol, (name, err, tol)


def cov_clip_mul(g, k):
    """corr_mul of golden case k of ppo_full_loss.npz: None unless the case ran the covariance clip, else the rows
    gymrl_amd's cov_clip_mask picks when fed the reference's own randperm."""
    if g[f"c{k}_cov_perm"].size == 0:
        return None
    import types
    import torch
    from gymrl_amd.ppo_full_lunarlander import cov_clip_mask
    ratio, cmin, cmax = (float(x) for x in g[f"c{k}_cov_cfg"])
    cfg = types.SimpleNamespace(clip_cov_ratio=ratio, clip_cov_min=cmin, clip_cov_max=cmax)
    return cov_clip_mask(cfg, torch.from_numpy(g[f"c{k}_logits"]), torch.from_numpy(g[f"c{k}_actions"]),
                         torch.from_numpy(g[f"c{k}_adv"]), perm=g[f"c{k}_cov_perm"]).numpy()
