"""Parity bookkeeping: every GPU parity comparison leaves its achieved error next to the tolerance it was held to.

The `-m gpu` tests call `close()` / `record()`; the numbers are merged into one JSON file (default
`gpurun_out/r6_parity.json` under the repository root, or `$R2S_PARITY_LOG`), so that the margin to the 1e-4 rel / 1e-5 abs
gates of BASELINE.json is a number and not just a passed assertion.  `profiles/r6_parity.json` is a committed copy of the
file a full `pytest -m gpu` run on the MI355X wrote."""
import inspect
import json
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _path():
    return os.environ.get("R2S_PARITY_LOG", os.path.join(_ROOT, "gpurun_out", "r6_parity.json"))


def _test_id():
    return os.environ.get("PYTEST_CURRENT_TEST", "interactive").split(" ")[0]


def record(label, **metrics):
    """Merge `metrics` under [current test][label]; numeric fields keep their MAXIMUM over repeated calls."""
    try:
        p = _path()
        os.makedirs(os.path.dirname(p), exist_ok=True)
        try:
            db = json.load(open(p))
        except Exception:
            db = {}
        ent = db.setdefault(_test_id(), {}).setdefault(label, {})
        for k, v in metrics.items():
            if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool):
                v = float(v)
                ent[k] = max(ent.get(k, v), v) if k != "tol" else v
            else:
                ent[k] = v
        ent["calls"] = ent.get("calls", 0) + 1
        with open(p + ".tmp", "w") as f:
            json.dump(db, f, indent=1, sort_keys=True)
        os.replace(p + ".tmp", p)
    except OSError:
        pass  # a read-only tree must not fail a parity test


def close(a, b, tol, what=None):
    """max |a - b| < tol, recorded with the tolerance (label = the calling line unless `what` is given)."""
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    b = b.detach().cpu().numpy() if hasattr(b, "detach") else np.asarray(b)
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0
    if what is None:
        fr = inspect.stack()[1]
        what = f"{os.path.basename(fr.filename)}:{fr.lineno}"
    record(what, max_abs_err=err, tol=float(tol))
    return err < tol
