import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "develop_golden.npz")


def load_golden():
    """-> list of dicts: name, params, wb, cm, zoom, pan, black_level, tw, th, cfa, f32, u8, f16, hist."""
    z = np.load(GOLDEN, allow_pickle=False)
    cases = json.loads(bytes(z["cases_json"]).decode())
    for c in cases:
        for k in ("cfa", "f32", "u8", "f16", "hist"):
            c[k] = z[f"{c['name']}/{k}"]
    return cases
