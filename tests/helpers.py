"""Shared helpers for the parity tests (oracle-side construction of inputs and comparisons)."""
import numpy as np

WB_DAYLIGHT = (2.0, 1.0, 1.5, 1.0)
CM_IDENTITY = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
CM_TEST = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)

PARAM_NAMES = ("exposure", "contrast", "highlights", "shadows", "whites", "blacks",
               "vibrance", "saturation", "temperature", "tint")
UI_RANGES = {"exposure": (-5, 5), "contrast": (-10, 10), "highlights": (-1, 1), "shadows": (-1, 1),
             "whites": (0.8, 1.2), "blacks": (0, 0.2), "vibrance": (-1, 1), "saturation": (-100, 100),
             "temperature": (-1, 1), "tint": (-1, 1)}


def random_params(rng):
    """Uniform draw from the UI ranges in EditParams field order, rounded to f32."""
    return {k: float(np.float32(rng.uniform(*UI_RANGES[k]))) for k in PARAM_NAMES}


MILD_SPAN = {"exposure": 1.0, "contrast": 30.0, "saturation": 40.0, "whites": 0.1, "blacks": 0.03}   # others: 0.3


def mild_params(rng):
    """A mild edit around the defaults: most pixels of a 12-bit frame stay strictly inside (0, 1), where rounding shows (a draw
    over the whole UI ranges saturates most of them)."""
    return {k: float(np.float32((1.0 if k == "whites" else 0.0) + rng.uniform(-1.0, 1.0) * MILD_SPAN.get(k, 0.3)))
            for k in PARAM_NAMES}


def random_cfa(rng, h, w, hi=4096):
    return rng.integers(0, hi, (h, w), dtype=np.uint16)


def ulp_diff(a, b):
    """Max distance in units of float32 representation between two float32 arrays (NaN == NaN)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2**31) - ai, ai)
    bi = np.where(bi < 0, np.int64(-2**31) - bi, bi)
    d = np.abs(ai - bi)
    d = np.where(np.isnan(a) & np.isnan(b), 0, d)
    return int(d.max()) if d.size else 0


def expected_taps(cfa):
    """Demosaic selection written from the table in SURVEY.md section 8 (a2) -- deliberately NOT
    shared with either oracle: for every pixel (py, px) the (r, g, b) raw samples it must pick."""
    h, w = cfa.shape

    def at(y, x):
        return int(cfa[min(max(y, 0), h - 1), min(max(x, 0), w - 1)])

    out = np.zeros((h, w, 3), np.int64)
    for py in range(h):
        for px in range(w):
            odd_row, even_col = (py % 2 == 1), (px % 2 == 0)
            if odd_row and even_col:
                g, b, r = at(py, px), at(py, px + 1), at(py + 1, px)
            elif odd_row:
                b, g, r = at(py, px), at(py, px - 1), at(py + 1, px - 1)
            elif even_col:
                r, g, b = at(py, px), at(py, px + 1), at(py - 1, px)
            else:
                g, r, b = at(py, px), at(py, px - 1), at(py - 1, px)
            out[py, px] = (r, g, b)
    return out
