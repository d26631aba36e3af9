"""EditParams -- host mirror of `state::edit::EditParams` (reference src/state/edit.rs:15-122).

Same ten f32 fields in the same order (it is the serde JSON order and the uniform-block order),
same defaults (all 0 except whites = 1, edit.rs:81-95), same helpers (`new`, `to_json`,
`from_json`, `is_unedited`, `reset`).  Values are kept as IEEE binary32, like the Rust struct.
"""
from __future__ import annotations

import dataclasses
import json

import numpy as np

from ._lib import RdEditParams

FIELDS = ("exposure", "contrast", "highlights", "shadows", "whites", "blacks",
          "vibrance", "saturation", "temperature", "tint")

# slider ranges of the Develop sidebar (reference src/main.rs:1624-1660)
UI_RANGES = {
    "exposure": (-5.0, 5.0), "contrast": (-10.0, 10.0), "highlights": (-1.0, 1.0),
    "shadows": (-1.0, 1.0), "whites": (0.8, 1.2), "blacks": (0.0, 0.2), "vibrance": (-1.0, 1.0),
    "saturation": (-100.0, 100.0), "temperature": (-1.0, 1.0), "tint": (-1.0, 1.0),
}


def _f32(x) -> float:
    return float(np.float32(x))


def ryu_f32(v) -> str:
    """The text serde_json writes for a finite f32 (EditParams::to_json, edit.rs:105-107): the `ryu` crate's format32 -- the
    shortest decimal digits that round-trip, laid out by the decimal exponent: with `digits` x 10^k and kk = len(digits) + k,
      0 <= k and kk <= 13   ->  digits, k zeros, ".0"         (1.0, 16777216.0, 9999999827968.0)
      0 <  kk <= 13         ->  point inside the digits       (12.34)
      -6 < kk <= 0          ->  "0.", -kk zeros, digits       (0.001234, 0.000001)
      otherwise             ->  d[.ddd]e<exp>, no '+', no padding   (1e-7, 1.5e-7, 1e13, 3.4028235e38)
    and a sign for negative values including -0.0.  (For f64 the two 13s are 16s; EditParams' fields are f32.)"""
    v = np.float32(v)
    if v == 0:
        return "-0.0" if np.signbit(v) else "0.0"
    s = np.format_float_scientific(v, unique=True, trim="-", exp_digits=1)        # '1.5e-07' style pieces: digits + exponent
    sign = "-" if s.startswith("-") else ""
    mant, exp = s.lstrip("-").split("e")
    digits = mant.replace(".", "")
    e10 = int(exp)
    n = len(digits)
    k, kk = e10 - (n - 1), e10 + 1
    if 0 <= k and kk <= 13:
        body = digits + "0" * k + ".0"
    elif 0 < kk <= 13:
        body = digits[:kk] + "." + digits[kk:]
    elif -6 < kk <= 0:
        body = "0." + "0" * (-kk) + digits
    elif n == 1:
        body = f"{digits}e{kk - 1}"
    else:
        body = f"{digits[0]}.{digits[1:]}e{kk - 1}"
    return sign + body


@dataclasses.dataclass
class EditParams:
    exposure: float = 0.0
    contrast: float = 0.0
    highlights: float = 0.0
    shadows: float = 0.0
    whites: float = 1.0
    blacks: float = 0.0
    vibrance: float = 0.0
    saturation: float = 0.0
    temperature: float = 0.0
    tint: float = 0.0

    def __post_init__(self):
        for f in FIELDS:
            setattr(self, f, _f32(getattr(self, f)))

    @classmethod
    def default(cls) -> "EditParams":        # edit.rs:79-96 (impl Default)
        return cls()

    @classmethod
    def new(cls) -> "EditParams":            # edit.rs:100-102
        return cls.default()

    def to_json(self) -> str:                 # edit.rs:105-107: serde's field order, ryu's text for every f32 (ryu_f32 above)
        def num(v: float) -> str:
            if not np.isfinite(v):
                return "null"                  # serde_json writes non-finite floats as null
            return ryu_f32(v)
        return "{" + ",".join(f'"{f}":{num(getattr(self, f))}' for f in FIELDS) + "}"

    @classmethod
    def from_json(cls, text: str) -> "EditParams":   # edit.rs:110-112: every field is required
        obj = json.loads(text)
        if not isinstance(obj, dict):
            raise ValueError("EditParams JSON must be an object")
        missing = [f for f in FIELDS if f not in obj]
        if missing:
            raise ValueError(f"missing field `{missing[0]}`")
        for f in FIELDS:                          # serde: an f32 field takes a JSON number and nothing else (null, which
            v = obj[f]                            # to_json writes for a non-finite value, is rejected on the way back in)
            if isinstance(v, bool) or not isinstance(v, (int, float)):
                raise ValueError(f"invalid type for `{f}`: expected f32, found {type(v).__name__}")
        return cls(**{f: obj[f] for f in FIELDS})

    def is_unedited(self) -> bool:            # edit.rs:115-117
        return self == EditParams()

    def reset(self) -> None:                  # edit.rs:120-122
        for f in FIELDS:
            setattr(self, f, getattr(EditParams(), f))

    def to_c(self) -> RdEditParams:
        return RdEditParams(*[getattr(self, f) for f in FIELDS])

    @classmethod
    def random(cls, rng: np.random.Generator) -> "EditParams":
        """Uniform draw from the UI ranges, in field order (SURVEY.md section 8d 'randomised')."""
        return cls(**{f: rng.uniform(*UI_RANGES[f]) for f in FIELDS})
