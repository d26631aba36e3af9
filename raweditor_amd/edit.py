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

    def to_json(self) -> str:                 # edit.rs:105-107 (serde field order, shortest f32 repr)
        def num(v: float) -> str:
            if not np.isfinite(v):
                return "null"                  # serde_json writes non-finite floats as null
            return np.format_float_positional(np.float32(v), unique=True, trim="0")
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
