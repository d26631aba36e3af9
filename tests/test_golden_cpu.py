"""CPU: the committed golden vectors (tests/golden/, made by tools/make_golden.py) still match both
oracle twins -- a regression pin on the checker itself."""
import numpy as np
import pytest

from oracle import develop_np as dn
from tests.golden_util import load_golden

CASES = load_golden()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_matches_oracles(refc, case):
    u = refc.make_uniforms(case["params"], case["wb"], case["cm"], case["zoom"], *case["pan"], case["black_level"])
    f32 = refc.render_f32(case["cfa"], u, case["tw"], case["th"])
    assert np.array_equal(f32.view(np.uint32), case["f32"].view(np.uint32))
    assert np.array_equal(refc.pack_u8(f32), case["u8"])
    assert np.array_equal(refc.pack_f16(f32).view(np.uint16), case["f16"])
    assert np.array_equal(refc.histogram(case["u8"]), case["hist"])
    twin = dn.render_f32(case["cfa"], dn.Uniforms(**case["params"], wb=tuple(case["wb"]), cm=tuple(case["cm"]),
                                                  zoom=case["zoom"], pan_x=case["pan"][0], pan_y=case["pan"][1],
                                                  black_level=case["black_level"]), case["tw"], case["th"])
    assert np.array_equal(twin.view(np.uint32), case["f32"].view(np.uint32))
