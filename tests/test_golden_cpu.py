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


def test_reference_kit_is_the_golden_vectors_as_plain_files(refc):
    """tests/golden/reference_kit/ (tools/export_reference_kit.py) -- the inputs and expected RGBA8
    bytes a Rust #[test] inside the reference reads -- holds exactly the full-resolution cases of the .npz, and the
    EditParams JSON in it round-trips through the host mirror with the reference's own field names."""
    import json
    import os
    import raweditor_amd as ra
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kit = os.path.join(root, "tests", "golden", "reference_kit")
    meta = json.load(open(os.path.join(kit, "cases.json")))
    by_name = {c["name"]: c for c in CASES}
    assert len(meta) >= 5
    for m in meta:
        c = by_name[m["name"]]
        h, w = c["cfa"].shape
        assert (m["width"], m["height"]) == (w, h) and c["zoom"] == 1.0 and c["black_level"] == 0 and (c["tw"], c["th"]) == (w, h)
        cfa = np.fromfile(os.path.join(kit, m["name"] + ".cfa.u16le"), dtype="<u2").reshape(h, w)
        assert np.array_equal(cfa, c["cfa"])
        rgba = np.fromfile(os.path.join(kit, m["name"] + ".rgba8"), dtype=np.uint8).reshape(h, w, 4)
        assert np.array_equal(rgba, c["u8"])
        p = ra.EditParams.from_json(json.dumps(m["params"]))           # every field present: serde would accept it
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, m["wb"], m["cm"])
        assert np.array_equal(refc.pack_u8(refc.render_f32(cfa, u)), rgba)
