"""CPU: the data formats either side of the path: the app's SQLite edits table /
EditParams JSON, and CFA ingest (u16 plane, uncompressed DNG) with loader.rs' post-decode normalisation."""
import sqlite3
import struct

import numpy as np
import pytest

import raweditor_amd as ra
from raweditor_amd import catalog, ingest


def test_edits_table_round_trip(tmp_path):
    conn = sqlite3.connect(tmp_path / "library.db")
    catalog.init_schema(conn)
    for k in range(3):
        conn.execute("INSERT INTO images (path, filename, width, height, imported_at) VALUES (?,?,?,?,?)",
                     (f"/photos/{k}.nef", f"{k}.nef", 6016, 4016, 0))
    conn.commit()
    assert not catalog.has_edits(conn, 1)
    with pytest.raises(KeyError):
        catalog.load_edit_params(conn, 1)                        # QueryReturnedNoRows
    assert catalog.load_edit_params_or_default(conn, 1).is_unedited()
    p = ra.EditParams(exposure=1.5, contrast=2.5, blacks=0.005, saturation=-10.0)
    catalog.save_edit_params(conn, 1, p)
    assert catalog.has_edits(conn, 1) and catalog.load_edit_params(conn, 1) == p
    p.exposure = -0.3
    catalog.save_edit_params(conn, 1, p)                         # library.rs:322-328: UPDATE the newest row
    assert conn.execute("SELECT COUNT(*) FROM edits WHERE image_id = 1").fetchone()[0] == 1
    assert catalog.load_edit_params(conn, 1).exposure == float(np.float32(-0.3))
    # a row written by the Rust app (serde_json: field order, 0.0-style floats) parses
    conn.execute("INSERT INTO edits (image_id, settings_json) VALUES (2, ?)",
                 ('{"exposure":0.7,"contrast":0.0,"highlights":-0.25,"shadows":0.0,"whites":1.0,"blacks":0.0,'
                  '"vibrance":0.0,"saturation":12.0,"temperature":0.1,"tint":0.0}',))
    conn.commit()
    q = catalog.load_edit_params(conn, 2)
    assert (q.exposure, q.highlights, q.saturation) == (float(np.float32(0.7)), -0.25, 12.0)
    man = catalog.export_manifest(conn)
    assert [e.image_id for e in man] == [1, 2, 3] and man[2].params.is_unedited() and man[0].width == 6016
    assert [e.image_id for e in catalog.export_manifest(conn, 1, 2)] == [2]
    back = catalog.manifest_from_json(catalog.manifest_to_json(man))
    assert [(e.image_id, e.path, e.params) for e in back] == [(e.image_id, e.path, e.params) for e in man]


def test_wb_and_matrix_normalisation_follow_loader_rs():
    f = np.float32
    assert ingest.normalise_wb([2.0, 1.0, 1.5, 1.0]) == [2.0, 1.0, 1.5, 1.0]
    wb = ingest.normalise_wb([512.0, 256.0, 384.0])                       # 3 coefficients: G2 := G
    assert wb == [2.0, 1.0, 1.5, 1.0]
    assert ingest.normalise_wb([]) == [1.0, 1.0, 1.0, 1.0]                # neutral fallback
    assert ingest.normalise_wb([2.0, 1.0, 1.5, float("nan")])[3] == 1.0   # invalid G2 -> G
    assert ingest.normalise_wb([2.0, 1.0, 1.5, 0.0])[3] == 1.0
    assert ingest.normalise_wb([1.0, 0.0, 1.0, 1.0])[0] == float(f(1.0) / f(0.001))   # G clamped to 0.001
    m = [[0.9, -0.2, -0.1, 0.0], [-0.4, 1.2, 0.2, 0.0], [-0.1, 0.2, 0.7, 0.0], [0.0, 0.0, 0.0, 0.0]]
    assert ingest.extract_matrix(m) == [float(f(x)) for r in m[:3] for x in r[:3]]
    assert ingest.extract_matrix([[0.0] * 4] * 3) == list(ra.IDENTITY_MATRIX)
    assert ingest.extract_matrix(None) == list(ra.IDENTITY_MATRIX)
    v = ingest.samples_to_u16(np.array([0.0, 0.5, 1.0, 1.5, -0.2], np.float32))      # float sensors
    assert v.tolist() == [0, 32767, 65535, 65535, 0]
    assert ingest.samples_to_u16(np.array([0, 4095, 65535], np.uint16)).tolist() == [0, 4095, 65535]


def test_raw_u16_plane(tmp_path, rng):
    cfa = rng.integers(0, 4096, (10, 14), dtype=np.uint16)
    path = tmp_path / "frame.u16"
    cfa.astype("<u2").tofile(path)
    r = ingest.load_raw_u16(str(path), 14, 10, wb_coeffs=[2.0, 1.0, 1.5])
    assert (r.width, r.height) == (14, 10) and np.array_equal(np.asarray(r.data).reshape(10, 14), cfa)
    assert r.wb_multipliers == [2.0, 1.0, 1.5, 1.0] and r.color_matrix == list(ra.IDENTITY_MATRIX)
    with pytest.raises(FileNotFoundError, match="File not found"):      # loader.rs:46-48, :158-165
        ingest.load_raw_u16(str(tmp_path / "nonexistent.nef"), 14, 10)
    with pytest.raises(ValueError):
        ingest.load_raw_u16(str(path), 15, 10)


def _pack_rows(cfa, bits):
    """samples of `bits` bits, MSB first, every row padded to a whole byte (TIFF FillOrder 1)"""
    h, w = cfa.shape
    b = ((cfa[:, :, None].astype(np.uint32) >> np.arange(bits - 1, -1, -1)) & 1).astype(np.uint8).reshape(h, w * bits)
    return np.packbits(b, axis=1)                        # pads each row with zero bits


def _write_dng(path, cfa, endian="<", neutral=(0.5, 1.0, 2.0 / 3.0), strips=2, bits=16, extra_raw_tags=()):
    """A minimal uncompressed CFA DNG: IFD0 (thumbnail-less metadata) -> SubIFD with the raw strips."""
    h, w = cfa.shape
    e = endian
    rows = (h + strips - 1) // strips
    if bits == 16:
        data, row_bytes = cfa.astype(e + "u2").tobytes(), w * 2
    else:
        packed = _pack_rows(cfa, bits)
        data, row_bytes = packed.tobytes(), packed.shape[1]
    chunks = [data[i * rows * row_bytes:(i + 1) * rows * row_bytes] for i in range(strips)]
    chunks = [c for c in chunks if c]

    def entry(tag, typ, vals, blob_off):
        fmt = {1: "B", 3: "H", 4: "I", 5: "I", 10: "i"}[typ]
        n = len(vals) // (2 if typ in (5, 10) else 1)
        raw = struct.pack(e + fmt * len(vals), *vals)
        if len(raw) <= 4:
            return struct.pack(e + "HHI", tag, typ, n) + raw.ljust(4, b"\0"), b""
        return struct.pack(e + "HHII", tag, typ, n, blob_off), raw

    def build_ifd(entries, base):
        body, blobs, off = b"", b"", base + 2 + 12 * len(entries) + 4
        for tag, typ, vals in sorted(entries):
            ent, blob = entry(tag, typ, vals, off + len(blobs))
            body += ent
            blobs += blob
        return struct.pack(e + "H", len(entries)) + body + struct.pack(e + "I", 0) + blobs

    header = (b"II" if e == "<" else b"MM") + struct.pack(e + "HI", 42, 8)
    neutral_r = [v for x in neutral for v in (int(round(x * 1000000)), 1000000)]
    cm = [v for x in (0.9, -0.2, -0.1, -0.4, 1.2, 0.2, -0.1, 0.2, 0.7) for v in (int(round(x * 10000)), 10000)]
    ifd0_entries = [(330, 4, [0]), (50728, 10, neutral_r), (50721, 10, cm)]
    ifd0_len = len(build_ifd(ifd0_entries, 8))
    sub_off = 8 + ifd0_len
    sub_entries = [(256, 4, [w]), (257, 4, [h]), (258, 3, [bits]), (259, 3, [1]), (262, 3, [32803]), (277, 3, [1]),
                   (278, 4, [rows]), (273, 4, [0] * len(chunks)), (279, 4, [len(c) for c in chunks])] + list(extra_raw_tags)
    sub_len = len(build_ifd(sub_entries, sub_off))
    data_off = sub_off + sub_len
    offs, o = [], data_off
    for c in chunks:
        offs.append(o)
        o += len(c)
    sub_entries[7] = (273, 4, offs)
    ifd0_entries[0] = (330, 4, [sub_off])
    with open(path, "wb") as fh:
        fh.write(header + build_ifd(ifd0_entries, 8) + build_ifd(sub_entries, sub_off) + b"".join(chunks))


@pytest.mark.parametrize("endian", ["<", ">"])
def test_uncompressed_dng(tmp_path, rng, endian):
    cfa = rng.integers(0, 16384, (9, 12), dtype=np.uint16)
    path = tmp_path / "frame.dng"
    _write_dng(path, cfa, endian, strips=3)
    r = ingest.load_dng_uncompressed(str(path))
    assert (r.width, r.height) == (12, 9) and np.array_equal(r.data.reshape(9, 12), cfa)
    assert r.wb_multipliers == pytest.approx([2.0, 1.0, 1.5, 1.0], rel=1e-5)       # 1/neutral, / G
    assert r.color_matrix == pytest.approx([0.9, -0.2, -0.1, -0.4, 1.2, 0.2, -0.1, 0.2, 0.7], rel=1e-6)
    (tmp_path / "junk.dng").write_bytes(b"not a tiff at all")
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.load_dng_uncompressed(str(tmp_path / "junk.dng"))


def test_dng_levels_and_cfa_pattern(tmp_path, rng):
    """BlackLevel / WhiteLevel / CFAPattern travel with the result as extras (the reference's struct drops them); a file
    without them keeps the reference's assumptions."""
    cfa = rng.integers(0, 4096, (6, 8), dtype=np.uint16)
    p = tmp_path / "levels.dng"
    _write_dng(p, cfa, extra_raw_tags=[(50713, 3, [2, 2]), (50714, 5, [2560, 10, 2570, 10, 2570, 10, 2580, 10]), (50717, 3, [4095]),
                                       (33421, 3, [2, 2]), (33422, 1, [1, 0, 2, 1])])
    r = ingest.load_dng(str(p))
    assert np.array_equal(r.data.reshape(6, 8), cfa)
    assert r.black_levels == pytest.approx([256.0, 257.0, 257.0, 258.0]) and r.black_level == 257
    assert r.white_level == 4095 and r.cfa_pattern == "GRBG"
    _write_dng(p, cfa, extra_raw_tags=[(50714, 4, [64])])
    r = ingest.load_dng(str(p))
    assert (r.black_level, r.black_levels, r.white_level, r.cfa_pattern) == (64, [64.0], 65535, "RGGB")
    _write_dng(p, cfa)
    r = ingest.load_dng(str(p))
    assert (r.black_level, r.black_levels, r.white_level, r.cfa_pattern) == (0, [], 65535, "RGGB")


def test_damaged_tiff_containers_fail_cleanly(tmp_path, rng):
    """The file is not trusted: an IFD chain that loops, offsets and sizes that point outside the
    file, missing strip tags and short files all end in the reference's "Failed to decode RAW" error
    (loader.rs:50-54) -- never in a hang, a struct.error or silently truncated data."""
    cfa = rng.integers(0, 16384, (6, 8), dtype=np.uint16)
    good = tmp_path / "good.dng"
    _write_dng(good, cfa, "<", strips=2)
    buf = bytearray(good.read_bytes())
    (n0,) = struct.unpack_from("<H", buf, 8)
    next_off_pos = 8 + 2 + 12 * n0

    def variant(name, edit):
        b = bytearray(buf)
        edit(b)
        p = tmp_path / name
        p.write_bytes(bytes(b))
        return str(p)

    # next-IFD pointer of IFD0 points back at IFD0: terminates, and still finds the CFA image
    loop = variant("loop.dng", lambda b: struct.pack_into("<I", b, next_off_pos, 8))
    r = ingest.load_dng_uncompressed(loop)
    assert np.array_equal(r.data.reshape(6, 8), cfa)
    # next-IFD pointer outside the file
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.load_dng_uncompressed(variant("far.dng", lambda b: struct.pack_into("<I", b, next_off_pos, len(b) + 100)))
    # truncated file: strips run past the end
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        p = tmp_path / "short.dng"
        p.write_bytes(bytes(buf[:-20]))
        ingest.load_dng_uncompressed(str(p))
    # a value offset outside the file (the ColorMatrix1 blob of IFD0)
    def break_value_offset(b):
        for i in range(n0):
            tag, typ, cnt, val = struct.unpack_from("<HHII", b, 10 + 12 * i)
            if tag == 50721:
                struct.pack_into("<I", b, 10 + 12 * i + 8, len(b) - 4)
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.load_dng_uncompressed(variant("val.dng", break_value_offset))
    for tiny in (b"", b"II*\x00", b"II*\x00\x08\x00\x00\x00"):
        p = tmp_path / "tiny.dng"
        p.write_bytes(tiny)
        with pytest.raises(ValueError, match="Failed to decode RAW"):
            ingest.load_dng_uncompressed(str(p))


def test_float_samples_and_json_edges():
    """loader.rs:62-73 casts with Rust's saturating `as u16` (NaN -> 0); serde rejects null for an f32 field."""
    v = np.array([np.nan, -1.0, 0.5, 2.0, np.inf, -np.inf], np.float32)
    assert ingest.samples_to_u16(v).tolist() == [0, 0, 32767, 65535, 65535, 0]
    from raweditor_amd import EditParams
    p = EditParams(exposure=float("nan"))
    text = p.to_json()
    assert '"exposure":null' in text
    with pytest.raises(ValueError, match="exposure"):
        EditParams.from_json(text)
    with pytest.raises(ValueError, match="contrast"):
        EditParams.from_json(EditParams().to_json().replace('"contrast":0.0', '"contrast":"1"'))
    with pytest.raises(ValueError):
        EditParams.from_json(EditParams().to_json().replace('"contrast":0.0', '"contrast":true'))
    assert EditParams.from_json(EditParams().to_json().replace('"contrast":0.0', '"contrast":3')).contrast == 3.0


@pytest.mark.parametrize("bits", [8, 10, 12, 14])
@pytest.mark.parametrize("endian", ["<", ">"])
def test_uncompressed_dng_with_packed_samples(tmp_path, rng, bits, endian):
    """BitsPerSample 8 / 10 / 12 / 14: samples packed MSB first whatever the file's byte order, rows byte-aligned (an odd
    width makes the padding real)."""
    cfa = rng.integers(0, 1 << bits, (7, 13), dtype=np.uint16)
    path = tmp_path / f"packed{bits}.dng"
    _write_dng(path, cfa, endian, strips=3, bits=bits)
    r = ingest.load_dng_uncompressed(str(path))
    assert (r.width, r.height) == (13, 7) and r.data.dtype == np.uint16
    assert np.array_equal(r.data.reshape(7, 13), cfa)
    assert np.array_equal(ingest.unpack_bits(_pack_rows(cfa, bits), 13, bits), cfa)


# ---- lossless-JPEG compressed DNG (Compression = 7): librawdev's decoder against an independent encoder ------------------
@pytest.mark.parametrize("predictor", [1, 2, 3, 4, 5, 6, 7])
def test_ljpeg_round_trip_every_predictor(rng, predictor):
    from tests.ljpeg_encoder import encode
    a = rng.integers(0, 1 << 14, (9, 22), dtype=np.uint16)
    a[3, 5], a[3, 6], a[4, 0] = 0, 16383, 16383                   # large differences of both signs
    for comps in (1, 2):
        got = ingest.ljpeg_decode(encode(a, components=comps, precision=14, predictor=predictor))
        assert got.shape == (9, 22) and np.array_equal(got, a), (predictor, comps)


def test_ljpeg_precision_restart_point_transform_and_edges(rng):
    from tests.ljpeg_encoder import encode
    full = rng.integers(0, 65536, (8, 12), dtype=np.uint16)      # 16-bit: differences wrap modulo 2^16, category 16 occurs
    full[0, 0], full[0, 1], full[1, 0], full[1, 1] = 0, 32768, 65535, 0
    assert np.array_equal(ingest.ljpeg_decode(encode(full, precision=16, predictor=1)), full)
    assert np.array_equal(ingest.ljpeg_decode(encode(full, components=4, precision=16, predictor=7)), full)
    twelve = rng.integers(0, 4096, (10, 16), dtype=np.uint16)
    assert np.array_equal(ingest.ljpeg_decode(encode(twelve, components=2, precision=12, predictor=6, restart_rows=3)), twelve)
    assert np.array_equal(ingest.ljpeg_decode(encode(twelve, precision=12, predictor=4, restart_rows=1)), twelve)
    pt = (twelve >> 2) << 2                                       # point transform 2: the low bits are not coded
    assert np.array_equal(ingest.ljpeg_decode(encode(pt, precision=12, predictor=1, point_transform=2)), pt)
    flat = np.full((5, 7), 0xff, np.uint16)                       # all-equal samples; 0xFF bytes in the stream get stuffed
    assert np.array_equal(ingest.ljpeg_decode(encode(flat, precision=8)), flat)
    good = encode(twelve, precision=12)
    for bad in (good[:40], b"\xff\xd8\xff\xd9", good.replace(b"\xff\xc3", b"\xff\xc1", 1), b"nope", good[:-len(good) // 2]):
        with pytest.raises(ValueError, match="Failed to decode RAW"):
            out = ingest.ljpeg_decode(bad)
            if bad is good[:-len(good) // 2]:
                assert out is None


def _write_ljpeg_dng(path, cfa, tile=None, comps=2, endian="<"):
    """A CFA DNG whose raw image is lossless-JPEG compressed: tiles (TileWidth x TileLength, edge tiles padded) or one strip."""
    from tests.ljpeg_encoder import encode
    h, w = cfa.shape
    e = endian
    if tile:
        tw, th = tile
        across, down = (w + tw - 1) // tw, (h + th - 1) // th
        padded = np.zeros((down * th, across * tw), np.uint16)
        padded[:h, :w] = cfa
        padded[h:, :] = padded[h - 1:h, :]                        # what encoders do with the padding: replicate
        padded[:, w:] = padded[:, w - 1:w]
        streams = [encode(padded[j * th:(j + 1) * th, i * tw:(i + 1) * tw], components=comps, precision=14, predictor=1)
                   for j in range(down) for i in range(across)]
    else:
        streams = [encode(cfa, components=comps, precision=14, predictor=1)]

    def entry(tag, typ, vals, blob_off):
        fmt = {3: "H", 4: "I"}[typ]
        raw = struct.pack(e + fmt * len(vals), *vals)
        if len(raw) <= 4:
            return struct.pack(e + "HHI", tag, typ, len(vals)) + raw.ljust(4, b"\0"), b""
        return struct.pack(e + "HHII", tag, typ, len(vals), blob_off), raw

    def build(entries, base):
        body, blobs, off = b"", b"", base + 2 + 12 * len(entries) + 4
        for tag, typ, vals in sorted(entries):
            ent, blob = entry(tag, typ, vals, off + len(blobs))
            body += ent
            blobs += blob
        return struct.pack(e + "H", len(entries)) + body + struct.pack(e + "I", 0) + blobs

    ents = [(256, 4, [w]), (257, 4, [h]), (258, 3, [14]), (259, 3, [7]), (262, 3, [32803]), (277, 3, [1])]
    if tile:
        ents += [(322, 4, [tile[0]]), (323, 4, [tile[1]]), (324, 4, [0] * len(streams)), (325, 4, [len(s) for s in streams])]
    else:
        ents += [(278, 4, [h]), (273, 4, [0]), (279, 4, [len(streams[0])])]
    data_off = 8 + len(build(ents, 8))
    offs, o = [], data_off
    for s_ in streams:
        offs.append(o)
        o += len(s_)
    ents = [(t, ty, (offs if t in (324, 273) else v)) for t, ty, v in ents]
    header = (b"II" if e == "<" else b"MM") + struct.pack(e + "HI", 42, 8)
    with open(path, "wb") as fh:
        fh.write(header + build(ents, 8) + b"".join(streams))


@pytest.mark.parametrize("layout", ["tiles", "strip"])
def test_ljpeg_compressed_dng(tmp_path, rng, layout):
    cfa = rng.integers(0, 1 << 14, (21, 38), dtype=np.uint16)    # 38 x 21 is no multiple of the 16 x 8 tile: edge tiles crop
    path = tmp_path / f"{layout}.dng"
    _write_ljpeg_dng(path, cfa, tile=(16, 8) if layout == "tiles" else None, endian=">" if layout == "strip" else "<")
    r = ingest.load_dng(str(path))
    assert (r.width, r.height) == (38, 21) and np.array_equal(r.data.reshape(21, 38), cfa)
    assert r.wb_multipliers == [1.0, 1.0, 1.0, 1.0] and r.color_matrix == list(ra.IDENTITY_MATRIX)   # no metadata tags: neutral
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.load_dng_uncompressed(str(path))                   # the strict entry point still refuses compressed data
    buf = bytearray(path.read_bytes())
    k = buf.find(b"\xff\xda") + 40                                # inside the first stream's entropy-coded data
    buf[k:k + 24] = b"\x55" * 24
    (tmp_path / "bad.dng").write_bytes(bytes(buf))
    try:
        bad = ingest.load_dng(str(tmp_path / "bad.dng"))
        assert not np.array_equal(bad.data.reshape(21, 38), cfa)  # either an error or different samples, never a crash
    except ValueError as exc:
        assert "Failed to decode RAW" in str(exc)


def test_ljpeg_decoder_survives_damaged_streams(rng):
    """The decoder is native code fed from files: 600 random mutations (byte flips, truncations, marker injections) of valid
    streams must each end in the "Failed to decode RAW" error or in an array of the right shape -- never in a crash."""
    from tests.ljpeg_encoder import encode
    a = rng.integers(0, 4096, (12, 20), dtype=np.uint16)
    streams = [encode(a, components=c, precision=12, predictor=p, restart_rows=r) for c, p, r in ((1, 1, 0), (2, 4, 0), (2, 7, 4))]
    outcomes = {"error": 0, "array": 0}
    for it in range(600):
        b = bytearray(streams[it % 3])
        kind = it % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            b = b[:int(rng.integers(3, len(b)))]
        elif kind == 2:
            k = int(rng.integers(2, len(b) - 2))
            b[k:k + 2] = bytes([0xff, int(rng.integers(0xc0, 0xff))])
        else:
            k = int(rng.integers(2, len(b)))
            b[k:k] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        try:
            out = ingest.ljpeg_decode(bytes(b))
            assert out.ndim == 2 and out.dtype == np.uint16
            outcomes["array"] += 1
        except ValueError as exc:
            assert "Failed to decode RAW" in str(exc) or "reshape" in str(exc), exc
            outcomes["error"] += 1
    assert outcomes["error"] > 100, outcomes


def test_ljpeg_decoder_under_sanitizers(tmp_path, rng):
    """The same kind of corpus through the decoder compiled with AddressSanitizer + UBSan (CPU build): 2000 damaged streams,
    a destination deliberately too small for some frames.  A single out-of-bounds access would abort the child."""
    import os
    import subprocess
    from tests.ljpeg_encoder import encode
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "fuzz_ljpeg"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(root, "raweditor_amd", "csrc"), os.path.join(root, "tests", "cpp", "fuzz_ljpeg.cpp"),
                    "-o", str(exe)], check=True)
    a = rng.integers(0, 65536, (24, 40), dtype=np.uint16)
    seeds = [encode(a[:12, :20], precision=16), encode(a, components=2, precision=16, predictor=5),
             encode(a, components=4, precision=16, predictor=7, restart_rows=6), encode(a[:8, :8] >> 4, precision=12, point_transform=1)]
    with open(tmp_path / "corpus.bin", "wb") as fh:
        for it in range(2000):
            b = bytearray(seeds[it % len(seeds)])
            for _ in range(int(rng.integers(0, 8))):
                op = int(rng.integers(0, 4))
                if op == 0 and len(b) > 4:
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
                elif op == 1 and len(b) > 8:
                    del b[int(rng.integers(4, len(b))):]
                elif op == 2 and len(b) > 6:
                    k = int(rng.integers(2, len(b) - 2))
                    b[k:k + 2] = bytes([0xff, int(rng.integers(0xc0, 0x100))])
                elif len(b) > 4:
                    k = int(rng.integers(2, len(b)))
                    b[k:k] = bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
            fh.write(struct.pack("<I", len(b)) + bytes(b))
    out = subprocess.run([str(exe), str(tmp_path / "corpus.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    assert "decoded" in out.stdout


# ---- round 3: the advisor's findings on the ingest side ---------------------------------------------------------------
def test_ljpeg_truncated_stream_declaring_a_huge_frame_fails_fast_and_small(rng):
    """A 100-byte tile must not be able to demand gigabytes: the frame size comes from the decoder's own header walk, is
    checked against what the TIFF tile can hold BEFORE anything is allocated, and a stream that ends early fails at the
    row where it ran dry instead of being decoded to the end of the declared frame out of injected zero bits."""
    import time
    import tracemalloc
    from tests.ljpeg_encoder import encode
    good = bytearray(encode(rng.integers(0, 4096, (8, 16), dtype=np.uint16), precision=12))
    i = good.find(b"\xff\xc3")
    huge = bytearray(good)
    huge[i + 5:i + 9] = struct.pack(">HH", 65535, 32767)          # SOF3 now declares 65535 x 32767 samples (4 GiB of u16)
    tracemalloc.start()
    t0 = time.time()
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.ljpeg_decode(bytes(huge), max_samples=16 * 8)      # the tile holds 128 samples
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.ljpeg_decode(bytes(huge))                          # no tile size known: the module's own ceiling
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.ljpeg_decode(bytes(huge), max_samples=1 << 31)     # a single-strip DNG's w*h: a caller's limit never LOOSENS the
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert time.time() - t0 < 2.0 and peak < (8 << 20), (time.time() - t0, peak)
    #                                                               module ceiling, nor the stream's own bound (>= 1 bit per sample)
    # within every bound but longer than the data: must stop at the first dry row, not after the declared rows of zeros
    rows = 8 * len(good) // 16 - 1                                # as many rows as the stream's byte count could possibly hold
    assert rows > 16
    tall = bytearray(good)
    tall[i + 5:i + 7] = struct.pack(">H", rows)
    t0 = time.time()
    with pytest.raises(ValueError, match="truncated"):
        ingest.ljpeg_decode(bytes(tall))
    assert time.time() - t0 < 1.0
    tall[i + 5:i + 7] = struct.pack(">H", 4096)                   # more samples than bits in the stream: refused outright
    with pytest.raises(ValueError, match="can hold"):
        ingest.ljpeg_decode(bytes(tall))


def test_ljpeg_frame_header_is_found_by_walking_segments_not_by_searching_bytes(rng):
    """0xFF 0xC3 inside an APPn / COM payload ahead of the real SOF3 must not size the allocation."""
    from tests.ljpeg_encoder import encode
    a = rng.integers(0, 4096, (6, 10), dtype=np.uint16)
    good = encode(a, precision=12)
    payload = b"junk\xff\xc3" + b"\x7f" * 12                       # would read as a 32639 x 32639 x 127 frame
    com = b"\xff\xfe" + struct.pack(">H", 2 + len(payload)) + payload
    assert np.array_equal(ingest.ljpeg_decode(good[:2] + com + good[2:]), a)


def test_ljpeg_restart_interval_must_be_whole_lines(rng):
    from tests.ljpeg_encoder import encode
    a = rng.integers(0, 4096, (6, 10), dtype=np.uint16)
    s = bytearray(encode(a, precision=12, restart_rows=2))
    i = s.find(b"\xff\xdd")
    assert i > 0 and struct.unpack_from(">H", s, i + 4)[0] == 20
    assert np.array_equal(ingest.ljpeg_decode(bytes(s)), a)
    s[i + 4:i + 6] = struct.pack(">H", 15)                        # 1.5 lines: would be predicted wrongly, silently
    with pytest.raises(ValueError, match="Failed to decode RAW"):
        ingest.ljpeg_decode(bytes(s))


def test_ljpeg_sizes_are_reported_before_the_capacity_check(rng):
    import ctypes as C
    from raweditor_amd import _lib
    from tests.ljpeg_encoder import encode
    s = encode(rng.integers(0, 4096, (5, 8), dtype=np.uint16), components=2, precision=12)
    dims = [C.c_uint32(99) for _ in range(4)]
    src = (C.c_uint8 * len(s)).from_buffer_copy(s)
    rc = _lib.lib().rd_ljpeg_decode(src, len(s), None, 0, *[C.byref(d) for d in dims])
    assert rc != 0 and [d.value for d in dims] == [4, 5, 2, 12]
    rc = _lib.lib().rd_ljpeg_decode(src, 20, None, 0, *[C.byref(d) for d in dims])      # fails before the frame is known
    assert rc != 0 and [d.value for d in dims] == [0, 0, 0, 0]


def test_unpack_bits_is_chunked_and_exact_at_scale(rng, monkeypatch):
    """10 / 12 / 14-bit unpacking by whole-byte groups, a bounded block of rows at a time: exact against the bit-by-bit
    definition, also when a row is shorter than its last group and when the block size forces many passes."""
    monkeypatch.setattr(ingest, "_UNPACK_CHUNK_BYTES", 64)        # several passes even for tiny inputs
    for bits in (10, 12, 14):
        for w in (1, 2, 3, 4, 5, 13, 64):
            cfa = rng.integers(0, 1 << bits, (9, w), dtype=np.uint16)
            assert np.array_equal(ingest.unpack_bits(_pack_rows(cfa, bits), w, bits), cfa), (bits, w)
    monkeypatch.undo()
    import tracemalloc
    cfa = rng.integers(0, 4096, (512, 2048), dtype=np.uint16)     # 1 MP
    packed = _pack_rows(cfa, 12)
    tracemalloc.start()
    got = ingest.unpack_bits(packed, 2048, 12)
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert np.array_equal(got, cfa)
    assert peak < 16 * cfa.size, peak                             # round 2's form needed ~100 bytes per sample


def test_dng_subifd_offsets_of_tiff_type_13(tmp_path, rng):
    """Tag 330 written with TIFF type 13 (IFD) instead of LONG -- what many writers do -- must be followed."""
    cfa = rng.integers(0, 4096, (6, 8), dtype=np.uint16)
    path = tmp_path / "t13.dng"
    _write_dng(path, cfa, "<", strips=1)
    buf = bytearray(path.read_bytes())
    (n,) = struct.unpack_from("<H", buf, 8)
    hit = False
    for k in range(n):
        tag, typ = struct.unpack_from("<HH", buf, 10 + 12 * k)
        if tag == 330:
            assert typ == 4
            struct.pack_into("<H", buf, 10 + 12 * k + 2, 13)
            hit = True
    assert hit
    path.write_bytes(bytes(buf))
    r = ingest.load_dng(str(path))
    assert np.array_equal(r.data.reshape(6, 8), cfa)


def test_to_json_is_serde_jsons_text_for_every_magnitude():
    """EditParams::to_json is serde_json::to_string (edit.rs:105-107), which writes an f32 with ryu's format32: positional
    between 1e-6 and 1e13, exponent form outside ("1e-7", "1e20": no '+', no padding), "-0.0" for negative zero.  Rows this
    library writes into `edits.settings_json` must be the bytes the app would have written."""
    from raweditor_amd import EditParams
    from raweditor_amd.edit import ryu_f32
    vectors = [(1e-7, "1e-7"), (1e20, "1e20"), (-0.0, "-0.0"), (0.0, "0.0"), (0.3, "0.3"), (1.0, "1.0"), (1234567.0, "1234567.0"),
               (1e13, "1e13"), (1e12, "1000000000000.0"), (1.5e-7, "1.5e-7"), (16777216.0, "16777216.0"), (1e-5, "0.00001"),
               (1e-6, "0.000001"), (9.9e-7, "9.9e-7"), (12.34, "12.34"), (3.4028235e38, "3.4028235e38"), (-2.5, "-2.5"),
               (0.001234, "0.001234"), (1.17549435e-38, "1.1754944e-38"), (1e-45, "1e-45"), (123456.79, "123456.79"), (100.0, "100.0"),
               (-1e-7, "-1e-7"), (0.1, "0.1"), (25.0, "25.0"), (0.005, "0.005")]
    for v, text in vectors:
        assert ryu_f32(v) == text, (v, ryu_f32(v), text)
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 0x7f800000, 20000, dtype=np.uint32) | (rng.integers(0, 2, 20000, dtype=np.uint32) << 31)
    for v in bits.view(np.float32):                               # every text reads back as the same float, and is the shortest that does
        t = ryu_f32(v)
        assert np.float32(float(t)) == v, (v, t)
        assert "+" not in t and "E" not in t and (("e" in t) != ("." in t) or "e" in t)
    p = EditParams(exposure=1e-7, contrast=1e20, tint=-0.0, blacks=0.005)
    text = p.to_json()
    assert text == ('{"exposure":1e-7,"contrast":1e20,"highlights":0.0,"shadows":0.0,"whites":1.0,"blacks":0.005,'
                    '"vibrance":0.0,"saturation":0.0,"temperature":0.0,"tint":-0.0}')
    q = EditParams.from_json(text)
    assert q == p and np.signbit(np.float32(q.tint))
    assert EditParams().to_json() == ('{"exposure":0.0,"contrast":0.0,"highlights":0.0,"shadows":0.0,"whites":1.0,"blacks":0.0,'
                                      '"vibrance":0.0,"saturation":0.0,"temperature":0.0,"tint":0.0}')      # edit.rs:135-150's shape


def test_init_schema_is_the_apps(tmp_path):
    """library.rs:52-121: a catalog created here has every column and index the app creates (the Phase-28 cache tiers and
    file_status arrive by ALTER TABLE there, so a second run must not fail either), with the same defaults."""
    conn = sqlite3.connect(tmp_path / "library.db")
    catalog.init_schema(conn)
    catalog.init_schema(conn)                                     # idempotent, like the app's every start
    cols = {r[1]: (r[2], r[3], r[4], r[5]) for r in conn.execute("PRAGMA table_info(images)")}
    assert list(cols) == ["id", "path", "filename", "width", "height", "imported_at", "cache_status", "cache_path_thumb",
                          "cache_path_instant", "cache_path_working", "file_status"]
    assert cols["cache_status"][2] == "'pending'" and cols["file_status"][2] == "'exists'" and cols["path"][1] == 1
    assert cols["cache_path_thumb"][0] == "TEXT" and cols["imported_at"] == ("INTEGER", 1, None, 0)
    ecols = [r[1] for r in conn.execute("PRAGMA table_info(edits)")]
    assert ecols == ["id", "image_id", "settings_json"]
    idx = {r[1] for r in conn.execute("PRAGMA index_list(images)")} | {r[1] for r in conn.execute("PRAGMA index_list(edits)")}
    assert {"idx_images_imported_at", "idx_edits_image_id", "idx_images_cache_status"} <= idx
    fk = conn.execute("PRAGMA foreign_key_list(edits)").fetchall()
    assert fk and fk[0][2] == "images" and fk[0][6] == "CASCADE"
    # the statements the app itself runs against such a file (library.rs: insert_image, update cache paths, file_status)
    conn.execute("INSERT INTO images (path, filename, width, height, imported_at) VALUES ('/a.nef', 'a.nef', 6016, 4016, 1)")
    conn.execute("UPDATE images SET cache_path_thumb = '/c/t.jpg', cache_path_instant = '/c/i.jpg', cache_path_working = '/c/w.jpg', "
                 "cache_status = 'cached' WHERE id = 1")
    row = conn.execute("SELECT cache_status, file_status, cache_path_working FROM images WHERE id = 1").fetchone()
    assert row == ("cached", "exists", "/c/w.jpg")
