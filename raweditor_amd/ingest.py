"""CFA ingest without rawloader -- the step BEFORE the path.

The reference decodes RAW files with the third-party `rawloader` crate (src/raw/loader.rs:50-54); its
source is not in the tree, so NEF decoding is out of scope and this module does NOT claim rawloader parity
("parity unpinned": there is nothing of the reference's to check a decoder against).  What IS restated
exactly is everything loader.rs does AFTER the decode (src/raw/loader.rs:57-152), which defines the
`RawDataResult` the develop path is fed:

  * integer samples are taken as they are; float samples become clamp(v*65535, 0, 65535) as u16 (:62-73);
  * white balance: 4 coefficients, or 3 with G2 := G, or neutral; divided by max(G, 0.001); a non-finite
    or non-positive G2 falls back to G (:78-110);
  * colour matrix: the first 3 columns of the first 3 rows of xyz_to_cam if [0][0] or [1][1] is non-zero,
    else identity (:115-134).

Two containers are read: a headerless u16 plane (dimensions from the caller / the catalog row), and
uncompressed CFA DNG/TIFF (Compression = 1, one sample per pixel, 8 / 10 / 12 / 14 bits packed or 16 bits).  No pixel arithmetic happens
here -- the samples go to HBM untouched.
"""
from __future__ import annotations

import dataclasses
import math
import os
import struct
from typing import List, Optional, Sequence

import numpy as np


@dataclasses.dataclass
class RawDataResult:
    """src/raw/loader.rs:11-19."""
    data: np.ndarray                 # uint16, length width*height, row-major
    width: int
    height: int
    wb_multipliers: List[float]      # [R/G, 1, B/G, G2/G]
    color_matrix: List[float]        # xyz_to_cam 3x3 row-major (or identity)
    # Extras the reference's struct does not carry (loader.rs drops them: its shader divides by 4096 whatever the sensor,
    # shaders.rs:167-168, and hard-codes RGGB, :127-155).  Filled from the container when it states them; the defaults are
    # what the reference assumes, and nothing applies them unless the caller passes black_level on (rd_frame.black_level).
    black_level: int = 0             # one level for the whole plane: the mean of the container's per-channel levels, rounded
    white_level: int = 65535
    cfa_pattern: str = "RGGB"        # 2x2 repeat, row-major; the develop path demosaics RGGB only
    black_levels: List[float] = dataclasses.field(default_factory=list)    # as stated by the container (1, 2x2 ... values)


def normalise_wb(wb_coeffs: Sequence[float]) -> List[float]:
    """loader.rs:78-110, f32 arithmetic."""
    f = np.float32
    c = [f(x) for x in wb_coeffs]
    if len(c) >= 4:
        wb = c[:4]
    elif len(c) >= 3:
        wb = [c[0], c[1], c[2], c[1]]
    else:
        wb = [f(1.0)] * 4
    g = max(wb[1], f(0.001))
    g2 = wb[3] / g if (math.isfinite(float(wb[3])) and wb[3] > 0) else wb[1] / g
    return [float(wb[0] / g), float(wb[1] / g), float(wb[2] / g), float(g2)]


def extract_matrix(xyz_to_cam: Optional[Sequence[Sequence[float]]]) -> List[float]:
    """loader.rs:115-134: rows 0..2, columns 0..2 of the [3+][3 or 4] matrix, or identity."""
    if xyz_to_cam is not None and (xyz_to_cam[0][0] != 0.0 or xyz_to_cam[1][1] != 0.0):
        return [float(np.float32(xyz_to_cam[r][c])) for r in range(3) for c in range(3)]
    return [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]


def samples_to_u16(values: np.ndarray) -> np.ndarray:
    """loader.rs:62-73."""
    values = np.asarray(values)
    if values.dtype.kind == "f":
        v = values.astype(np.float32) * np.float32(65535.0)
        v = np.nan_to_num(v, nan=0.0, posinf=65535.0, neginf=0.0)     # Rust's saturating `as u16`: NaN -> 0
        return np.clip(v, np.float32(0.0), np.float32(65535.0)).astype(np.uint16)
    return values.astype(np.uint16, copy=False)


_UNPACK_GROUP = {10: (5, 4), 12: (3, 2), 14: (7, 4)}      # bits -> (bytes, samples) of the smallest whole-byte group
_UNPACK_CHUNK_BYTES = 8 << 20                              # packed bytes per pass: bounds the temporaries (~6x that)


def unpack_bits(rows: np.ndarray, width: int, bits: int) -> np.ndarray:
    """(n_rows, row_bytes) uint8, samples of `bits` bits packed MSB first -> (n_rows, width) uint16.
    Whole-byte groups (5 bytes = 4 x 10 bit, 3 = 2 x 12, 7 = 4 x 14) are gathered into one u64 and split with shifts, a
    bounded block of rows at a time: a 24 MP 12-bit strip needs tens of MB of temporaries, not gigabytes."""
    if bits == 8:
        return rows[:, :width].astype(np.uint16)
    if bits == 16:
        return (rows[:, 0:2 * width:2].astype(np.uint16) << 8) | rows[:, 1:2 * width:2]
    if bits not in _UNPACK_GROUP:
        raise ValueError(f"unsupported packed sample width {bits}")
    gb, gs = _UNPACK_GROUP[bits]
    n_rows, row_bytes = rows.shape
    groups = (width + gs - 1) // gs
    need = groups * gb
    out = np.empty((n_rows, width), np.uint16)
    step = max(1, _UNPACK_CHUNK_BYTES // max(1, need))
    mask = np.uint64((1 << bits) - 1)
    for r0 in range(0, n_rows, step):
        blk = rows[r0:r0 + step]
        if row_bytes < need:                                 # the last group of a row may lack its padding bytes
            blk = np.concatenate([blk, np.zeros((blk.shape[0], need - row_bytes), np.uint8)], axis=1)
        g = blk[:, :need].reshape(blk.shape[0], groups, gb)
        acc = np.zeros((blk.shape[0], groups), np.uint64)
        for k in range(gb):
            acc = (acc << np.uint64(8)) | g[:, :, k]
        vals = np.empty((blk.shape[0], groups, gs), np.uint16)
        for k in range(gs):
            vals[:, :, k] = (acc >> np.uint64(bits * (gs - 1 - k))) & mask
        out[r0:r0 + step] = vals.reshape(blk.shape[0], groups * gs)[:, :width]
    return out


def load_raw_u16(path: str, width: int, height: int, wb_coeffs: Sequence[float] = (),
                 xyz_to_cam: Optional[Sequence[Sequence[float]]] = None, memmap: bool = True) -> RawDataResult:
    """Headerless little-endian u16 plane.  Errors mirror loader.rs:46-48 ("File not found: ...")."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"File not found: {path}")
    n = int(width) * int(height)
    if os.path.getsize(path) != 2 * n:
        raise ValueError(f"{path}: {os.path.getsize(path)} bytes, expected {2 * n} for {width}x{height} u16")
    data = np.memmap(path, dtype="<u2", mode="r", shape=(n,)) if memmap else np.fromfile(path, dtype="<u2")
    return RawDataResult(data, int(width), int(height), normalise_wb(wb_coeffs), extract_matrix(xyz_to_cam))


# ---- minimal TIFF / DNG ----------------------------------------------------------------------------------
# TIFF field types; 13 = IFD (an offset, read like LONG: what many writers use for the SubIFDs tag 330)
_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 13: 4}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 11: "f", 12: "d", 13: "I"}
TAG_WIDTH, TAG_LENGTH, TAG_BITS, TAG_COMPRESSION, TAG_PHOTOMETRIC = 256, 257, 258, 259, 262
TAG_STRIP_OFFSETS, TAG_SPP, TAG_ROWS_PER_STRIP, TAG_STRIP_BYTES, TAG_SUBIFD = 273, 277, 278, 279, 330
TAG_TILE_WIDTH, TAG_TILE_LENGTH, TAG_TILE_OFFSETS, TAG_TILE_BYTES = 322, 323, 324, 325
TAG_COLOR_MATRIX1, TAG_AS_SHOT_NEUTRAL = 50721, 50728
TAG_CFA_REPEAT_DIM, TAG_CFA_PATTERN, TAG_BLACK_LEVEL, TAG_WHITE_LEVEL = 33421, 33422, 50714, 50717
COMPRESSION_NONE, COMPRESSION_LJPEG = 1, 7
PHOTOMETRIC_CFA = 32803


MAX_IFDS = 64                       # a DNG has a handful; a file that chains more is damaged or hostile


def _decode_error(msg: str) -> ValueError:
    return ValueError(f"Failed to decode RAW: {msg}")           # the reference's error text (loader.rs:50-54)


def _read_ifd(buf: bytes, off: int, e: str):
    """One IFD.  The file is not trusted: every offset and size is checked against the buffer."""
    if off + 2 > len(buf):
        raise _decode_error(f"IFD offset {off} is outside the file")
    (n,) = struct.unpack_from(e + "H", buf, off)
    if off + 2 + 12 * n + 4 > len(buf):
        raise _decode_error(f"IFD at {off} with {n} entries runs past the end of the file")
    tags = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack_from(e + "HHI4s", buf, off + 2 + 12 * i)
        size = _TYPE_SIZE.get(typ, 1) * cnt
        if size <= 4:
            raw = val[:size]
        else:
            (vo,) = struct.unpack(e + "I", val)
            if vo + size > len(buf):
                raise _decode_error(f"tag {tag}: value ({size} bytes at {vo}) is outside the file")
            raw = buf[vo:vo + size]
        if typ in (5, 10):                                  # (S)RATIONAL
            ints = struct.unpack(e + ("I" if typ == 5 else "i") * (2 * cnt), raw)
            vals = [ints[2 * k] / ints[2 * k + 1] if ints[2 * k + 1] else 0.0 for k in range(cnt)]
        elif typ in _TYPE_FMT:
            vals = list(struct.unpack(e + _TYPE_FMT[typ] * cnt, raw))
        else:
            vals = [raw]
        tags[tag] = vals
    (nxt,) = struct.unpack_from(e + "I", buf, off + 2 + 12 * n)
    return tags, nxt


LJPEG_MAX_SAMPLES = 1 << 28          # 268 M samples (512 MiB of u16) per stream unless the caller knows the tile size


def ljpeg_decode(stream: bytes, max_samples: Optional[int] = None) -> np.ndarray:
    """One lossless-JPEG stream (ITU-T T.81 SOF3) -> (height, width * components) uint16, through librawdev's host-side
    decoder (rd_ljpeg_decode).  Errors carry the reference's "Failed to decode RAW" text.
    The stream is not trusted: the frame size comes from the decoder's own marker walk (a first call with capacity 0 --
    not a byte search, which an APPn / COM payload could satisfy), and a frame that declares more than `max_samples`
    samples (the TIFF tile / strip size when called from load_dng) is refused BEFORE anything is allocated; a stream that
    ends early fails at the row where it ran dry."""
    import ctypes as C
    from . import _lib
    buf = bytes(stream)
    if len(buf) < 12 or buf[:2] != b"\xff\xd8":
        raise _decode_error("not a JPEG stream")
    # the module ceiling always holds; a caller's limit can only tighten it.  And the stream bounds itself: a Huffman-coded
    # sample costs at least one bit, so no stream of len(buf) bytes holds more than 8 * len(buf) samples.
    limit = min(LJPEG_MAX_SAMPLES, 8 * len(buf))
    if max_samples is not None:
        limit = min(limit, int(max_samples))
    dims = [C.c_uint32() for _ in range(4)]
    src = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
    L = _lib.lib()
    rc = L.rd_ljpeg_decode(src, len(buf), None, 0, *[C.byref(d) for d in dims])      # sizes only: ends with "too small"
    n = int(dims[0].value) * int(dims[1].value) * int(dims[2].value)
    if n <= 0:                                               # the header walk itself failed (rc says why)
        raise ValueError(L.rd_last_error().decode("utf-8", "replace") if rc != 0 else "Failed to decode RAW: empty frame")
    if n > limit:
        raise _decode_error(f"lossless-JPEG frame declares {dims[0].value}x{dims[1].value}x{dims[2].value} samples, "
                            f"more than the {limit} its tile / strip / {len(buf)}-byte stream can hold")
    out = np.empty(n, np.uint16)
    rc = L.rd_ljpeg_decode(src, len(buf), out.ctypes.data_as(C.c_void_p), n, *[C.byref(d) for d in dims])
    if rc != 0:
        raise ValueError(L.rd_last_error().decode("utf-8", "replace"))
    hh, ww = dims[1].value, dims[0].value * dims[2].value
    return out[:hh * ww].reshape(hh, ww)


def load_dng_uncompressed(path: str) -> RawDataResult:
    """Uncompressed (Compression = 1) single-sample CFA image (8 ... 16 bits) from a DNG/TIFF; anything else is an error."""
    return load_dng(path, allow_compressed=False)


def load_dng(path: str, allow_compressed: bool = True) -> RawDataResult:
    """The CFA image of a DNG/TIFF: uncompressed strips (8 ... 16 bits), or lossless-JPEG strips / tiles (Compression = 7,
    what Adobe's DNG converter and DNG-writing cameras produce).  The reference hands every RAW file to `rawloader`
    (loader.rs:50-54); its source is absent, so there is no parity claim against it -- lossless JPEG decodes exactly."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"File not found: {path}")
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 8:
        raise _decode_error("not a TIFF container")
    if buf[:2] == b"II":
        e = "<"
    elif buf[:2] == b"MM":
        e = ">"
    else:
        raise ValueError("Failed to decode RAW: not a TIFF container")
    magic, off = struct.unpack_from(e + "HI", buf, 2)
    if magic != 42:
        raise ValueError("Failed to decode RAW: bad TIFF magic")
    todo, ifds, seen = [off], [], set()
    while todo:
        o = todo.pop(0)
        if not o or o in seen:                                  # 0 ends a chain; an offset seen before is a loop
            continue
        seen.add(o)
        if len(ifds) >= MAX_IFDS:
            raise _decode_error(f"more than {MAX_IFDS} IFDs")
        tags, nxt = _read_ifd(buf, o, e)
        ifds.append(tags)
        todo.extend(int(x) for x in tags.get(TAG_SUBIFD, []) if isinstance(x, int))
        todo.append(nxt)
    raw = next((t for t in ifds if t.get(TAG_PHOTOMETRIC, [None])[0] == PHOTOMETRIC_CFA), None)
    if raw is None:
        raise ValueError("Failed to decode RAW: no CFA image in the file")
    bits = raw.get(TAG_BITS, [0])[0]
    compression = raw.get(TAG_COMPRESSION, [1])[0]
    if raw.get(TAG_SPP, [1])[0] != 1:
        raise _decode_error("only single-sample CFA data is supported")
    if compression == COMPRESSION_LJPEG and allow_compressed:
        data, w, h = _read_ljpeg_image(buf, raw)
        return _finish(ifds, raw, data, w, h)
    if compression != COMPRESSION_NONE or bits not in (8, 10, 12, 14, 16):
        raise ValueError("Failed to decode RAW: only uncompressed single-sample CFA data of 8, 10, 12, 14 or 16 bits is supported"
                         + (" (or lossless-JPEG compressed)" if allow_compressed else ""))
    for need in (TAG_WIDTH, TAG_LENGTH, TAG_STRIP_OFFSETS, TAG_STRIP_BYTES):
        if not raw.get(need):
            raise _decode_error(f"CFA image lacks tag {need}")
    w, h = raw[TAG_WIDTH][0], raw[TAG_LENGTH][0]
    if len(raw[TAG_STRIP_OFFSETS]) != len(raw[TAG_STRIP_BYTES]):
        raise _decode_error("StripOffsets and StripByteCounts differ in length")
    for o, nbytes in zip(raw[TAG_STRIP_OFFSETS], raw[TAG_STRIP_BYTES]):
        if o + nbytes > len(buf) or (bits == 16 and nbytes % 2):
            raise _decode_error(f"strip ({nbytes} bytes at {o}) is outside the file or not whole samples")
    if bits == 16:
        parts = [np.frombuffer(buf, dtype=e + "u2", count=nbytes // 2, offset=o)
                 for o, nbytes in zip(raw[TAG_STRIP_OFFSETS], raw[TAG_STRIP_BYTES])]
    else:
        # 8 / 10 / 12 / 14 bits: samples packed most-significant-bit first whatever the file's byte order (TIFF FillOrder 1,
        # DNG spec "BitsPerSample"), every row starting on a byte boundary
        row_bytes = (w * bits + 7) // 8
        parts = []
        for o, nbytes in zip(raw[TAG_STRIP_OFFSETS], raw[TAG_STRIP_BYTES]):
            if nbytes % row_bytes:
                raise _decode_error(f"strip of {nbytes} bytes is not whole rows of {row_bytes} bytes")
            parts.append(unpack_bits(np.frombuffer(buf, dtype=np.uint8, count=nbytes, offset=o).reshape(-1, row_bytes), w, bits).reshape(-1))
    data = np.concatenate(parts).astype(np.uint16) if len(parts) > 1 else parts[0].astype(np.uint16)
    if data.size != w * h:
        raise ValueError(f"Failed to decode RAW: {data.size} samples for {w}x{h}")
    return _finish(ifds, raw, data, w, h)


def _read_ljpeg_image(buf: bytes, raw: dict):
    """Compression = 7: every strip or tile is one lossless-JPEG stream whose samples, read in stream order, are the
    strip's / tile's CFA samples row by row (encoders commonly declare two components of half the width -- or two rows per
    JPEG line -- to give the predictor same-colour neighbours; the sample order is unchanged by that)."""
    for need in (TAG_WIDTH, TAG_LENGTH):
        if not raw.get(need):
            raise _decode_error(f"CFA image lacks tag {need}")
    w, h = int(raw[TAG_WIDTH][0]), int(raw[TAG_LENGTH][0])
    if w <= 0 or h <= 0 or w * h > (1 << 31):
        raise _decode_error(f"implausible image size {w}x{h}")
    if raw.get(TAG_TILE_OFFSETS):
        for need in (TAG_TILE_WIDTH, TAG_TILE_LENGTH, TAG_TILE_BYTES):
            if not raw.get(need):
                raise _decode_error(f"tiled CFA image lacks tag {need}")
        tw, th = int(raw[TAG_TILE_WIDTH][0]), int(raw[TAG_TILE_LENGTH][0])
        offs, sizes = raw[TAG_TILE_OFFSETS], raw[TAG_TILE_BYTES]
    else:
        for need in (TAG_STRIP_OFFSETS, TAG_STRIP_BYTES):
            if not raw.get(need):
                raise _decode_error(f"CFA image lacks tag {need}")
        tw, th = w, int(raw.get(TAG_ROWS_PER_STRIP, [h])[0]) or h
        th = min(th, h)
        offs, sizes = raw[TAG_STRIP_OFFSETS], raw[TAG_STRIP_BYTES]
    if tw <= 0 or th <= 0:
        raise _decode_error("bad tile / strip size")
    across, down = (w + tw - 1) // tw, (h + th - 1) // th
    if len(offs) != len(sizes) or len(offs) < across * down:
        raise _decode_error(f"{len(offs)} tiles / strips for a {across}x{down} grid")
    # a Huffman-coded sample costs at least one bit: the strips / tiles of this file cannot hold more samples than 8 x
    # their bytes, whatever the header claims -- checked before the image is allocated
    if w * h > 8 * sum(int(x) for x in sizes[:across * down]):
        raise _decode_error(f"{w}x{h} image cannot come out of {sum(int(x) for x in sizes[:across * down])} bytes of lossless JPEG")
    out = np.zeros((h, w), np.uint16)
    for k in range(across * down):
        o, nbytes = int(offs[k]), int(sizes[k])
        if o + nbytes > len(buf) or nbytes < 4:
            raise _decode_error(f"tile {k} ({nbytes} bytes at {o}) is outside the file")
        tile = ljpeg_decode(buf[o:o + nbytes], max_samples=tw * th)      # no stream may declare more than its tile holds
        y0, x0 = (k // across) * th, (k % across) * tw
        rows = min(th, h - y0) if not raw.get(TAG_TILE_OFFSETS) else th
        if tile.size < rows * tw:
            raise _decode_error(f"tile {k} holds {tile.size} samples, expected {rows * tw}")
        tile = tile.reshape(-1)[:rows * tw].reshape(rows, tw)
        hh, ww = min(rows, h - y0), min(tw, w - x0)
        out[y0:y0 + hh, x0:x0 + ww] = tile[:hh, :ww]          # edge tiles are padded to the full tile size: crop
    return out.reshape(-1), w, h


def _finish(ifds, raw, data, w, h) -> RawDataResult:
    meta = ifds[0]
    neutral = meta.get(TAG_AS_SHOT_NEUTRAL) or raw.get(TAG_AS_SHOT_NEUTRAL)
    wb = [1.0 / x if x else 0.0 for x in neutral[:3]] if neutral else []      # multipliers = 1 / neutral
    cm = meta.get(TAG_COLOR_MATRIX1) or raw.get(TAG_COLOR_MATRIX1)
    xyz_to_cam = [cm[0:3], cm[3:6], cm[6:9]] if cm and len(cm) >= 9 else None
    res = RawDataResult(data, int(w), int(h), normalise_wb(wb), extract_matrix(xyz_to_cam))
    # DNG 1.x: BlackLevel (one value, or one per position of BlackLevelRepeatDim), WhiteLevel, CFAPattern (0 R, 1 G, 2 B)
    levels = [float(x) for x in raw.get(TAG_BLACK_LEVEL, []) if isinstance(x, (int, float)) and math.isfinite(float(x))]
    if levels:
        res.black_levels = levels
        res.black_level = int(min(max(round(sum(levels) / len(levels)), 0), 65535))
    white = [x for x in raw.get(TAG_WHITE_LEVEL, []) if isinstance(x, (int, float))]
    if white:
        res.white_level = int(min(max(round(float(white[0])), 0), 65535))
    pat = raw.get(TAG_CFA_PATTERN, [])
    if len(pat) == 1 and isinstance(pat[0], (bytes, bytearray)):              # type UNDEFINED arrives as raw bytes
        pat = list(pat[0])
    if len(pat) == 4 and raw.get(TAG_CFA_REPEAT_DIM, [2, 2])[:2] == [2, 2] and all(isinstance(c, int) and 0 <= c <= 2 for c in pat):
        res.cfa_pattern = "".join("RGB"[c] for c in pat)
    return res
