"""-m gpu: frame widths that are not a multiple of the 128-pixel tile (round 5).

The reference renders a texture of any size (shaders.rs:181-187; the loader takes the sensor's own dimensions,
loader.rs:57-58) and its JPEG export strips alpha for any width (main.rs:1777-1786).  Cameras make 6000, 8256, 5472,
7360 ... wide frames; 6016 is the exception.  Every even width >= 128 runs the export kernel's whole-tile instances with
the last tile of a row pair pulled back to end at the row's end (rd_kernels.h, RD_TILES_OVERLAP): the overlapped quads
are computed and stored twice with the same bytes and counted in the histogram once.  Checked here, bit for bit against
the oracle:

  * small frames around one, two and three tiles (130 ... 386 wide; W % 4 == 2 makes RGB8 rows and tiles start on an odd
    16-bit boundary), 1 ... 9 rows, all four surfaces + the fused histogram, through the pipeline, the batch path in both
    launch modes and the export ring;
  * whole frames at 6000 x 4000 and 8256 x 5504 (and a W % 4 == 2 strip, 6002 x 402): every byte of the f32, f16, RGBA8
    and RGB8 surfaces and the exact histogram; the same through the 8-band host render and through a multi-frame launch;
  * RD_TILES=overlap on a width that needs no overlap (the instance's degenerate case);
  * round 6: odd widths (whole quads + rd_develop_lastcol) on every entry point, and the f32 surface's shifted-window tiling
    (RD_TILES_SHIFT: every width here that is not a multiple of 4, its edge widths 250 / 251 / 374 / 375 among them).
"""
import numpy as np
import pytest

from tests.gpu_util import DevBuf, sync
from tests.helpers import CM_IDENTITY, CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu


def _surface(refc, ra, exp32, fmt):
    """The oracle's f32 surface in the byte layout of `fmt`."""
    if fmt == ra.FMT_RGBA_F32:
        return exp32.view(np.uint32)
    if fmt == ra.FMT_RGBA_F16:
        return refc.pack_f16(exp32).view(np.uint16)
    u8 = refc.pack_u8(exp32)
    return u8[..., :3] if fmt == ra.FMT_RGB_U8 else u8


def _view(ra, arr, fmt):
    return arr.view(np.uint32) if fmt == ra.FMT_RGBA_F32 else arr.view(np.uint16) if fmt == ra.FMT_RGBA_F16 else arr


def _fmts(ra):
    return (ra.FMT_RGBA_F32, ra.FMT_RGBA_F16, ra.FMT_RGBA_U8, ra.FMT_RGB_U8)


# (250 and 374: W / 2 = 62 T - 61, the f32 surface's shifted-window tiling with a last tile that owns a single quad -- rd_kernels.h,
#  RD_TILES_SHIFT; every width here that is not a multiple of 4 takes that tiling on the f32 surface)
SMALL = [(1, 130), (2, 134), (5, 190), (8, 192), (3, 202), (9, 254), (4, 258), (6, 382), (7, 386), (2, 128), (5, 320), (6, 250), (9, 374), (4, 498)]


@pytest.mark.parametrize("math", [0, 1])
def test_small_ragged_frames_every_surface(gpu_lib, refc, math):
    ra = gpu_lib
    rng = np.random.default_rng([0x52415745, 5, math])
    for h, w in SMALL:
        cfa = random_cfa(rng, h, w, 65536)
        for cm, params in ((CM_TEST, random_params(rng)), (CM_IDENTITY, {"exposure": 0.5, "contrast": 4.0})):      # general / separable
            pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, cm)
            pipe.set_math_mode(math)
            u = refc.make_uniforms(params, WB_DAYLIGHT, cm, math_mode=math)
            exp32 = refc.render_f32(cfa, u)
            exp_hist = refc.histogram(refc.pack_u8(exp32))
            for fmt in _fmts(ra):
                got, hist = pipe.render(fmt=fmt, with_histogram=True)
                assert np.array_equal(_view(ra, got, fmt), _surface(refc, ra, exp32, fmt)), (h, w, fmt, "surface")
                assert np.array_equal(hist, exp_hist), (h, w, fmt, "histogram")
                got = pipe.render(fmt=fmt)
                assert np.array_equal(_view(ra, got, fmt), _surface(refc, ra, exp32, fmt)), (h, w, fmt, "no histogram")
            pipe.close()


def _batch(ra, refc, h, w, n, fmt, bands=1, hist=True):
    rng = np.random.default_rng([0x52415745, h, w, n, fmt])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    bpp = ra.BYTES_PER_PIXEL[fmt]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * bpp) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, fmt, hist)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    be.develop(frames, row_bands=bands)
    if hist:
        be.histogram(d_hist.ptr)
    sync()
    exp_hist = np.zeros(768, np.uint64)
    dt, ch = {ra.FMT_RGBA_F32: (np.uint32, 4), ra.FMT_RGBA_F16: (np.uint16, 4), ra.FMT_RGBA_U8: (np.uint8, 4), ra.FMT_RGB_U8: (np.uint8, 3)}[fmt]
    for c, p, o in zip(cfas, params, d_out):
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
        e = refc.render_f32(c, u, nthreads=8)
        exp_hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
        assert np.array_equal(o.to_array(dt, (h, w, ch)), _surface(refc, ra, e, fmt)), (h, w, fmt)
    if hist:
        assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist), (h, w, fmt)
    be.close()
    for b in d_in + d_out + [d_hist]:
        b.free()


@pytest.fixture(params=["multi_frame", "per_frame"])
def launch_mode(request, monkeypatch):
    if request.param == "per_frame":
        monkeypatch.setenv("RD_BATCH_PERSISTENT", "0")
    return request.param


def test_batch_ragged_small(gpu_lib, refc, launch_mode):
    ra = gpu_lib
    for fmt in _fmts(ra):
        _batch(ra, refc, 9, 130, 4, fmt)
        _batch(ra, refc, 34, 202, 3, fmt, bands=3)
        _batch(ra, refc, 5, 386, 5, fmt, hist=False)


def test_batch_ragged_ticket_scheduling(gpu_lib, refc, launch_mode):
    """More tiles than resident waves at a ragged width (3000 x 2000: 24 x 1001 tiles per frame, the last of every row
    pair pulled back by 28 quads): the ticket-dealt tiles, the row bands and the multi-frame front all see the same rows."""
    ra = gpu_lib
    _batch(ra, refc, 2000, 3000, 2, ra.FMT_RGBA_F32)
    _batch(ra, refc, 2000, 3000, 2, ra.FMT_RGB_U8, bands=3)
    _batch(ra, refc, 1001, 3002, 2, ra.FMT_RGBA_U8)
    _batch(ra, refc, 1001, 3002, 2, ra.FMT_RGBA_F16, bands=2)


@pytest.mark.parametrize("h,w", [(4000, 6000), (5504, 8256), (402, 6002)])
def test_full_size_ragged_frames(gpu_lib, refc, h, w):
    """Whole camera-sized frames: every byte of every surface and the exact histogram (single launch with histogram, the
    8-band host render without), then the same frames through one multi-frame launch of the batch path."""
    ra = gpu_lib
    rng = np.random.default_rng([0x52415745, h, w])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(7, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    exp32 = refc.render_f32(cfa, u, nthreads=16)
    exp_hist = refc.histogram(refc.pack_u8(exp32))
    for fmt in _fmts(ra):
        exp = _surface(refc, ra, exp32, fmt)
        got, hist = pipe.render(fmt=fmt, with_histogram=True)
        assert np.array_equal(hist, exp_hist), (fmt, "histogram")
        assert np.array_equal(_view(ra, got, fmt), exp), (fmt, "one launch")
        got = pipe.render(fmt=fmt)                                 # >= 16 MiB: row bands + chunked read-back
        assert np.array_equal(_view(ra, got, fmt), exp), (fmt, "row bands")
        del got, exp
    assert np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), refc.pack_u8(exp32))
    pipe.close()
    del exp32
    if h * w <= 24_000_000:
        for fmt in _fmts(ra):
            _batch(ra, refc, h, w, 2, fmt)


def test_export_ring_rgb8_ragged(gpu_lib, refc):
    """main.rs:1777-1786 strips alpha for any width: the RGB8 ring at 6000 x 4000 (HBM- and host-fed) and at a width whose
    rows start on odd 16-bit boundaries."""
    ra = gpu_lib
    for h, w in ((4000, 6000), (64, 130), (33, 202)):
        rng = np.random.default_rng([0x52415745, 3, h, w])
        cfa = random_cfa(rng, h, w)
        p = random_params(rng)
        d = DevBuf.from_array(cfa)
        u = refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST)
        exp = refc.pack_u8(refc.render_f32(cfa, u, nthreads=16))[..., :3]
        ex = ra.Exporter(0, w, h, ra.FMT_RGB_U8, n_slots=2)
        fr = ex.frame(d.ptr, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)
        s = ex.submit(fr)
        assert np.array_equal(ex.wait(s), exp), (h, w, "device-fed")
        ex.release(s)
        s = ex.submit_host(cfa, fr)
        assert np.array_equal(ex.wait(s), exp), (h, w, "host-fed")
        ex.release(s)
        ex.close()
        d.free()
    with pytest.raises(ra.RawdevError):
        ra.Exporter(0, 126, 64, ra.FMT_RGB_U8)                  # narrower than one tile: the pipeline's map kernel serves it
    with pytest.raises(ra.RawdevError):
        ra.BatchExporter(0, 126, 64, ra.FMT_RGB_U8, False)


def test_overlap_instance_on_an_aligned_width(gpu_lib, refc, monkeypatch):
    """RD_TILES=overlap runs the pulled-back-last-tile instance where nothing needs pulling back (W % 128 == 0): the
    overlap is zero quads, every lane counts."""
    import subprocess
    import sys
    code = r"""
import numpy as np
import raweditor_amd as ra
from oracle import ref_c
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params
rng = np.random.default_rng(11)
h, w = 70, 384
cfa = random_cfa(rng, h, w)
p = random_params(rng)
pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)
e = ref_c.render_f32(cfa, ref_c.make_uniforms(p, WB_DAYLIGHT, CM_TEST))
for fmt in (ra.FMT_RGBA_F32, ra.FMT_RGBA_U8, ra.FMT_RGB_U8, ra.FMT_RGBA_F16):
    got, hist = pipe.render(fmt=fmt, with_histogram=True)
    assert np.array_equal(hist, ref_c.histogram(ref_c.pack_u8(e)))
    if fmt == ra.FMT_RGBA_F32:
        assert np.array_equal(got.view(np.uint32), e.view(np.uint32))
    elif fmt == ra.FMT_RGBA_F16:
        assert np.array_equal(got.view(np.uint16), ref_c.pack_f16(e).view(np.uint16))
    else:
        assert np.array_equal(got, ref_c.pack_u8(e)[..., :got.shape[-1]])
print("ok")
"""
    import os
    env = dict(os.environ, RD_TILES="overlap")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


# ------------------------------------------------------------------------------------------------
# round 6: ODD widths.  The reference renders any texture size (shaders.rs:181-187, loader.rs:57-58); a cropped plane can be
# an odd number of pixels wide.  The export kernel takes the W // 2 whole quads of every row pair (rows then start on odd
# 16-bit boundaries: 2-byte aligned dword loads, byte-aligned RGB8 dword stores) and rd_develop_lastcol the last column; no
# entry point refuses a width for its parity any more.
# ------------------------------------------------------------------------------------------------
SMALL_ODD = [(7, 131), (1, 129), (2, 131), (5, 191), (4, 257), (3, 385), (6, 1), (5, 3), (4, 7), (9, 127), (2, 65), (8, 193), (9, 251), (6, 375), (5, 253)]


@pytest.mark.parametrize("math", [0, 1])
def test_small_odd_width_frames_every_surface(gpu_lib, refc, math):
    ra = gpu_lib
    rng = np.random.default_rng([0x52415745, 6, math])
    for h, w in SMALL_ODD:
        cfa = random_cfa(rng, h, w, 65536)
        for cm, params in ((CM_TEST, random_params(rng)), (CM_IDENTITY, {"exposure": 0.5, "contrast": 4.0})):
            pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, cm)
            pipe.set_math_mode(math)
            u = refc.make_uniforms(params, WB_DAYLIGHT, cm, math_mode=math)
            exp32 = refc.render_f32(cfa, u)
            exp_hist = refc.histogram(refc.pack_u8(exp32))
            for fmt in _fmts(ra):
                got, hist = pipe.render(fmt=fmt, with_histogram=True)
                assert np.array_equal(_view(ra, got, fmt), _surface(refc, ra, exp32, fmt)), (h, w, fmt, "surface")
                assert np.array_equal(hist, exp_hist), (h, w, fmt, "histogram")
                got = pipe.render(fmt=fmt)
                assert np.array_equal(_view(ra, got, fmt), _surface(refc, ra, exp32, fmt)), (h, w, fmt, "no histogram")
            pipe.close()


def test_batch_odd_widths(gpu_lib, refc, launch_mode):
    """rd_batch_create takes odd widths (it answered RD_ERR_UNSUPPORTED through round 5): 131 x 7, one
    whole tile + the column (129), the overlapped last tile (203, 387), narrower than a tile (5, 1: the masked instance, or no
    quad at all), row bands, with and without the histogram."""
    ra = gpu_lib
    for fmt in _fmts(ra):
        _batch(ra, refc, 7, 131, 4, fmt)
        _batch(ra, refc, 9, 129, 3, fmt, bands=2)
        _batch(ra, refc, 34, 203, 3, fmt, bands=3)
        _batch(ra, refc, 5, 387, 5, fmt, hist=False)
        if fmt != ra.FMT_RGB_U8:                                  # (RGB8 narrower than one tile stays with the pipeline's map kernel)
            _batch(ra, refc, 7, 5, 3, fmt)
            _batch(ra, refc, 6, 1, 2, fmt)
    _batch(ra, refc, 1001, 3001, 2, ra.FMT_RGBA_F32)            # more tiles than resident waves
    _batch(ra, refc, 9, 251, 4, ra.FMT_RGBA_F32, bands=2)       # the shifted-window tiling's single-quad last tile, odd and even
    _batch(ra, refc, 8, 374, 3, ra.FMT_RGBA_F32)
    _batch(ra, refc, 1000, 3002, 2, ra.FMT_RGBA_F32, bands=3)   # W % 4 == 2
    _batch(ra, refc, 1001, 3001, 2, ra.FMT_RGB_U8, bands=3)


def test_full_size_odd_frame(gpu_lib, refc):
    """6001 x 4001: every byte of all four surfaces and the exact histogram -- one launch with the
    histogram, the 8-band host render without, render_full_res_to_bytes, and a multi-frame launch of the batch path."""
    ra = gpu_lib
    h, w = 4001, 6001
    rng = np.random.default_rng([0x52415745, h, w])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(7, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    exp32 = refc.render_f32(cfa, u, nthreads=16)
    exp_hist = refc.histogram(refc.pack_u8(exp32))
    for fmt in _fmts(ra):
        exp = _surface(refc, ra, exp32, fmt)
        got, hist = pipe.render(fmt=fmt, with_histogram=True)
        assert np.array_equal(hist, exp_hist), (fmt, "histogram")
        assert np.array_equal(_view(ra, got, fmt), exp), (fmt, "one launch")
        got = pipe.render(fmt=fmt)                                 # >= 16 MiB: row bands + chunked read-back
        assert np.array_equal(_view(ra, got, fmt), exp), (fmt, "row bands")
        del got, exp
    assert np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), refc.pack_u8(exp32))
    pipe.close()
    del exp32
    for fmt in _fmts(ra):
        _batch(ra, refc, h, w, 2, fmt)


def test_export_ring_odd_widths(gpu_lib, refc):
    ra = gpu_lib
    for h, w, fmt in ((4001, 6001, ra.FMT_RGB_U8), (7, 131, ra.FMT_RGB_U8), (7, 131, ra.FMT_RGBA_U8), (33, 203, ra.FMT_RGBA_U8), (5, 3, ra.FMT_RGBA_U8)):
        rng = np.random.default_rng([0x52415745, 4, h, w])
        cfa = random_cfa(rng, h, w)
        p = random_params(rng)
        d = DevBuf.from_array(cfa)
        u = refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST)
        exp = refc.pack_u8(refc.render_f32(cfa, u, nthreads=16))[..., :3 if fmt == ra.FMT_RGB_U8 else 4]
        ex = ra.Exporter(0, w, h, fmt, n_slots=2)
        fr = ex.frame(d.ptr, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)
        s = ex.submit(fr)
        assert np.array_equal(ex.wait(s), exp), (h, w, fmt, "device-fed")
        ex.release(s)
        s = ex.submit_host(cfa, fr)
        assert np.array_equal(ex.wait(s), exp), (h, w, fmt, "host-fed")
        ex.release(s)
        ex.close()
        d.free()
