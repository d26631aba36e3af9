"""-m gpu: the export feed (rd_exporter_*, RD_FMT_RGB_U8) -- SURVEY.md section 8f rank 1.  Bytes must equal the
oracle's RGBA8 pack (pipeline.rs:322), with alpha dropped for RGB8 exactly like main.rs:1777-1786."""
import numpy as np
import pytest

from tests.gpu_util import DevBuf
from tests.helpers import CM_IDENTITY, CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu


def _expected(refc, cfa, p, rgb):
    u = refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST)
    u8 = refc.pack_u8(refc.render_f32(cfa, u))
    return u8[..., :3] if rgb else u8


@pytest.mark.parametrize("fmt_name", ["RGBA8", "RGB8"])
def test_exporter_ring(gpu_lib, refc, fmt_name):
    ra = gpu_lib
    fmt = ra.FMT_RGB_U8 if fmt_name == "RGB8" else ra.FMT_RGBA_U8
    h, w, n = 70, 256, 7
    rng = np.random.default_rng([5, fmt])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    ps = [random_params(rng) for _ in range(n)]
    dev = [DevBuf.from_array(c) for c in cfas]
    ex = ra.Exporter(0, w, h, fmt, n_slots=3)   # (three slots here on purpose: the default is two)
    frames = [ex.frame(d.ptr, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST) for d, p in zip(dev, ps)]
    seen = []
    for i, surf in ex.export(frames):
        assert surf.shape == (h, w, 3 if fmt_name == "RGB8" else 4)
        assert np.array_equal(surf, _expected(refc, cfas[i], ps[i], fmt_name == "RGB8")), i
        seen.append(i)
    assert seen == list(range(n))
    # ring discipline: a slot must be released before it is reused
    slots = [ex.submit(frames[k]) for k in range(3)]
    with pytest.raises(ra.RawdevError):
        ex.submit(frames[3])
    ex.wait(slots[0])
    ex.release(slots[0])
    s = ex.submit(frames[3])
    assert s == slots[0]
    assert np.array_equal(ex.wait(s), _expected(refc, cfas[3], ps[3], fmt_name == "RGB8"))
    ex.close()


@pytest.mark.parametrize("h,w,fmt_name", [(70, 256, "RGBA8"), (41, 130, "RGBA8"), (2056, 4224, "RGB8")])
def test_exporter_from_host_planes(gpu_lib, refc, h, w, fmt_name):
    """rd_exporter_submit_host: the CFA plane arrives in host memory (RawDataResult.data, raw/loader.rs:11-19) -- pageable
    (staged, reusable at once: the source is scribbled over right after the call), page-locked (read in place), mixed with
    device-resident frames in one ring; 17 MB planes cross the 8 MiB staging pieces."""
    ra = gpu_lib
    fmt = ra.FMT_RGB_U8 if fmt_name == "RGB8" else ra.FMT_RGBA_U8
    n = 7 if h < 1000 else 5
    rng = np.random.default_rng([6, h, w])
    cfas = [random_cfa(rng, h, w, 65536 if k == 2 else 4096) for k in range(n)]
    ps = [random_params(rng) for _ in range(n)]
    exp = [refc.pack_u8(refc.render_f32(c, refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST), nthreads=8)) for c, p in zip(cfas, ps)]
    exp = [e[..., :3] if fmt_name == "RGB8" else e for e in exp]
    ex = ra.Exporter(0, w, h, fmt, n_slots=2)
    pins = [ra.PinnedBytes(h * w * 2) for _ in range(2)]
    dev = DevBuf.from_array(cfas[3])
    pending, seen = [], []

    def drain_one():
        j, s = pending.pop(0)
        assert np.array_equal(ex.wait(s), exp[j]), (j, "host" if j != 3 else "device")
        ex.release(s)
        seen.append(j)

    for k in range(n):
        if len(pending) == 2:
            drain_one()
        fr = ex.frame(0, ra.EditParams(**ps[k]), WB_DAYLIGHT, CM_TEST)
        if k == 3:                                           # a device-resident frame between host-fed ones
            pending.append((k, ex.submit(ex.frame(dev.ptr, ra.EditParams(**ps[k]), WB_DAYLIGHT, CM_TEST))))
        elif k % 2:                                          # page-locked source: DMA reads it in place
            pin = pins[(k // 2) % 2]
            pin.array.view(np.uint16)[:] = cfas[k].reshape(-1)
            assert ra._lib.lib().rd_debug_is_pinned_host(pin.ptr, pin.nbytes) == 1
            pending.append((k, ex.submit_host(pin.array.view(np.uint16), fr)))
        else:                                                # pageable source: free for reuse when the call returns
            src = cfas[k].copy()
            pending.append((k, ex.submit_host(src, fr)))
            src[:] = 0xdead
    while pending:
        drain_one()
    assert seen == list(range(n))
    got = [surf.copy() for _, surf in ex.export_host((c, ex.frame(0, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)) for c, p in zip(cfas, ps))]
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    with pytest.raises(ra.RawdevError):
        ex.submit_host(cfas[0][:-1], ex.frame(0, ra.EditParams(), WB_DAYLIGHT, CM_TEST))       # wrong size
    rc = ra._lib.lib().rd_exporter_submit_host(ex._h, None, None, None)
    assert rc != 0
    for pin in pins:
        pin.free()
    ex.close()


def test_rgb8_surface_through_the_pipeline(gpu_lib, refc):
    """RD_FMT_RGB_U8 via rd_render: export kernel (even W >= 128), map kernel (other widths, preview), edges."""
    ra = gpu_lib
    rng = np.random.default_rng(8)
    for h, w in ((2, 128), (3, 128), (66, 384), (9, 13), (16, 24), (64, 130)):
        cfa = random_cfa(rng, h, w)
        p = random_params(rng)
        pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)
        got, hist = pipe.render(fmt=ra.FMT_RGB_U8, with_histogram=True)
        exp = _expected(refc, cfa, p, True)
        assert got.shape == (h, w, 3) and np.array_equal(got, exp), (h, w)
        assert np.array_equal(hist, refc.histogram(_expected(refc, cfa, p, False)))
        small = pipe.render(11, 5, ra.FMT_RGB_U8)
        u = refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST)
        assert np.array_equal(small, refc.pack_u8(refc.render_f32(cfa, u, 11, 5))[..., :3])
    with pytest.raises(ra.RawdevError):
        ra.Exporter(0, 126, 64, ra.FMT_RGB_U8)          # RGB8 export needs one whole tile (W >= 128)
    with pytest.raises(ra.RawdevError):
        ra.Exporter(0, 128, 64, ra.FMT_RGBA_U8, n_slots=0)


def test_exporter_full_size_rgb8(gpu_lib, refc):
    """6016 x 4016 through the ring; sampled bands against the oracle + alpha-strip consistency with RGBA8."""
    ra = gpu_lib
    h, w = 4016, 6016
    rng = np.random.default_rng(24)
    cfa = random_cfa(rng, h, w)
    p = random_params(rng)
    d = DevBuf.from_array(cfa)
    u = refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST)
    ex3 = ra.Exporter(0, w, h, ra.FMT_RGB_U8, n_slots=2)
    ex4 = ra.Exporter(0, w, h, ra.FMT_RGBA_U8, n_slots=2)
    f3 = ex3.frame(d.ptr, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)
    s3, s4 = ex3.submit(f3), ex4.submit(f3)
    rgb, rgba = ex3.wait(s3), ex4.wait(s4)
    assert np.array_equal(rgb, rgba[..., :3]) and np.all(rgba[..., 3] == 255)
    for r0, r1 in ((0, 4), (2007, 2011), (h - 4, h)):
        assert np.array_equal(rgb[r0:r1], refc.pack_u8(refc.render_band(cfa, u, r0, r1))[..., :3])
    ex3.close(); ex4.close()


def test_catalog_to_export_end_to_end(gpu_lib, refc, tmp_path):
    """The caller's flow (main.rs:973-1006 then :1744-1799) with this build's pieces on both sides of the path:
    DNG / u16 ingest -> RawDataResult -> identity matrix stub -> edits from the app's SQLite table -> export ring."""
    import sqlite3
    from raweditor_amd import catalog, ingest
    from tests.test_ingest_catalog_cpu import _write_dng
    ra = gpu_lib
    rng = np.random.default_rng(31)
    h, w = 40, 128
    conn = sqlite3.connect(tmp_path / "library.db")
    catalog.init_schema(conn)
    cfas, stacks = [], []
    for k in range(4):
        cfa = random_cfa(rng, h, w)
        cfas.append(cfa)
        if k % 2:
            path = tmp_path / f"f{k}.dng"
            _write_dng(path, cfa, "<", strips=2)
        else:
            path = tmp_path / f"f{k}.u16"
            cfa.astype("<u2").tofile(path)
        conn.execute("INSERT INTO images (path, filename, width, height, imported_at) VALUES (?,?,?,?,0)",
                     (str(path), path.name, w, h))
        if k != 3:                                         # image 4 has no edits -> default stack
            p = ra.EditParams(**random_params(rng))
            catalog.save_edit_params(conn, k + 1, p)
            stacks.append(p)
        else:
            stacks.append(ra.EditParams())
    conn.commit()
    ex = ra.Exporter(0, w, h, ra.FMT_RGB_U8, n_slots=2)
    keep, frames, wbs = [], [], []
    for e in catalog.export_manifest(conn):
        raw = (ingest.load_dng_uncompressed(e.path) if e.path.endswith(".dng")
               else ingest.load_raw_u16(e.path, e.width, e.height, wb_coeffs=[2.0, 1.0, 1.5]))
        cm = ra.calculate_cam_to_srgb_matrix(raw.color_matrix)          # identity, like the reference
        d = DevBuf.from_array(np.asarray(raw.data))
        keep.append(d)
        wbs.append(raw.wb_multipliers)
        frames.append(ex.frame(d.ptr, e.params, raw.wb_multipliers, cm))
    for i, surf in ex.export(frames):
        u = refc.make_uniforms({f: getattr(stacks[i], f) for f in ra.FIELDS}, wbs[i], CM_IDENTITY)
        assert np.array_equal(surf, refc.pack_u8(refc.render_f32(cfas[i], u))[..., :3]), i
    ex.close()
