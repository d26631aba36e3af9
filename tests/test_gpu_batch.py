"""-m gpu: the batch export entry points (rd_batch_*): device-resident frames, u64 histogram accumulation across
frames.  Two ways through rd_batch_develop are covered: the default multi-frame launch (one launch covers as many
frames of the call as the limits allow; tickets, uniforms and surfaces change frame inside the kernel) and, with
RD_BATCH_PERSISTENT=0, one fused launch per frame or per row band."""
import numpy as np
import pytest

from tests.gpu_util import DevBuf, sync
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu


def _run_batch(ra, refc, h, w, n, fmt, bands, with_hist=True, math=0):
    rng = np.random.default_rng([0x52415745, h, w, n])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    bpp = ra.BYTES_PER_PIXEL[fmt]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * bpp) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, fmt, with_hist, math_mode=math)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    exp_hist = np.zeros(768, np.uint64)
    exps = []
    for c, p in zip(cfas, params):
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST, math_mode=math)
        e = refc.render_f32(c, u)
        exps.append(e)
        exp_hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
    for rep in range(2):                       # second pass: the accumulator was reset by histogram()
        be.develop(frames, row_bands=bands)
        if with_hist:
            be.histogram(d_hist.ptr)
        sync()
        for e, o in zip(exps, d_out):
            if fmt == ra.FMT_RGBA_F32:
                got = o.to_array(np.float32, (h, w, 4))
                assert np.array_equal(got.view(np.uint32), e.view(np.uint32))
            elif fmt == ra.FMT_RGBA_F16:
                got = o.to_array(np.uint16, (h, w, 4))
                assert np.array_equal(got, refc.pack_f16(e).view(np.uint16))
            else:
                got = o.to_array(np.uint8, (h, w, 4))
                assert np.array_equal(got, refc.pack_u8(e))
        if with_hist:
            got_hist = d_hist.to_array(np.uint64, (768,))
            assert np.array_equal(got_hist, exp_hist)
    # accumulate over two develop() calls before one histogram(): counts add up in u64
    if with_hist:
        be.develop(frames, row_bands=bands)
        be.develop(frames, row_bands=1)
        be.histogram(d_hist.ptr)
        sync()
        assert np.array_equal(d_hist.to_array(np.uint64, (768,)), 2 * exp_hist)
    be.close()


@pytest.fixture(params=["multi_frame", "per_frame"])
def launch_mode(request, monkeypatch):
    if request.param == "per_frame":
        monkeypatch.setenv("RD_BATCH_PERSISTENT", "0")
    return request.param


@pytest.mark.parametrize("fmt_name", ["F32", "F16", "U8"])
@pytest.mark.parametrize("bands", [1, 3])
def test_batch_small_frames(gpu_lib, refc, fmt_name, bands, launch_mode):
    ra = gpu_lib
    fmt = {"F32": ra.FMT_RGBA_F32, "F16": ra.FMT_RGBA_F16, "U8": ra.FMT_RGBA_U8}[fmt_name]
    _run_batch(ra, refc, 34, 48, 5, fmt, bands)


def test_batch_odd_height_many_bands_no_hist(gpu_lib, refc, launch_mode):
    ra = gpu_lib
    _run_batch(ra, refc, 33, 64, 3, ra.FMT_RGBA_F32, 8, with_hist=False)
    _run_batch(ra, refc, 2, 2, 2, ra.FMT_RGBA_F32, 4)          # more bands than units


def test_batch_mid_size_frames(gpu_lib, refc, launch_mode):
    """A frame big enough to occupy every workgroup of the fixed grid (1504 x 1000)."""
    ra = gpu_lib
    _run_batch(ra, refc, 1000, 1504, 2, ra.FMT_RGBA_F32, 1)
    _run_batch(ra, refc, 1000, 1504, 2, ra.FMT_RGBA_F16, 4)


def test_batch_ticket_scheduling(gpu_lib, refc, monkeypatch, launch_mode):
    """More tiles than resident waves (3008 x 2008: 24 120 tiles against 8192 waves), so most tiles are dealt by the
    ticket counters; repeated launches (the counters must come back to zero), row bands and the optional two-stream
    issue all give the oracle's bits and the same histogram."""
    ra = gpu_lib
    _run_batch(ra, refc, 2008, 3008, 2, ra.FMT_RGBA_F32, 1)
    _run_batch(ra, refc, 2008, 3008, 2, ra.FMT_RGBA_U8, 3)
    monkeypatch.setenv("RD_BATCH_STREAMS", "2")
    _run_batch(ra, refc, 2008, 3008, 3, ra.FMT_RGBA_F16, 2)
    _run_batch(ra, refc, 34, 48, 5, ra.FMT_RGBA_F32, 3)
    monkeypatch.delenv("RD_BATCH_STREAMS")


def test_batch_contracted_math(gpu_lib, refc, launch_mode):
    ra = gpu_lib
    _run_batch(ra, refc, 34, 256, 3, ra.FMT_RGBA_F32, 2, math=ra.MATH_CONTRACTED)
    _run_batch(ra, refc, 34, 48, 3, ra.FMT_RGBA_U8, 1, math=ra.MATH_CONTRACTED)


def test_multi_frame_launch_limits(gpu_lib, refc, monkeypatch):
    """The multi-frame path cuts a call into launches: at most RD_BATCH_MAX_FRAMES frames each (diagnostic cap), and
    never two frames whose surfaces overlap in one launch.  7 frames with caps 1, 2, 3 and 7, then 7 frames written to
    a ring of 3 surfaces (every launch ends where the ring would wrap): same bits, same histogram."""
    ra = gpu_lib
    for cap in ("1", "2", "3", "7"):
        monkeypatch.setenv("RD_BATCH_MAX_FRAMES", cap)
        _run_batch(ra, refc, 260, 384, 7, ra.FMT_RGBA_F32, 1)
    monkeypatch.delenv("RD_BATCH_MAX_FRAMES")
    # ring of 3: what survives in slot k is the last frame sent there
    h, w, n, ring = 260, 384, 7, 3
    rng = np.random.default_rng([0x52415745, 77])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 4) for _ in range(ring)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_U8, True)
    frames = be.make_frames([b.ptr for b in d_in], [d_out[i % ring].ptr for i in range(n)], params, WB_DAYLIGHT, CM_TEST)
    be.develop(frames)
    be.histogram(d_hist.ptr)
    sync()
    exp_hist = np.zeros(768, np.uint64)
    exp8 = []
    for c, p in zip(cfas, params):
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
        exp8.append(refc.pack_u8(refc.render_f32(c, u)))
        exp_hist += refc.histogram(exp8[-1]).reshape(-1).astype(np.uint64)
    for slot in range(ring):
        last = max(i for i in range(n) if i % ring == slot)
        assert np.array_equal(d_out[slot].to_array(np.uint8, (h, w, 4)), exp8[last]), slot
    assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist)
    be.close()


def test_batch_rejects_bad_arguments(gpu_lib):
    ra = gpu_lib
    ra.BatchExporter(0, 7, 8, ra.FMT_RGBA_F32).close()           # odd width: taken since round 6 (tests/test_gpu_ragged.py)
    with pytest.raises(ra.RawdevError):
        ra.BatchExporter(0, 0, 8, ra.FMT_RGBA_F32)               # empty frame
    with pytest.raises(ra.RawdevError):
        ra.BatchExporter(0, 8, 8, 5)                             # unknown format
    be = ra.BatchExporter(0, 8, 8, ra.FMT_RGBA_F32, with_histogram=False)
    with pytest.raises(ra.RawdevError):
        be.histogram(1234)                                       # created without a histogram
    buf = DevBuf(8 * 8 * 16 + 16)
    frames = be.make_frames([buf.ptr], [buf.ptr + 4], [ra.EditParams()], WB_DAYLIGHT, CM_TEST)
    with pytest.raises(ra.RawdevError):
        be.develop(frames)                                       # misaligned f32 surface


def test_multi_frame_launch_with_heterogeneous_frames(gpu_lib, refc):
    """One multi-frame launch whose frames differ in everything a descriptor carries: slider stacks that elide different
    steps (defaults + identity matrix, a few sliders, all ten + camera matrix), white balance, colour matrix and black
    level -- a wave switches uniforms (and elision branches) when its tiles cross into the next frame.  Burst-eligible
    size (1024 x 1030, f32), then the same call with one CFA plane at a 4-byte-aligned-only address (the whole call then
    runs the instantiation without the read burst)."""
    import ctypes as C
    from raweditor_amd._lib import RdFrame
    from tests.helpers import CM_IDENTITY
    ra = gpu_lib
    h, w = 1030, 1024
    rng = np.random.default_rng([0x52415745, 606])
    stacks = [dict(), dict(exposure=1.0, contrast=5.0), random_params(rng), dict(saturation=40.0, vibrance=-0.5, blacks=0.1),
              random_params(rng), dict(whites=0.9, temperature=0.3, tint=-0.2, highlights=0.4, shadows=-0.3)]
    cms = [CM_IDENTITY, CM_IDENTITY, CM_TEST, CM_IDENTITY, CM_TEST, CM_TEST]
    wbs = [(1, 1, 1, 1), WB_DAYLIGHT, WB_DAYLIGHT, (1.7, 1.0, 1.9, 1.0), (2.2, 1.0, 1.3, 1.0), WB_DAYLIGHT]
    bls = [0, 0, 64, 0, 0, 256]
    n = len(stacks)
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    exp, exp_hist = [], np.zeros(768, np.uint64)
    for c, p, cm, wb, bl in zip(cfas, stacks, cms, wbs, bls):
        u = refc.make_uniforms(p, wb, cm, black_level=bl)
        e = refc.render_f32(c, u, nthreads=8)
        exp.append(e)
        exp_hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
    for misalign in (0, 4):
        d_in = []
        for i, c in enumerate(cfas):
            if i == 3 and misalign:                             # this plane starts 4 bytes into its buffer
                buf = DevBuf(c.nbytes + 16)
                from raweditor_amd._lib import check
                from raweditor_amd import _lib
                check(_lib.lib().rd_memcpy_h2d(0, C.c_void_p(buf.ptr + misalign), c.ctypes.data_as(C.c_void_p), c.nbytes))
                d_in.append((buf, buf.ptr + misalign))
            else:
                buf = DevBuf.from_array(c)
                d_in.append((buf, buf.ptr))
        d_out = [DevBuf(h * w * 16) for _ in range(n)]
        d_hist = DevBuf(768 * 8)
        be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, True)
        frames = (RdFrame * n)()
        for i in range(n):
            frames[i].cfa_dev = d_in[i][1]
            frames[i].out_dev = d_out[i].ptr
            frames[i].params = ra.EditParams(**stacks[i]).to_c()
            frames[i].wb_multipliers[:] = [float(x) for x in wbs[i]]
            frames[i].color_matrix[:] = [float(x) for x in cms[i]]
            frames[i].black_level = bls[i]
        for rep in range(2):
            be.develop(frames)
            assert be.last_launch_count() == 1                  # six frames, one launch
            be.histogram(d_hist.ptr)
            sync()
            for i in range(n):
                got = d_out[i].to_array(np.float32, (h, w, 4))
                assert np.array_equal(got.view(np.uint32), exp[i].view(np.uint32)), (misalign, rep, i)
            assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist), (misalign, rep)
        be.close()


def test_same_batch_on_two_streams_in_turn(gpu_lib, refc):
    """The descriptor array of a call is kept and reused when the next call brings the same frames -- also when that call
    comes on another stream: the reuse waits for the copy that filled the array."""
    import ctypes as C
    from raweditor_amd import _lib
    from raweditor_amd._lib import check
    ra = gpu_lib
    h, w, n = 260, 384, 6
    rng = np.random.default_rng([0x52415745, 2020])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 16) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, True)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    exp_hist = np.zeros(768, np.uint64)
    exps = []
    for c, p in zip(cfas, params):
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
        exps.append(refc.render_f32(c, u))
        exp_hist += refc.histogram(refc.pack_u8(exps[-1])).reshape(-1).astype(np.uint64)
    streams = []
    for _ in range(2):
        s = C.c_void_p()
        check(_lib.lib().rd_stream_create(0, C.byref(s)))
        streams.append(s.value)
    for rep in range(4):
        s = streams[rep % 2]
        be.develop(frames, stream=s)
        be.histogram(d_hist.ptr, stream=s)
        check(_lib.lib().rd_stream_synchronize(0, C.c_void_p(s)))
        assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist), rep
        for e, o in zip(exps, d_out):
            assert np.array_equal(o.to_array(np.float32, (h, w, 4)).view(np.uint32), e.view(np.uint32)), rep
    for s in streams:
        check(_lib.lib().rd_stream_destroy(0, C.c_void_p(s)))
    be.close()


def test_three_frame_arrays_rotating_over_two_unsynchronised_streams(gpu_lib, refc):
    """Calls alternate between two streams WITHOUT a synchronise in between while three different frame arrays rotate: an
    array is reused from the other stream (the call queues behind the array's previous reader) and rewritten two calls later
    (after that reader has finished).  Every array has its own surfaces; all of them must hold the oracle's bits at the end."""
    import ctypes as C
    from raweditor_amd import _lib
    from raweditor_amd._lib import check
    ra = gpu_lib
    h, w, n = 130, 640, 5
    rng = np.random.default_rng([0x52415745, 2026])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, False)
    arrays, outs, exps = [], [], []
    for a in range(3):
        params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
        d_out = [DevBuf(h * w * 16) for _ in range(n)]
        arrays.append(be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST))
        outs.append(d_out)
        exps.append([refc.render_f32(c, refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST))
                     for c, p in zip(cfas, params)])
    streams = []
    for _ in range(2):
        s = C.c_void_p()
        check(_lib.lib().rd_stream_create(0, C.byref(s)))
        streams.append(s.value)
    order = [0, 0, 1, 2, 0, 1, 1, 2, 2, 0, 1, 2, 0, 0, 2, 1] * 3             # reuse from the other stream, rewrites, repeats
    for k, a in enumerate(order):
        be.develop(arrays[a], stream=streams[k % 2])
    for s in streams:
        check(_lib.lib().rd_stream_synchronize(0, C.c_void_p(s)))
    for a in range(3):
        for e, o in zip(exps[a], outs[a]):
            assert np.array_equal(o.to_array(np.float32, (h, w, 4)).view(np.uint32), e.view(np.uint32)), a
    for s in streams:
        check(_lib.lib().rd_stream_destroy(0, C.c_void_p(s)))
    be.close()
    for b in d_in + [o for d in outs for o in d]:
        b.free()


# ------------------------------------------------------------------------------------------------
# round 6: the measurement aids behind bench.py's self-diagnosis (rd_batch_set_launch_timing / _launch_timeline,
# rd_batch_probe_pattern, rd_batch_measure_clock)
# ------------------------------------------------------------------------------------------------
def _taps_np(cfa):
    """tests.helpers.expected_taps (the SURVEY 8 a2 table), vectorised: (h, w, 3) float32 of raw / 4096."""
    h, w = cfa.shape
    yy, xx = np.mgrid[0:h, 0:w]

    def at(y, x):
        return cfa[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)].astype(np.float32) / np.float32(4096.0)
    odd, ecol = (yy % 2 == 1), (xx % 2 == 0)
    r = np.where(odd & ecol, at(yy + 1, xx), np.where(odd, at(yy + 1, xx - 1), np.where(ecol, at(yy, xx), at(yy, xx - 1))))
    g = np.where(odd & ecol, at(yy, xx), np.where(odd, at(yy, xx - 1), np.where(ecol, at(yy, xx + 1), at(yy, xx))))
    b = np.where(odd & ecol, at(yy, xx + 1), np.where(odd, at(yy, xx), at(yy - 1, xx)))
    return np.stack([r, g, b], axis=-1).astype(np.float32)


def test_launch_timeline_keeps_the_last_calls(gpu_lib, refc):
    ra = gpu_lib
    from tests.helpers import expected_taps
    small = random_cfa(np.random.default_rng(3), 5, 6)
    assert np.array_equal(_taps_np(small), (expected_taps(small) / 4096.0).astype(np.float32))     # the helper above, against the table
    h, w, n = 66, 256, 20                                       # 20 frames, at most 8 per launch: 3 launches per call
    rng = np.random.default_rng([6, h, w])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 16) for _ in range(n)]
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, True)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    assert be.launch_timeline() == []                            # off by default
    be.develop(frames)
    sync()
    assert be.launch_timeline() == []
    be.set_launch_timing(2)
    for _ in range(3):
        be.develop(frames)
    sync()
    tl = be.launch_timeline()
    per_call = be.last_launch_count()
    assert per_call == 3 and len(tl) == 2 * per_call             # the window holds the last two of the three calls
    assert [c for c, _, _ in tl] == [0] * per_call + [1] * per_call
    assert tl[0][1] == 0.0
    for i, (_, st, en) in enumerate(tl):
        assert en > st >= 0.0, tl
        if i:
            assert st >= tl[i - 1][2] - 1e-3, tl                 # one stream: a launch starts after its predecessor ended
    u = refc.make_uniforms({f: getattr(params[7], f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
    assert np.array_equal(d_out[7].to_array(np.float32, (h, w, 4)).view(np.uint32), refc.render_f32(cfas[7], u).view(np.uint32))   # timed launches are ordinary launches
    be.set_launch_timing(0)
    assert be.launch_timeline() == []
    be.develop(frames)
    sync()
    assert be.launch_timeline() == []
    with pytest.raises(ra.RawdevError):
        be.set_launch_timing(65)
    be.close()


def test_pattern_probe_moves_the_kernels_bytes_without_its_arithmetic(gpu_lib, refc):
    """rd_batch_probe_pattern: the f32 export kernel's loads, sweeps, tickets, LDS stage and stores with the colour stack, the
    gamma and the histogram removed -- every pixel receives its three demosaic taps as raw / 4096 and alpha 1; the histogram
    accumulator is not touched; a develop on the same context afterwards is an ordinary develop."""
    ra = gpu_lib
    h, w, n = 1026, 1024, 3                                     # (h / 2 + 1) * w >= 2^19: the read-burst instance takes it
    rng = np.random.default_rng([66, h, w])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 16) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, True)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    be.set_launch_timing(1)
    be.probe_pattern(frames)
    sync()
    assert len(be.launch_timeline()) == be.last_launch_count() == 1
    be.set_launch_timing(0)
    for c, o in zip(cfas, d_out):
        got = o.to_array(np.float32, (h, w, 4))
        assert np.array_equal(got[..., :3], _taps_np(c)) and np.all(got[..., 3] == 1.0)
    be.develop(frames)
    be.histogram(d_hist.ptr)
    sync()
    exp_hist = np.zeros(768, np.uint64)
    for c, p, o in zip(cfas, params, d_out):
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
        e = refc.render_f32(c, u)
        assert np.array_equal(o.to_array(np.float32, (h, w, 4)).view(np.uint32), e.view(np.uint32))
        exp_hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
    assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist)      # the probe counted nothing
    be.close()
    # what the probe is not built for is refused, not approximated
    be8 = ra.BatchExporter(0, w, h, ra.FMT_RGBA_U8, True)
    with pytest.raises(ra.RawdevError) as ei:
        be8.probe_pattern(frames)
    assert ei.value.code == -5
    be8.close()
    small = ra.BatchExporter(0, 64, 34, ra.FMT_RGBA_F32, True)
    s_in, s_out = DevBuf.from_array(random_cfa(rng, 34, 64)), DevBuf(34 * 64 * 16)
    with pytest.raises(ra.RawdevError) as ei:
        small.probe_pattern(small.make_frames([s_in.ptr], [s_out.ptr], params[:1], WB_DAYLIGHT, CM_TEST))
    assert ei.value.code == -5
    small.close()


def test_measure_clock_is_an_ordinary_develop_that_also_reads_the_clock(gpu_lib, refc):
    ra = gpu_lib
    h, w, n = 1026, 1024, 9                                     # two launches (8 + 1 frames): the LAST one's stamps are reported
    rng = np.random.default_rng([67, h, w])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 16) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, True)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    ghz, lo, hi, busy_us = be.measure_clock(frames)
    assert 0.3 < lo <= ghz <= hi < 3.5 and busy_us > 0.0, (ghz, lo, hi, busy_us)
    be.histogram(d_hist.ptr)
    sync()
    exp_hist = np.zeros(768, np.uint64)
    for c, p, o in zip(cfas, params, d_out):
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
        e = refc.render_f32(c, u)
        assert np.array_equal(o.to_array(np.float32, (h, w, 4)).view(np.uint32), e.view(np.uint32))
        exp_hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
    assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist)
    be.close()
    nohist = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, False)
    with pytest.raises(ra.RawdevError) as ei:
        nohist.measure_clock(frames)
    assert ei.value.code == -5
    nohist.close()
