"""-m gpu: render_full_res_to_bytes as its caller invokes it (pipeline.rs:526-606; export_image_async, main.rs:1749-1754).

  * the band-pipelined read-back gives the oracle's bytes for a pageable destination (fresh and caller-provided), a
    page-locked one (rd_host_alloc: the direct-DMA path), surfaces just above the 16 MiB banding threshold, the RGB8 and
    f32 surfaces and a fused histogram (one launch, chunked copies);
  * the call does not hold the pipeline's lock: while one thread loops render_full_res_to_bytes on a 24 MP frame,
    another thread's render_to_bytes (the UI thread's preview, main.rs:1525) keeps a median under 1 ms and stays
    bit-identical to the oracle;
  * the uniforms are snapshotted: an export that races rd_update_uniforms is entirely one stack or entirely the other.
"""
import statistics
import threading
import time

import numpy as np
import pytest

from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu


def _oracle8(refc, cfa, params, tw=None, th=None, zoom=1.0, pan=(0.0, 0.0)):
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST, zoom, pan[0], pan[1])
    return refc.pack_u8(refc.render_f32(cfa, u, tw, th, nthreads=8))


@pytest.mark.parametrize("h,w", [(2056, 2048), (1030, 4224), (2009, 2304)])
def test_banded_readback_sizes_and_destinations(gpu_lib, refc, h, w):
    """Surfaces of 16.1-18.5 MiB: eight row bands of unequal height (odd H: the last unit has one row), chunk boundaries
    inside bands, every kind of destination."""
    ra = gpu_lib
    rng = np.random.default_rng([0x52415745, h, w])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(3, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    exp = _oracle8(refc, cfa, params)
    assert np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), exp), "fresh pageable destination"
    mine = np.full(h * w * 4, 0x5a, np.uint8)
    assert pipe.render_full_res_to_bytes(out=mine) is mine
    assert np.array_equal(mine.reshape(h, w, 4), exp), "caller-provided pageable destination"
    pin = ra.PinnedBytes(h * w * 4)
    pin.array[:] = 0xa5
    pipe.render_full_res_to_bytes(out=pin.array)
    assert np.array_equal(pin.array.reshape(h, w, 4), exp), "page-locked destination (direct DMA)"
    pin.free()
    # the general form: fused histogram (one launch) + chunked copies; the f32 and RGB8 surfaces in bands
    got8, hist = pipe.render(fmt=ra.FMT_RGBA_U8, with_histogram=True)
    assert np.array_equal(got8, exp) and np.array_equal(hist, refc.histogram(exp))
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    f32 = refc.render_f32(cfa, u, nthreads=8)
    got32 = pipe.render(fmt=ra.FMT_RGBA_F32)
    assert np.array_equal(got32.view(np.uint32), f32.view(np.uint32))
    got3 = pipe.render(fmt=ra.FMT_RGB_U8)
    assert np.array_equal(got3, exp[..., :3])
    with pytest.raises(ra.RawdevError):
        pipe.render_full_res_to_bytes(out=np.empty(h * w * 4 - 4, np.uint8))
    pipe.close()


def test_export_does_not_block_the_ui_thread(gpu_lib, refc):
    """main.rs:1749-1754 (export on a blocking thread) beside main.rs:1515-1531 (view() on the UI thread), one
    Arc<RenderPipeline>: 24 MP exports back to back while the other thread renders previews."""
    ra = gpu_lib
    h, w = 4016, 6016
    rng = np.random.default_rng([0x52415745, 77])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(9, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    exp_prev = _oracle8(refc, cfa, params, tw=pipe.preview_width, th=pipe.preview_height)
    exp_full_rows = _oracle8(refc, cfa[:8], params)          # rows 0..5 of the frame only need CFA rows 0..7
    pin = ra.PinnedBytes(h * w * 4)
    stop = threading.Event()
    exports, errors = [], []

    def export_loop():
        try:
            k = 0
            while not stop.is_set():
                t0 = time.perf_counter()
                out = pipe.render_full_res_to_bytes(out=pin.array if k % 2 == 0 else None)   # direct DMA / staged, alternating
                exports.append((time.perf_counter() - t0) * 1e3)
                if not np.array_equal(out.reshape(h, w, 4)[:6], exp_full_rows[:6]):
                    errors.append(f"export {k}: first rows differ from the oracle")
                k += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    # the preview alone (no export running): the baseline of this box
    for _ in range(5):
        pipe.render_to_bytes()
    alone = []
    for _ in range(40):
        t0 = time.perf_counter()
        pipe.render_to_bytes()
        alone.append((time.perf_counter() - t0) * 1e3)
    t = threading.Thread(target=export_loop)
    t.start()
    while len(exports) < 2 and t.is_alive():                  # both lanes warm (first calls allocate)
        time.sleep(0.005)
    beside, bad_preview = [], 0
    for i in range(120):
        t0 = time.perf_counter()
        prev = pipe.render_to_bytes()
        beside.append((time.perf_counter() - t0) * 1e3)
        if i % 20 == 0 and not np.array_equal(prev.reshape(exp_prev.shape), exp_prev):
            bad_preview += 1
        time.sleep(0.002)
    n_exports = len(exports)
    stop.set()
    t.join()
    assert not errors, errors
    assert bad_preview == 0
    assert n_exports >= 4, f"only {n_exports} exports ran beside the previews"
    med_alone, med_beside = statistics.median(alone), statistics.median(beside)
    print(f"\npreview alone {med_alone:.3f} ms; beside back-to-back 24 MP exports {med_beside:.3f} ms (max {max(beside):.3f}); "
          f"{n_exports} exports, median {statistics.median(exports):.2f} ms")
    assert med_beside < 1.0, f"preview median {med_beside:.3f} ms while an export runs (alone: {med_alone:.3f} ms)"
    # and the whole 24 MP surface once, against the oracle
    full = pipe.render_full_res_to_bytes(out=pin.array).reshape(h, w, 4)
    assert np.array_equal(full, _oracle8(refc, cfa, params))
    pin.free()
    pipe.close()


def test_export_races_uniform_updates_without_tearing(gpu_lib, refc):
    ra = gpu_lib
    h, w = 2056, 2048
    rng = np.random.default_rng([0x52415745, 78])
    cfa = random_cfa(rng, h, w)
    stacks = [random_params(rng), random_params(rng)]
    exps = [_oracle8(refc, cfa, s) for s in stacks]
    pipe = ra.RenderPipeline.new(5, cfa.reshape(-1), w, h, ra.EditParams(**stacks[0]), WB_DAYLIGHT, CM_TEST)
    stop = threading.Event()

    def drag():
        k = 0
        while not stop.is_set():
            pipe.update_uniforms(ra.EditParams(**stacks[k & 1]))
            k += 1

    t = threading.Thread(target=drag)
    t.start()
    seen = [0, 0]
    try:
        for _ in range(24):
            got = pipe.render_full_res_to_bytes().reshape(h, w, 4)
            which = [np.array_equal(got, e) for e in exps]
            assert any(which), "an export mixed two slider stacks (or matches neither)"
            seen[which.index(True)] += 1
    finally:
        stop.set()
        t.join()
    assert sum(seen) == 24
    pipe.close()


def test_page_locked_destinations_are_detected(gpu_lib):
    """The direct-DMA path is taken for page-locked memory however it was obtained (rd_host_alloc, a HIP host allocation made
    by somebody else -- torch's pin_memory here) and for every sub-range of it; ordinary memory is staged."""
    import ctypes as C
    import torch
    ra = gpu_lib
    from raweditor_amd import _lib
    L = _lib.lib()
    n = 24 << 20
    pin = ra.PinnedBytes(n)
    assert L.rd_debug_is_pinned_host(C.c_void_p(pin.ptr), n) == 1
    assert L.rd_debug_is_pinned_host(C.c_void_p(pin.ptr + 4096), n - 8192) == 1
    t = torch.empty(n, dtype=torch.uint8).pin_memory()
    assert L.rd_debug_is_pinned_host(C.c_void_p(t.data_ptr()), n) == 1
    plain = np.empty(n, np.uint8)
    assert L.rd_debug_is_pinned_host(plain.ctypes.data_as(C.c_void_p), n) == 0
    assert L.rd_debug_is_pinned_host(None, n) == 0
    from tests.gpu_util import DevBuf
    dbuf = DevBuf(n)
    assert L.rd_debug_is_pinned_host(C.c_void_p(dbuf.ptr), n) == 0          # device memory is not a host destination ...
    # a render into torch's pinned tensor (the same bytes as any other destination)
    h, w = 2056, 2048
    rng = np.random.default_rng([0x52415745, 91])
    cfa = random_cfa(rng, h, w)
    pipe = ra.RenderPipeline.new(4, cfa.reshape(-1), w, h, ra.EditParams(), WB_DAYLIGHT, CM_TEST)
    out_t = torch.empty(h * w * 4, dtype=torch.uint8).pin_memory()
    a = pipe.render_full_res_to_bytes(out=out_t.numpy())
    b = pipe.render_full_res_to_bytes()
    assert np.array_equal(a, b)
    # ... and a full-resolution host render refuses it instead of memcpy-ing into it
    assert L.rd_render_full_res_to_bytes(pipe._h, C.c_void_p(dbuf.ptr), h * w * 4) == -1
    assert b"device memory" in L.rd_last_error()
    pin.free()
    pipe.close()


def test_registered_host_windows_and_the_gap_between_two(gpu_lib, refc):
    """a host range whose two ENDS are page-locked but whose middle is not (two hipHostRegister windows with a
    gap, or a buffer larger than its registered window) must not be handed to the DMA engine.  A whole registered window --
    what a Rust host does to a Vec it keeps -- takes the direct path, any sub-range of it too; the range that spans the gap is
    staged; and a full-resolution render into each of them gives the oracle's bytes."""
    import ctypes as C
    ra = gpu_lib
    from raweditor_amd import _lib
    L = _lib.lib()
    path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)      # the HIP runtime this process has mapped
    hip = C.CDLL(path)
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    h, w = 2056, 2048                                            # a 16.1 MiB RGBA8 surface: the band-pipelined read-back
    need = h * w * 4
    page = 4096
    buf = np.zeros(need + 3 * page, np.uint8)
    base = (buf.ctypes.data + page - 1) & ~(page - 1)
    rng = np.random.default_rng([0x52415745, 92])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(5, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    exp = _oracle8(refc, cfa, params).reshape(-1)
    view = np.frombuffer((C.c_uint8 * need).from_address(base), dtype=np.uint8)
    # (1) one window over the whole destination
    assert L.rd_debug_is_pinned_host(C.c_void_p(base), need) == 0
    assert hip.hipHostRegister(C.c_void_p(base), need + page, 0) == 0
    try:
        assert L.rd_debug_is_pinned_host(C.c_void_p(base), need) == 1
        assert L.rd_debug_is_pinned_host(C.c_void_p(base + page), need - 2 * page) == 1
        view[:] = 0x5a
        pipe.render_full_res_to_bytes(out=view)
        assert np.array_equal(view, exp), "registered window (direct DMA)"
    finally:
        assert hip.hipHostUnregister(C.c_void_p(base)) == 0
    # (2) two windows, first and last megabyte, nothing in between: both ends page-locked, the middle pageable
    mb = 1 << 20
    last = (base + need - mb) & ~(page - 1)
    assert hip.hipHostRegister(C.c_void_p(base), mb, 0) == 0
    assert hip.hipHostRegister(C.c_void_p(last), mb + page, 0) == 0
    try:
        assert L.rd_debug_is_pinned_host(C.c_void_p(base), mb) == 1
        assert L.rd_debug_is_pinned_host(C.c_void_p(base), need) == 0, "ends registered, middle not: must be staged"
        assert L.rd_debug_is_pinned_host(C.c_void_p(base), mb + page) == 0, "a range that outgrows its window"
        view[:] = 0xa5
        pipe.render_full_res_to_bytes(out=view)
        assert np.array_equal(view, exp), "range across the gap (staged)"
    finally:
        hip.hipHostUnregister(C.c_void_p(base))
        hip.hipHostUnregister(C.c_void_p(last))
    pipe.close()


def test_six_threads_share_four_render_lanes(gpu_lib, refc):
    """More concurrent host renders than lanes: every call takes a lane for its duration, the fifth and sixth wait for one;
    previews, histogram renders, calculate_histogram, f32 renders with a fused histogram and band-pipelined exports all
    come out bit-identical to the oracle, and the pipeline never holds more than four lanes."""
    ra = gpu_lib
    from raweditor_amd import _lib
    h, w = 2056, 2048
    rng = np.random.default_rng([0x52415745, 92])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(6, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    f32 = refc.render_f32(cfa, u, nthreads=8)
    full8 = refc.pack_u8(f32)
    hist_full = refc.histogram(full8)
    prev8 = _oracle8(refc, cfa, params, tw=pipe.preview_width, th=pipe.preview_height)
    hb8 = _oracle8(refc, cfa, params, tw=pipe.histogram_width, th=pipe.histogram_height)
    hist_small = refc.histogram(hb8)
    bad = []

    def work(k):
        try:
            for it in range(8):
                kind = (k + it) % 5
                if kind == 0:
                    ok = np.array_equal(pipe.render_to_bytes().reshape(prev8.shape), prev8)
                elif kind == 1:
                    hb = pipe.render_to_histogram_bytes()
                    ok = np.array_equal(hb.reshape(hb8.shape), hb8) and np.array_equal(pipe.calculate_histogram(hb), hist_small)
                elif kind == 2:
                    ok = np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), full8)
                elif kind == 3:
                    got, hist = pipe.render(fmt=ra.FMT_RGBA_F32, with_histogram=True)
                    ok = np.array_equal(got.view(np.uint32), f32.view(np.uint32)) and np.array_equal(hist, hist_full)
                else:
                    got, hist = pipe.render(fmt=ra.FMT_RGBA_U8, with_histogram=True)
                    ok = np.array_equal(got, full8) and np.array_equal(hist, hist_full)
                if not ok:
                    bad.append((k, it, kind))
        except Exception as e:  # noqa: BLE001
            bad.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad, bad
    assert 2 <= _lib.lib().rd_debug_lane_count(pipe._h) <= 4
    pipe.close()


def test_borrowed_surface(gpu_lib, refc):
    """rd_render_full_res_borrow: the export into page-locked memory the pipeline owns and lends; the same bytes, the same
    buffer again after a release, at most four out at a time, a foreign or double release is an error."""
    import ctypes as C
    ra = gpu_lib
    from raweditor_amd import _lib
    h, w = 2056, 2048
    rng = np.random.default_rng([0x52415745, 93])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(8, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    exp = _oracle8(refc, cfa, params)
    with pipe.render_full_res_borrowed() as s:
        first = s.ptr
        assert s.nbytes == h * w * 4 and _lib.lib().rd_debug_is_pinned_host(C.c_void_p(s.ptr), s.nbytes) == 1
        assert np.array_equal(s.array.reshape(h, w, 4), exp)
    with pipe.render_full_res_borrowed() as s:
        assert s.ptr == first                                  # reused, not re-allocated
        assert np.array_equal(s.array.reshape(h, w, 4), exp)
    out = [pipe.render_full_res_borrowed() for _ in range(4)]
    assert len({s.ptr for s in out}) == 4
    with pytest.raises(ra.RawdevError):
        pipe.render_full_res_borrowed()                        # a fifth while four are out
    assert all(np.array_equal(s.array.reshape(h, w, 4), exp) for s in out)
    p0 = out[0].ptr
    out[0].release()
    assert _lib.lib().rd_surface_release(pipe._h, C.c_void_p(p0)) == -1          # released twice
    assert _lib.lib().rd_surface_release(pipe._h, C.c_void_p(p0 + 64)) == -1     # not a surface of this pipeline
    for s in out[1:]:
        s.release()
    pipe.close()


def test_pipeline_per_image_does_not_leak(gpu_lib, refc):
    """The reference builds a new RenderPipeline for every image it opens (main.rs:993-1006, drop at :1028): forty pipelines,
    each used the way the app uses one (preview, histogram render, a staged export, a lent surface, a direct-DMA export),
    then dropped -- device memory ends where it started (lanes, staging slots, lent surfaces, scheduler state, the graph
    cache and the CFA copy all go with the pipeline)."""
    import ctypes as C
    ra = gpu_lib
    from raweditor_amd import _lib

    def free_mb():
        f, t = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.lib().rd_device_memory(0, C.byref(f), C.byref(t)))
        return f.value / 2**20

    h, w = 2056, 2048
    rng = np.random.default_rng([0x52415745, 94])
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    exp = _oracle8(refc, cfa, params)
    pin = ra.PinnedBytes(h * w * 4)

    def one(i):
        pipe = ra.RenderPipeline.new(i, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
        pipe.render_to_bytes()
        pipe.calculate_histogram(pipe.render_to_histogram_bytes())
        a = pipe.render_full_res_to_bytes()
        with pipe.render_full_res_borrowed() as s:
            ok = np.array_equal(s.array, a)
        pipe.render_full_res_to_bytes(out=pin.array)
        ok = ok and np.array_equal(pin.array, a) and (i % 10 or np.array_equal(a.reshape(h, w, 4), exp))
        pipe.close()
        return ok

    assert one(0)                                              # first use: library-level one-time allocations
    before = free_mb()
    assert all(one(i) for i in range(1, 41))
    after = free_mb()
    assert before - after < 64, f"device memory shrank by {before - after:.0f} MiB over 40 pipelines"
    pin.free()


KNOB_SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import raweditor_amd as ra
from oracle import ref_c as refc
from tests.gpu_util import DevBuf
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params
h, w = 2056, 2048
rng = np.random.default_rng([0x52415745, 95])
cfa = random_cfa(rng, h, w)
params = random_params(rng)
u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
f32 = refc.render_f32(cfa, u, nthreads=8)
exp = refc.pack_u8(f32)
pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
pin = ra.PinnedBytes(h * w * 4)
assert np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), exp)
assert np.array_equal(pipe.render_full_res_to_bytes(out=pin.array).reshape(h, w, 4), exp)
got, hist = pipe.render(fmt=ra.FMT_RGBA_F32, with_histogram=True)
assert np.array_equal(got.view(np.uint32), f32.view(np.uint32)) and np.array_equal(hist, refc.histogram(exp))
out, hd = DevBuf(h * w * 16), DevBuf(768 * 4)
for _ in range(3):                                            # rd_render_device: develop + fold (RD_GRAPH=1: as one graph, re-parameterised)
    pipe.render_device(w, h, ra.FMT_RGBA_F32, out.ptr, hd.ptr)
    pipe.update_uniforms(ra.EditParams(**params))
from tests.gpu_util import sync
sync()
assert np.array_equal(out.to_array(np.float32, (h, w, 4)).view(np.uint32), f32.view(np.uint32))
assert np.array_equal(hd.to_array(np.uint32, (768,)).reshape(3, 256), refc.histogram(exp))
print("knobs ok")
"""


@pytest.mark.parametrize("env", [{"RD_COPY_THREADS": "0"}, {"RD_COPY_THREADS": "9"}, {"RD_ASSUME_PAGEABLE": "1"}, {"RD_RENDER_BANDS": "1"},
                                 {"RD_RENDER_BANDS": "3", "RD_COPY_CHUNK_MB": "4"}, {"RD_DST_ADVISE": "huge"}, {"RD_DST_ADVISE": "populate"},
                                 {"RD_GRAPH": "1"}], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_readback_knobs_do_not_change_bytes(gpu_lib, env):
    """The host-side switches of the read-back (helper threads, forced staging, band count, fixed piece size, madvise hints)
    and the graph form of develop + fold are read once per process: each runs in a fresh interpreter and must give the
    oracle's bytes for a pageable and a page-locked destination, a fused-histogram render and rd_render_device."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", KNOB_SCRIPT % {"root": root}], env=dict(os.environ, **env), capture_output=True,
                         text=True, timeout=300, cwd=root)
    assert out.returncode == 0 and "knobs ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
