#!/usr/bin/env python3
"""Randomised stress of the round-4 host side (render lanes, band-pipelined read-back, staging slots, copy pool, lent
surfaces): several threads call every render entry of SHARED pipelines at once -- full-resolution exports into every kind
of destination, previews, histogram thumbnails, rd_calculate_histogram, the general rd_render in all four surface formats
with and without the fused histogram -- and every result is compared, bit for bit, with the CPU oracle's answer for the
uniforms of that epoch.  Uniforms change only between epochs (a barrier), so every call has exactly one right answer;
the race of an export against rd_update_uniforms is tests/test_gpu_fullres.py's subject.

    python tools/stress_fullres.py [seconds=120] [threads=6] [seed=1]       (GPU box; exit 1 on the first mismatch)

Test infrastructure like tests/: the oracle is the checker here.
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raweditor_amd as ra
from oracle import ref_c as refc
from tests.helpers import CM_IDENTITY, CM_TEST, WB_DAYLIGHT, random_cfa, random_params

SIZES = [(2056, 2048), (1500, 3074), (2009, 2304), (2731, 4096), (4016, 6016), (1030, 4224)]     # (h, w): 16-92 MiB as RGBA8


class Frame:
    def __init__(self, rng, idx, h, w):
        self.h, self.w, self.idx = h, w, idx
        self.cfa = random_cfa(rng, h, w, 65536 if idx % 3 == 2 else 4096)
        self.cm = CM_IDENTITY if idx % 2 else CM_TEST
        self.pipe = ra.RenderPipeline.new(100 + idx, self.cfa.reshape(-1), w, h, ra.EditParams(), WB_DAYLIGHT, self.cm)
        self.pin = ra.PinnedBytes(h * w * 4)
        self.pin_lock = threading.Lock()
        self.exp = {}

    def new_epoch(self, rng):
        params = random_params(rng)
        if rng.random() < 0.3:                                    # a channel-separable stack (the usual edit)
            for k in ("highlights", "shadows", "vibrance", "saturation"):
                params[k] = 0.0
        self.params = params
        self.pipe.update_uniforms(ra.EditParams(**params))
        u = refc.make_uniforms(params, WB_DAYLIGHT, self.cm)
        f32 = refc.render_f32(self.cfa, u, nthreads=16)
        p = self.pipe
        self.exp = {
            "f32": f32,
            "u8": refc.pack_u8(f32),
            "f16": refc.pack_f16(f32),
            "preview": refc.pack_u8(refc.render_f32(self.cfa, u, p.preview_width, p.preview_height, nthreads=16)),
            "thumb": refc.pack_u8(refc.render_f32(self.cfa, u, p.histogram_width, p.histogram_height, nthreads=16)),
        }
        self.exp["hist"] = refc.histogram(self.exp["u8"])
        self.exp["thumb_hist"] = refc.histogram(self.exp["thumb"])

    def small(self, tw, th):
        u = refc.make_uniforms(self.params, WB_DAYLIGHT, self.cm)
        return refc.pack_u8(refc.render_f32(self.cfa, u, tw, th, nthreads=2))


def one_call(rng, fr, scratch):
    """One random entry on frame fr; returns (name, ok)."""
    p, e, h, w = fr.pipe, fr.exp, fr.h, fr.w
    k = int(rng.integers(0, 12))
    if k == 0:
        return "full fresh", np.array_equal(p.render_full_res_to_bytes().reshape(h, w, 4), e["u8"])
    if k == 1:
        buf = scratch.setdefault((h, w), np.empty(h * w * 4, np.uint8))
        buf[:: 4096] = 0x5a
        return "full reused", np.array_equal(p.render_full_res_to_bytes(out=buf).reshape(h, w, 4), e["u8"])
    if k == 2:
        if not fr.pin_lock.acquire(blocking=False):
            return "full pinned (busy)", True
        try:
            fr.pin.array[:: 4096] = 0xa5
            p.render_full_res_to_bytes(out=fr.pin.array)
            return "full pinned", np.array_equal(fr.pin.array.reshape(h, w, 4), e["u8"])
        finally:
            fr.pin_lock.release()
    if k == 3:
        try:
            s = p.render_full_res_borrowed()
        except ra.RawdevError as ex:                              # all lendable surfaces are out (RD_LENT_MAX): a refusal, not a fault
            return "full borrowed (refused)", "have not been released" in str(ex)
        with s:
            return "full borrowed", np.array_equal(s.array.reshape(h, w, 4), e["u8"])
    if k == 4:
        return "preview", np.array_equal(p.render_to_bytes().reshape(e["preview"].shape), e["preview"])
    if k == 5:
        t = p.render_to_histogram_bytes()
        ok = np.array_equal(t.reshape(e["thumb"].shape), e["thumb"])
        return "thumb + calculate_histogram", ok and np.array_equal(p.calculate_histogram(t), e["thumb_hist"])
    if k == 6:
        got, hist = p.render(fmt=ra.FMT_RGBA_U8, with_histogram=True)
        return "render u8 + hist", np.array_equal(got, e["u8"]) and np.array_equal(hist, e["hist"])
    if k == 7:
        got = p.render(fmt=ra.FMT_RGBA_F32)
        return "render f32", np.array_equal(got.view(np.uint32), e["f32"].view(np.uint32))
    if k == 8:
        got, hist = p.render(fmt=ra.FMT_RGBA_F16, with_histogram=True)
        return "render f16 + hist", (np.array_equal(got.view(np.uint16), e["f16"].view(np.uint16))
                                     and np.array_equal(hist, e["hist"]))
    if k == 9:
        return "render rgb8", np.array_equal(p.render(fmt=ra.FMT_RGB_U8), e["u8"][..., :3])
    if k == 10:
        return "calculate_histogram(full)", np.array_equal(p.calculate_histogram(e["u8"]), e["hist"])
    tw, th = int(rng.integers(1, 900)), int(rng.integers(1, 700))                 # a small target through the map kernel
    return "render small", np.array_equal(p.render(tw, th, fmt=ra.FMT_RGBA_U8), fr.small(tw, th))


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng([0x52345354, seed])
    picks = rng.choice(len(SIZES), 3, replace=False)
    frames = [Frame(rng, i, *SIZES[int(k)]) for i, k in enumerate(picks)]
    counts, failures, lock = {}, [], threading.Lock()
    t_end = time.time() + secs
    epoch = 0
    lanes_seen = 0
    while time.time() < t_end and not failures:
        for fr in frames:
            fr.new_epoch(np.random.default_rng([seed, epoch, fr.idx]))
        epoch_end = min(t_end, time.time() + 8.0)

        def worker(tid):
            wrng = np.random.default_rng([seed, epoch, 1000 + tid])
            scratch = {}
            while time.time() < epoch_end and not failures:
                fr = frames[int(wrng.integers(0, len(frames)))]
                try:
                    name, ok = one_call(wrng, fr, scratch)
                except Exception as ex:                           # an error code is a failure too
                    name, ok = f"exception {type(ex).__name__}: {ex}", False
                with lock:
                    counts[name] = counts.get(name, 0) + 1
                    if not ok:
                        failures.append((epoch, tid, fr.idx, (fr.h, fr.w), name))

        ts = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        lanes_seen = max(lanes_seen, max(ra._lib.lib().rd_debug_lane_count(fr.pipe._h) for fr in frames))
        epoch += 1
        print(f"epoch {epoch}: {sum(counts.values())} calls so far, lanes <= {lanes_seen}", flush=True)
    total = sum(counts.values())
    sizes = ", ".join(f"{fr.w}x{fr.h}" for fr in frames)
    print(f"stress_fullres seed {seed}: {nthreads} threads x {epoch} epochs on 3 shared pipelines ({sizes}), {total} calls "
          f"against the oracle: {len(failures)} mismatching")
    for name in sorted(counts):
        print(f"  {counts[name]:6d}  {name}")
    for f in failures[:10]:
        print("  MISMATCH", f)
    for fr in frames:
        fr.pin.free()
        fr.pipe.close()
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
