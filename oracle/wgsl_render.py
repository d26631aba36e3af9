"""oracle/wgsl_render.py -- draw the reference's full-screen triangle with its own shader text (TEST INFRASTRUCTURE ONLY).

Runs a WGSL module shaped like the reference's (/root/reference/src/gpu/shaders.rs:14-267: entry points `vs_main(vertex_index)`
and `fs_main(VertexOutput)`, bindings `input_texture` = the CFA plane as texture_2d<u32> and `params` = the uniform block that
gpu/pipeline.rs:373-388 fills) through oracle/wgsl_eval.py, one fragment per output pixel -- what wgpu's render pass of
pipeline.rs:567-590 does with `draw(0..3, 0..1)` on a tw x th target.

The rasteriser between the two stages is fixed-function hardware; two models of it are offered:
  "pixel_centre_f32"  tex_coords = vs_main's arithmetic carried out at the fragment's own position, in f32, in the order the
                      shader writes it: s = (i + 0.5) / tw;  tex = ((s - 0.5) / zoom - pan) + 0.5 .  This is the choice the
                      oracle pinned ("pixel-centre sampling"); `check_vertex_stage` verifies against the
                      shader text that vs_main IS that affine map (it runs vs_main at the three vertices).
  "barycentric_f64"   run vs_main at the three vertices, interpolate tex_coords at the pixel centre with exact-to-binary64
                      barycentric weights, round once to f32 -- a model that takes nothing from the oracle at all.
The two differ by an ulp here and there in tex_coords and therefore only where a pixel boundary is hit within that ulp.
"""
from __future__ import annotations

import math

import numpy as np

from . import wgsl_eval as we

F32 = np.float32
SLIDERS = ("exposure", "contrast", "highlights", "shadows", "whites", "blacks", "vibrance", "saturation", "temperature", "tint")


def uniform_block(params=None, wb=(1, 1, 1, 1), cm=(1, 0, 0, 0, 1, 0, 0, 0, 1), zoom=1.0, pan_x=0.0, pan_y=0.0):
    """The uniform block as the reference's host fills it: GpuEditParams::from (pipeline.rs:48-68: the ten sliders as they
    are, defaults of state/edit.rs:81-95) + update_uniforms_with_zoom (pipeline.rs:373-388: wb, the flat matrix split into
    three rows of three, zoom, pan)."""
    p = dict(exposure=0.0, contrast=0.0, highlights=0.0, shadows=0.0, whites=1.0, blacks=0.0, vibrance=0.0, saturation=0.0,
             temperature=0.0, tint=0.0)
    for k, v in (params or {}).items():
        if k not in p:
            raise KeyError(k)
        p[k] = v
    cm = [float(x) for x in cm]
    p.update(wb_multipliers=[float(x) for x in wb], color_matrix_0=cm[0:3], color_matrix_1=cm[3:6], color_matrix_2=cm[6:9],
             zoom=zoom, pan_x=pan_x, pan_y=pan_y)
    return p


def tex_coords_pixel_centre_f32(tw, th, zoom, pan_x, pan_y):
    """(th, tw, 2) float32: the pinned rasteriser model (see the module docstring)."""
    zoom, pan_x, pan_y = F32(zoom), F32(pan_x), F32(pan_y)
    sx = (np.arange(tw, dtype=F32) + F32(0.5)) / F32(tw)
    sy = (np.arange(th, dtype=F32) + F32(0.5)) / F32(th)
    with np.errstate(all="ignore"):
        tx = ((sx - F32(0.5)) / zoom - pan_x) + F32(0.5)
        ty = ((sy - F32(0.5)) / zoom - pan_y) + F32(0.5)
    out = np.empty((th, tw, 2), F32)
    out[..., 0] = tx[None, :]
    out[..., 1] = ty[:, None]
    return out


def run_vertex_stage(mod, vertex_fn="vs_main"):
    """-> [(clip xy as python floats, tex_coords as python floats)] for vertex_index 0, 1, 2."""
    out = []
    for vi in range(3):
        vo = mod.call(vertex_fn, we.Sc("u32", vi))
        clip, tex = vo.f["clip_position"], vo.f["tex_coords"]
        if float(clip.c[3]) != 1.0:
            raise we.WgslError("the vertex stage is expected to emit w = 1 (no perspective)")
        out.append(((float(clip.c[0]), float(clip.c[1])), (float(tex.c[0]), float(tex.c[1]))))
    return out


def tex_coords_barycentric_f64(verts, tw, th):
    """(th, tw, 2) float32 from the three vertex outputs: the pixel centre (i + 0.5, j + 0.5) of a tw x th viewport sits at
    NDC (2 (i + 0.5) / tw - 1, 1 - 2 (j + 0.5) / th) (framebuffer y points down)."""
    (p0, t0), (p1, t1), (p2, t2) = verts
    det = (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])
    if det == 0:
        raise we.WgslError("degenerate triangle")
    out = np.empty((th, tw, 2), F32)
    for j in range(th):
        yn = 1.0 - 2.0 * (j + 0.5) / th
        for i in range(tw):
            xn = 2.0 * (i + 0.5) / tw - 1.0
            w1 = ((xn - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (yn - p0[1])) / det
            w2 = ((p1[0] - p0[0]) * (yn - p0[1]) - (xn - p0[0]) * (p1[1] - p0[1])) / det
            w0 = 1.0 - w1 - w2
            if min(w0, w1, w2) < -1e-12:
                raise we.WgslError(f"pixel ({i}, {j}) is outside the triangle: the draw would not cover the target")
            out[j, i, 0] = F32(w0 * t0[0] + w1 * t1[0] + w2 * t2[0])
            out[j, i, 1] = F32(w0 * t0[1] + w1 * t1[1] + w2 * t2[1])
    return out


def check_vertex_stage(verts, zoom, pan_x, pan_y, tol=4e-7):
    """The three vertices of vs_main against the affine map the pinned rasteriser model assumes: NDC position (x, -y) and
    tex = ((x + 1) / 2 - 0.5) / zoom - pan + 0.5 per axis with (x, y) = (-1, -1), (3, -1), (-1, 3).  Raises on a mismatch."""
    for (clip, tex), (x, y) in zip(verts, ((-1.0, -1.0), (3.0, -1.0), (-1.0, 3.0))):
        if clip != (x, -y):
            raise we.WgslError(f"vertex position {clip} is not ({x}, {-y})")
        for got, ndc, pan in ((tex[0], x, pan_x), (tex[1], y, pan_y)):
            with np.errstate(all="ignore"):                      # zoom = 0: the quotient is an infinity (or a NaN), also in vs_main
                want = float(np.float64((ndc + 1.0) * 0.5 - 0.5) / np.float64(F32(zoom)) - np.float64(F32(pan)) + 0.5)
            if abs(want) > 3.4028235677973366e38:                # beyond f32: the shader's own quotient has overflowed
                want = math.copysign(math.inf, want)
            if math.isnan(want) or math.isinf(want):
                if not (math.isnan(got) and math.isnan(want)) and got != want:
                    raise we.WgslError(f"vertex tex coordinate {got} differs from the affine model's {want}")
            elif not abs(got - want) <= tol * max(1.0, abs(want)):
                raise we.WgslError(f"vertex tex coordinate {got} differs from the affine model's {want}")


def bind(source, cfa, uniforms, lowering, texture="input_texture", block="params"):
    mod = we.Module(source, lowering)
    tex = we.Texture2D(np.asarray(cfa, np.uint16))
    mod.bind(texture, tex)
    mod.bind(block, uniforms)
    return mod, tex


def render(source, cfa, uniforms, tw=None, th=None, lowering=None, raster="pixel_centre_f32",
           vertex_fn="vs_main", fragment_fn="fs_main"):
    """-> dict(rgba = (th, tw, 4) float32, tex = (th, tw, 2) float32, oob_loads, nan_to_int)."""
    h, w = np.asarray(cfa).shape
    tw, th = (w if tw is None else int(tw)), (h if th is None else int(th))
    if lowering is None:
        lowering = we.Lowering(pow=we.pow_f64_rounded)
    mod, tex = bind(source, cfa, uniforms, lowering)
    verts = run_vertex_stage(mod, vertex_fn)
    check_vertex_stage(verts, uniforms["zoom"], uniforms["pan_x"], uniforms["pan_y"])
    if raster == "pixel_centre_f32":
        tc = tex_coords_pixel_centre_f32(tw, th, uniforms["zoom"], uniforms["pan_x"], uniforms["pan_y"])
    elif raster == "barycentric_f64":
        tc = tex_coords_barycentric_f64(verts, tw, th)
    else:
        raise ValueError(raster)
    out = np.empty((th, tw, 4), F32)
    varying = mod.fns[fragment_fn][0][0][1].name              # the struct type of the fragment entry point's parameter
    for j in range(th):
        for i in range(tw):
            frag = mod.make_struct(varying, dict(clip_position=[i + 0.5, j + 0.5, 0.0, 1.0],
                                                 tex_coords=[tc[j, i, 0], tc[j, i, 1]]))
            rgba = mod.call(fragment_fn, frag)
            out[j, i] = rgba.c
    return dict(rgba=out, tex=tc, oob_loads=tex.oob_loads, nan_to_int=mod.nan_to_int)


# ---------------------------------------------------------------------------------------------------------------------------
# whole frames: the same draw through the many-fragments-at-a-time evaluator (oracle/wgsl_vec.py)
# ---------------------------------------------------------------------------------------------------------------------------
def render_rows(source, cfa, uniforms, lowering, row0, row1, tw=None, th=None, fragment_fn="fs_main", module=None):
    """Rows [row0, row1) of the tw x th target as ((row1 - row0), tw, 4) float32, rasteriser model "pixel_centre_f32", all
    fragments of the band evaluated at once.  `lowering.pow` must map an array and one exponent to an array.  Pass the
    returned module back in (`module=`) to reuse the parsed shader and its bindings for the next band.
    -> (rgba, module)"""
    from . import wgsl_vec as wv
    h, w = np.asarray(cfa).shape
    tw, th = (w if tw is None else int(tw)), (h if th is None else int(th))
    if module is None:
        module = wv.VectorModule(source, lowering)
        module.texture = we.Texture2D(np.asarray(cfa, np.uint16))
        module.bind("input_texture", module.texture)
        module.bind("params", uniforms)
        check_vertex_stage(run_vertex_stage(module), uniforms["zoom"], uniforms["pan_x"], uniforms["pan_y"])
    zoom, pan_x, pan_y = F32(uniforms["zoom"]), F32(uniforms["pan_x"]), F32(uniforms["pan_y"])
    sx = (np.arange(tw, dtype=F32) + F32(0.5)) / F32(tw)
    sy = (np.arange(row0, row1, dtype=F32) + F32(0.5)) / F32(th)
    with np.errstate(all="ignore"):
        tx = ((sx - F32(0.5)) / zoom - pan_x) + F32(0.5)
        ty = ((sy - F32(0.5)) / zoom - pan_y) + F32(0.5)
    n = (row1 - row0) * tw
    txs = np.ascontiguousarray(np.broadcast_to(tx[None, :], (row1 - row0, tw))).reshape(n)
    tys = np.ascontiguousarray(np.broadcast_to(ty[:, None], (row1 - row0, tw))).reshape(n)
    varying = module.fns[fragment_fn][0][0][1].name
    zero = np.zeros(n, F32)
    frag = module.make_struct(varying, dict(clip_position=[zero, zero, zero, zero], tex_coords=[txs, tys]))
    out = module.call(fragment_fn, frag)
    rgba = np.empty((n, 4), F32)
    for k, comp in enumerate(out.c):
        rgba[:, k] = comp
    return rgba.reshape(row1 - row0, tw, 4), module


def pow_pinned_lanes(x, y):
    """The pinned pow pair (oracle/develop_np.py: pow_pinned) for one value or an array of them, one exponent."""
    from . import develop_np as dn
    if np.ndim(x) == 0:
        return dn.pow_pinned(np.array([x], F32), y)[0]
    return dn.pow_pinned(np.asarray(x, F32), y)
