"""oracle/ -- the CPU checkers of the develop path (TEST INFRASTRUCTURE; the product never imports this package).

develop_ref.c / ref_c.py   scalar C restatement of the reference's shader and host arithmetic (the checker, and bench.py's
                           cpu_baseline leg)
develop_np.py              an independently written numpy twin
wgsl_vec.py                the same evaluator over arrays of fragments (whole frames: tools/make_wgsl_fullsize.py ->
                           tests/golden/wgsl_fullsize.json)
wgsl_eval.py               a WGSL evaluator written from the WGSL specification; wgsl_render.py draws the reference's
                           full-screen triangle with it.  tools/make_wgsl_golden.py runs the reference's own shader text through
                           it (read where it lies under /root/reference, never copied) and commits the vectors both restatements
                           and the HIP path must reproduce bit for bit: tests/golden/wgsl_golden.npz
"""
