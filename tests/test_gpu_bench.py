"""-m gpu: bench.py's entry points exactly as a driver would call them, small workloads, on the one-GPU box.

* `python bench.py --gpus 2 ...` with NO launcher around it: the parent starts two fresh rank processes itself
  (RAWDEV_DIST_BACKEND=gloo: both ranks share device 0 and the histogram all-reduce goes over gloo -- RCCL refuses two
  ranks on one device; the driver's multi-GPU runs use nccl on distinct devices, which has never run on this pool's
  one-GPU boxes: N > 1 on distinct devices stays UNMEASURED).
* `python bench.py --host node --gpus 2 ...`: ONE process, rd_node_batch_* (RD_NODE_REDUCE=host: device 0 listed twice).
* N = 1 with `extra_configs` on a reduced batch.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, extra_env, timeout=900):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bare_multi_gpu_call_launches_its_own_ranks(gpu_lib):
    r = _bench(["--gpus", "2", "--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
               {"RAWDEV_DIST_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["verified"] is True
    assert r["config"]["frames_per_gpu"] == 8 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["roofline"]["frac"] > 0
    assert "extra_configs" not in r and "cpu_baseline" not in r      # N = 1 only
    assert r["allreduce_us"]["median"] > 0 and r["allreduce_us"]["bytes"] == 6144     # reported separately


def test_node_host_mode_one_process(gpu_lib):
    r = _bench(["--host", "node", "--gpus", "2", "--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
               {"RD_NODE_REDUCE": "host"})
    assert r["n_gpus"] == 2 and r["verified"] is True
    assert "rd_node_batch" in r["config"]["host"] and "REHEARSAL" in r["config"]["host"]
    assert r["histogram_call_us"]["median"] > 0 and "host fold" in r["histogram_call_us"]["reduction"]
    r1 = _bench(["--host", "node", "--gpus", "1", "--frames", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {})
    assert r1["n_gpus"] == 1 and r1["verified"] is True and "REHEARSAL" not in r1["config"]["host"]


def test_single_gpu_line_carries_the_other_configs(gpu_lib):
    r = _bench(["--frames", "16", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"], {})
    assert r["n_gpus"] == 1 and r["verified"] is True
    assert r["config"]["workload"].startswith("BASELINE configs[2]")
    assert "uploads its descriptors" in r["config"]["descriptors"]
    ex = r["extra_configs"]
    assert "error" not in ex, ex
    for key in ("single_frame_f32", "batch_rgba8", "config5_shape_f16"):
        e = ex[key]
        assert e["verified"] is True, (key, e)
        assert e["ms"] > 0 and e["MP_per_s"] > 0
        assert 0 < e["roofline"]["frac"] < 1 and e["roofline"]["achieved"] > 0
    assert r["cpu_baseline"]["kind"] == "port"
