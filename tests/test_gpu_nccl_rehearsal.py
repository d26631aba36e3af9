"""-m gpu: bench.py's N > 1 collectives on a one-rank nccl (= RCCL) process group, in a child process (the process group is
global state).  See tools/nccl_one_rank.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_collectives_on_a_one_rank_rccl_group():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_one_rank.py")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "nccl one-rank rehearsal ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
