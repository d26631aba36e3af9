#!/usr/bin/env python3
"""The export ring fed from host memory (rd_exporter_submit_host) beside the device-resident feed: ms per 24 MP frame for
page-locked and pageable CFA planes, RGB8 and RGBA8 (bench.py's extra_export_ring, on its own).

    python tools/bench_ring_host.py [frames=48]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import raweditor_amd as ra


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    dev = torch.device("cuda:0")
    cfas, params = bench.make_batch(torch, np, ra, dev, 6016, 4016, 8, 1 << 20, 1)
    out = bench.extra_export_ring(torch, np, ra, dev, 0, cfas, params, n_frames=n)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
