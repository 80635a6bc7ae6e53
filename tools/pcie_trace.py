#!/usr/bin/env python3
"""Driver for a rocprofv3 --kernel-trace --memory-copy-trace timeline of the host-fed lanes (tools/r04_pcie_trace.sh).
usage: pcie_trace.py <down mode> <lanes> [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, args = sys.argv[:1], sys.argv[1:]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("pr2", os.path.join(ROOT, "tools", "pcie_rate2.py"))
src = open(os.path.join(ROOT, "tools", "pcie_rate2.py")).read().split("out = {")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "pcie_rate2.py")}
exec(compile(src, "pcie_rate2_head", "exec"), ns)
down, n = args[0], int(args[1])
steps = int(args[2]) if len(args) > 2 else 24
lanes = [ns["Lane"](down, "copy") for _ in range(n)]
for l in lanes:
    l.step()
torch.cuda.synchronize()
for k in range(steps):
    lanes[k % n].step()
torch.cuda.synchronize()
