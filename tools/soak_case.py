#!/usr/bin/env python3
"""Re-run ONE case of tools/soak.py and say which outputs differ.  usage: SOAK_SEED=.. [SOAK_GEOM=1] python3 tools/soak_case.py <case index>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslam2_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

i = int(sys.argv[1])
seed = int(os.environ.get("SOAK_SEED", "7000"))
src = open(os.path.join(ROOT, "tools", "soak.py")).read()
body = src[src.index("    rng = np.random.default_rng("):src.index("    ok = (np.array_equal(")]
body = "\n".join(l[4:] if l.startswith("    ") else l for l in body.split("\n")).replace("continue", "raise SystemExit('refused')")
ns = dict(np=np, os=os, api=api, synth=synth, O=O, i=i, sys=sys)
exec(body, ns)
out, kl, kr, dl, dr, ur, dp = (ns[k] for k in ("out", "kl", "kr", "dl", "dr", "ur", "dp"))
print("case", i, ns["w"], ns["h"], ns["kw"], "keypoints", len(kl), len(kr))
for name, a, b in (("kps_left", out["kps_left"], kl.astype(api.KP_DTYPE)), ("kps_right", out["kps_right"], kr.astype(api.KP_DTYPE)), ("desc_left", out["desc_left"], dl),
                   ("desc_right", out["desc_right"], dr), ("u_right", out["u_right"], ur), ("depth", out["depth"], dp)):
    same = a.shape == b.shape and np.array_equal(a, b)
    print(name, "equal" if same else "DIFFER", a.shape, b.shape)
    if not same and a.shape == b.shape:
        bad = np.argwhere(a != b) if a.dtype.names is None else np.argwhere(np.array([x != y for x, y in zip(a, b)]))
        print("  first differing indices", bad[:8].ravel().tolist(), "of", len(bad))
        for j in bad[:4].ravel():
            print("   ", j, a[j], b[j], "| left kp", kl[j] if name in ("u_right", "depth") else "")
