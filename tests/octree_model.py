"""Array ("workgroup") formulation of DistributeOctTree used by the HIP kernel.

Test infrastructure: a pure-Python model of the data-parallel restatement that
orbslam2_amd/csrc implements on the GPU (node array kept in std::list order,
children written n4..n1 in front, stable 4-way partition of each node's points,
"largest node first" phase as a sort on (count desc, list position asc)).
tests/test_octree_model.py checks it against the literal list-based oracle
(oracle/orb_oracle.c: orc_distribute_octtree, reference src/ORBextractor.cc:533-757),
so a mistake in the reformulation is caught on the CPU before any kernel runs.
"""
from __future__ import annotations

import math

import numpy as np


def _f32(v):
    return np.float32(v)


def distribute(xs, ys, scores, min_x, max_x, min_y, max_y, n_features):
    xs = np.asarray(xs, np.int64); ys = np.asarray(ys, np.int64); scores = np.asarray(scores, np.int64)
    n = len(xs)
    w = max_x - min_x
    h = max_y - min_y
    # roundf(float(w)/float(h)) : half away from zero
    q = float(_f32(w) / _f32(h))
    n_ini = int(math.floor(q + 0.5))
    if n_ini < 1:
        n_ini = 1
    hx = _f32(w) / _f32(n_ini)
    # nodes: list of dicts in list order (front -> back)
    nodes = []
    bucket = np.zeros(n, np.int64)
    for i in range(n):
        b = int(_f32(xs[i]) / hx)
        bucket[i] = min(max(b, 0), n_ini - 1)
    for i in range(n_ini):
        pts = [k for k in range(n) if bucket[k] == i]
        if pts:
            nodes.append(dict(x0=int(hx * _f32(i)), x1=int(hx * _f32(i + 1)), y0=0, y1=h, pts=pts))

    def split(nd):
        x0, y0, x1, y1 = nd["x0"], nd["y0"], nd["x1"], nd["y1"]
        hxh = int(math.ceil(float(_f32(x1 - x0) / _f32(2))))
        hyh = int(math.ceil(float(_f32(y1 - y0) / _f32(2))))
        mx, my = x0 + hxh, y0 + hyh
        ch = [dict(x0=x0, y0=y0, x1=mx, y1=my, pts=[]), dict(x0=mx, y0=y0, x1=x1, y1=my, pts=[]),
              dict(x0=x0, y0=my, x1=mx, y1=y1, pts=[]), dict(x0=mx, y0=my, x1=x1, y1=y1, pts=[])]
        for k in nd["pts"]:
            c = (0 if xs[k] < mx else 1) + (0 if ys[k] < my else 2)
            ch[c]["pts"].append(k)
        return ch

    while True:
        prev = len(nodes)
        multi = [i for i, nd in enumerate(nodes) if len(nd["pts"]) > 1]
        children = {i: split(nodes[i]) for i in multi}
        # full pass: every multi-point node, in list order
        front = []
        for i in multi:  # processing order; later-processed blocks end up nearer the front
            blk = [c for c in reversed(children[i]) if c["pts"]]
            front = blk + front
        rest = [nd for i, nd in enumerate(nodes) if i not in children]
        nodes = front + rest
        n_to_expand = sum(1 for nd in front if len(nd["pts"]) > 1)
        if len(nodes) >= n_features or len(nodes) == prev:
            break
        if len(nodes) + 3 * n_to_expand > n_features:
            done = False
            while not done:
                prev = len(nodes)
                multi = [i for i, nd in enumerate(nodes) if len(nd["pts"]) > 1]
                order = sorted(multi, key=lambda i: (-len(nodes[i]["pts"]), i))
                size = len(nodes)
                processed = []
                for i in order:
                    ch = [c for c in split(nodes[i]) if c["pts"]]
                    processed.append((i, ch))
                    size += len(ch) - 1
                    if size >= n_features:
                        break
                front = []
                pset = set()
                for i, ch in processed:
                    front = list(reversed(ch)) + front
                    pset.add(i)
                rest = [nd for i, nd in enumerate(nodes) if i not in pset]
                nodes = front + rest
                if len(nodes) >= n_features or len(nodes) == prev:
                    done = True
            break
    out = []
    for nd in nodes:
        best = nd["pts"][0]
        for k in nd["pts"][1:]:
            if scores[k] > scores[best]:
                best = k
        out.append(best)
    return np.array(out, np.int32)
