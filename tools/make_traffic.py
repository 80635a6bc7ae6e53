#!/usr/bin/env python3
"""pmc_FETCH_SIZE.txt + pmc_WRITE_SIZE.txt (tools/pmc_summary.py output) -> traffic json read by bench.py.

usage: make_traffic.py <dir with pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt> <pairs per launch> <out.json> [build id]
Counter values are KB per launch (mean over launches); a stage's launches per step (the pyramid has several) are summed:
launches per step = the kernel's dispatch count / fast_cell_kernel's (one per step).
"""
import ast, json, sys

# first match wins (pyr_resize_blur_kernel also contains "blur_kernel"): the fused resize + blur launches count as "pyramid", so "blur" is the
# launch for the levels that are not blurred beside a resize (4-7 at KITTI geometry)
STAGE_OF = {"ingest_kernel": "ingest", "ingest16_kernel": "ingest", "pyr_resize": "pyramid", "pyr_tail_kernel": "pyramid", "blur_kernel": "blur", "fast_cell_kernel": "fast",
            "octree": "octree", "describe_kernel": "describe", "stereo_match_kernel": "stereo_match",
            "stereo_rowtable_kernel": "stereo_match", "stereo_rowlist_kernel": "stereo_match", "stereo_median_kernel": "stereo_median"}


def read(path, counter):
    rows = []
    for line in open(path):
        if "{" not in line:
            continue
        name = line[:line.index("{")].strip()
        vals = ast.literal_eval(line[line.index("{"):line.rindex("}") + 1])
        n = int(line[line.rindex("n=") + 2:]) if "n=" in line[line.rindex("}"):] else 1
        rows.append((name, vals, n))
    steps = max([n for name, _, n in rows if "fast_cell_kernel" in name] or [1])
    out = {}
    for name, vals, n in rows:
        for k, st in STAGE_OF.items():
            if k in name and "gather" not in name:
                out[st] = out.get(st, 0.0) + vals[counter] * (n / steps)
                break
    return out


d, pairs, dst = sys.argv[1], float(sys.argv[2]), sys.argv[3]
f, w = read(d + "/pmc_FETCH_SIZE.txt", "FETCH_SIZE"), read(d + "/pmc_WRITE_SIZE.txt", "WRITE_SIZE")
json.dump({
    "build_id": sys.argv[4] if len(sys.argv) > 4 else None,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 "
              "--no-check; KB per launch averaged over launches, %g pairs per launch" % pairs,
    "correction": "gfx950: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section); narrow access widths are uncalibrated",
    "kernels": {s: {"fetch_kb_per_pair": f[s] / pairs, "write_kb_per_pair": w.get(s, 0.0) / pairs} for s in sorted(f)},
}, open(dst, "w"), indent=1)
print(open(dst).read())
