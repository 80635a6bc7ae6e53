#!/usr/bin/env python3
"""Mix-weighted VALU issue ceiling (round-2 verdict item 4).

gfx950 issues VALU opcodes at two rates (profiles/r02_valu_peak.json, r03_valu_peak.json; tools/ubench/valu_peak*.hip): ~2.3 cycles
per wave-instruction per SIMD for plain add / sub / logic / mov / fp32 mul-add-fma, ~4.1 for everything packed, min / max, shifts,
perm, dot products, 24-bit multiplies ...  Pricing a kernel's SQ_INSTS_VALU as if every instruction were half-rate over-states
how close it is to the issue limit.  This tool

  1. compiles the device code of the hot translation units to assembly (hipcc -S --offload-device-only, the Makefile's flags; for
     fast / describe the -DORBFE_PROFILE_CUTS build, whose cut points leave "; ORBFE_PHASE_END n" markers in the text),
  2. histograms every kernel's VALU opcodes -- per phase where there are markers -- and prices each opcode with its MEASURED
     cycles (unmeasured opcodes: listed, priced half-rate),
  3. weights the phases by their measured dynamic instruction counts (tools/fast_insts.sh, tools/desc_insts.sh: SQ_INSTS_VALU per
     wave at the cut points, profiles/r03_fast_insts.txt / r03_desc_insts.txt); kernels without cut points use their static mix,
  4. lists the scalar instructions by family, which is what explains SQ_INSTS_SALU (FAST: 0.5 per VALU).

The static mix of a phase stands for its dynamic mix (loops dominate every phase and their bodies are homogeneous); that is the
approximation.      python3 tools/isa_mix.py > profiles/r03_isa_mix.json"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orbslam2_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt".split()
UNITS = {  # translation unit -> [(kernel name fragment, report name)]
    "orbfe_fast.hip": [("fast_cell_kernelILi48ELb1E", "fast_cell_kernel<48, true>")],
    "orbfe_describe.hip": [("15describe_kernel", "describe_kernel")],
    "orbfe_pyramid.hip": [("pyr_resize_blur_kernelILi4E", "pyr_resize_blur_kernel<4>"), ("pyr_resize_direct_kernelILi4E", "pyr_resize_direct_kernel<4>"),
                          ("pyr_resize_kernelILi4E", "pyr_resize_kernel<4>"), ("pyr_tail_kernelILi3E", "pyr_tail_kernel<3>"), ("11blur_kernel", "blur_kernel"),
                          ("ingest16_kernel", "ingest16_kernel")],
    "orbfe_stereo.hip": [("stereo_match_kernel", "stereo_match_kernel"), ("stereo_median_kernel", "stereo_median_kernel")],
    "orbfe_octree3.hip": [("octree3_kernelILb0E", "octree3_kernel<false>")],
}
CUTS = {"orbfe_fast.hip", "orbfe_describe.hip"}
PHASE_NAMES = {"fast_cell_kernel<48, true>": {1: "prologue + tile staging", 3: "A: necessary test + queues", 4: "C: exact score", 5: "D / E: NMS, compaction, buckets", 0: "D / E: NMS, compaction, bucket partials"},
               "describe_kernel": {1: "prologue", 2: "slot data + raw patch loads", 3: "moments, angle, first patch staging", 0: "descriptors + records + row lists"}}
ROUND = os.environ.get("ORBFE_PROFILE_ROUND", "r05")  # the per-phase instruction counts of THIS round's kernels (tools/fast_insts.sh, tools/desc_insts.sh on the cuts build)
DYN_FILES = {"fast_cell_kernel<48, true>": ROUND + "_fast_insts.txt", "describe_kernel": ROUND + "_desc_insts.txt"}


def measured_cycles():
    cyc = {}
    for name in ("r02_valu_peak.json", "r03_valu_peak.json", "r05_valu_peak_sdwa.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            for op, v in json.load(open(p))["ops"].items():
                c = v["8"]["cycles_per_wave_inst_per_simd"]
                if op == "v_cndmask_b32" and c > 12:  # r02's stream selected on a VCC nothing ever wrote: 23 cycles at every occupancy, not an issue cost
                    continue                          # (r03 re-measures it with an SGPR-pair mask)
                cyc[re.sub(r"_row_shr$", "", re.sub(r"_sgpr$", "", op))] = c
    return cyc


def asm_of(unit, cuts):
    out = os.path.join(tempfile.gettempdir(), "isa_mix_%s%s.s" % (unit, "_cuts" if cuts else ""))
    cmd = ["hipcc"] + FLAGS + (["-DORBFE_PROFILE_CUTS"] if cuts else []) + ["--offload-device-only", "-S", "-o", out, os.path.join(CSRC, unit)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def kernel_lines(path, frag):
    cur, lines = False, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = frag in m.group(1)
            continue
        if cur:
            lines.append(line)
            if line.strip().startswith("s_endpgm"):
                break
    return lines


def base(op):
    """encoding suffixes dropped, except that a DPP / SDWA operand modifier makes it another (slower) instruction"""
    op = re.sub(r"_(e32|e64)$", "", op)
    op = re.sub(r"_e64_dpp$", "_dpp", op)
    return op


def histogram(lines):
    """[(phase_end_marker, Counter of VALU opcodes, Counter of scalar opcodes, lds, vmem)] in program order"""
    phases, seen = [], set()
    v, s, lds, vm = collections.Counter(), collections.Counter(), 0, 0
    for line in lines:
        m = re.match(r"\s*; ORBFE_PHASE_END (\d+)", line)
        if m and int(m.group(1)) not in seen:
            seen.add(int(m.group(1)))
            phases.append((int(m.group(1)), v, s, lds, vm))
            v, s, lds, vm = collections.Counter(), collections.Counter(), 0, 0
            continue
        m = re.match(r"^\s+([a-z]+_\w+)", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            v[base(op)] += 1
        elif op.startswith("s_"):
            s[op] += 1
        elif op.startswith("ds_"):
            lds += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            vm += 1
    phases.append((0, v, s, lds, vm))
    return phases


def salu_families(c):
    fam = collections.Counter()
    for op, n in c.items():
        if op.startswith("s_waitcnt") or op == "s_nop":
            fam["wait / nop (s_waitcnt, s_nop)"] += n
        elif op.endswith("_b64"):
            fam["64-bit mask logic and moves (exec, ballot results, s_and_saveexec)"] += n
        elif op.startswith(("s_cbranch", "s_branch", "s_cmp", "s_bitcmp")):
            fam["compare + branch"] += n
        elif op.startswith(("s_load", "s_buffer_load")):
            fam["scalar loads (kernel arguments, tables)"] += n
        elif op.startswith(("s_bcnt", "s_ff1", "s_flbit", "s_lshl", "s_lshr", "s_bfe", "s_bfm", "s_ashr")):
            fam["bit counts / shifts on masks and indices"] += n
        else:
            fam["scalar arithmetic / moves (addresses, loop counters)"] += n
    return dict(fam.most_common())


def family(op):
    """compare opcodes differ only in the predicate: one measured member stands for its operand type"""
    if op.endswith("_dpp"):   # every measured DPP form issues at the half rate whatever its base opcode
        return "v_add_u32_dpp" if op != "v_mov_b32_dpp" else op
    if op.endswith("_sdwa"):
        return "v_add_u32_dpp"  # sub-dword operand selection: priced like DPP (not measured separately)
    m = re.match(r"v_cmpx?_\w+?_(i32|u32|i16|u16|f32|f16|i64|u64)$", op)
    if m:
        return {"i32": "v_cmp_lt_i32", "u32": "v_cmp_lt_i32", "i16": "v_cmp_gt_i16", "u16": "v_cmp_gt_i16", "f32": "v_cmp_ge_f32"}.get(m.group(1), op)
    return {"v_min_u16": "v_max_u16", "v_min3_u16": "v_max3_u16", "v_lshrrev_b16": "v_pk_lshrrev_b16", "v_sub_u16": "v_max_u16", "v_add_u16": "v_max_u16",
            "v_min_f32": "v_max_f32", "v_max_u32": "v_min_u32", "v_min_i16": "v_max_u16", "v_max_i16": "v_max_u16", "v_pk_min_u16": "v_pk_max_i16", "v_pk_max_u16": "v_pk_max_i16",
            "v_pk_min_i16": "v_pk_max_i16", "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32"}.get(op, op)


def price(v0, cyc):
    v = collections.Counter()
    for op, n in v0.items():
        v[family(op)] += n
    tot = sum(v.values())
    known = {op: n for op, n in v.items() if op in cyc}
    unknown = {op: n for op, n in v.items() if op not in cyc}
    c = sum(n * cyc[op] for op, n in known.items()) + sum(unknown.values()) * 4.13
    full = sum(n for op, n in known.items() if cyc[op] < 3.0)
    return {"valu_static": tot, "full_rate_share": full / tot if tot else 0.0, "mean_cycles_per_valu": c / tot if tot else 0.0,
            "unmeasured_share": sum(unknown.values()) / tot if tot else 0.0, "top_opcodes": dict(v.most_common(12)), "unmeasured": dict(collections.Counter(unknown).most_common(8))}


def dynamic_counts(path):
    """cumulative {cut: {counter: per-wave value}} from tools/*_insts.sh output"""
    if not os.path.exists(path):
        return None
    out = {}
    for line in open(path):
        m = re.match(r"dbg (\d+) per wave: (\{.*\})", line.strip())
        if m:
            out[int(m.group(1))] = eval(m.group(2), {"__builtins__": {}})
    return out or None


def main():
    cyc = measured_cycles()
    half = 4.13
    res = {"build_id": sys.argv[1] if len(sys.argv) > 1 else None,  # orbfe_build_id() of the sources disassembled here: bench.py replays the mix only for that build
           "method": " ".join(__doc__.split("\n\n")[1].split()), "opcodes_measured": len(cyc), "kernels": {}}
    for unit, kernels in UNITS.items():
        path = asm_of(unit, unit in CUTS)
        for frag, name in kernels:
            lines = kernel_lines(path, frag)
            if not lines:
                continue
            ph = histogram(lines)
            allv, alls = collections.Counter(), collections.Counter()
            for _, v, s, _, _ in ph:
                allv.update(v); alls.update(s)
            entry = {"static": price(allv, cyc), "salu_static": sum(alls.values()), "salu_families_static": salu_families(alls),
                     "lds_static": sum(p[3] for p in ph), "vmem_static": sum(p[4] for p in ph)}
            dyn = dynamic_counts(os.path.join(ROOT, "profiles", DYN_FILES.get(name, "none")))
            if dyn and len(ph) > 1:
                prev = {"SQ_INSTS_VALU": 0.0, "SQ_INSTS_SALU": 0.0, "SQ_INSTS_LDS": 0.0}
                phases, t_cyc, t_n, t_full = [], 0.0, 0.0, 0.0
                pend_v, pend_s = collections.Counter(), collections.Counter()
                for cut, v, s, lds, vm in ph:
                    pend_v.update(v); pend_s.update(s)
                    if cut not in dyn:  # a marker without a measured cut: its code joins the next phase
                        continue
                    d = dyn[cut]
                    n = d["SQ_INSTS_VALU"] - prev["SQ_INSTS_VALU"]
                    pr = price(pend_v, cyc)
                    phases.append({"phase": PHASE_NAMES.get(name, {}).get(cut, str(cut)), "valu_per_wave": round(n, 1), "salu_per_wave": round(d["SQ_INSTS_SALU"] - prev["SQ_INSTS_SALU"], 1),
                                   "lds_per_wave": round(d["SQ_INSTS_LDS"] - prev["SQ_INSTS_LDS"], 1), "full_rate_share": round(pr["full_rate_share"], 3),
                                   "mean_cycles_per_valu": round(pr["mean_cycles_per_valu"], 3), "valu_static": pr["valu_static"], "salu_static": sum(pend_s.values()),
                                   "salu_families_static": salu_families(pend_s), "top_opcodes": pr["top_opcodes"]})
                    t_cyc += n * pr["mean_cycles_per_valu"]; t_n += n; t_full += n * pr["full_rate_share"]
                    prev = d
                    pend_v, pend_s = collections.Counter(), collections.Counter()
                entry["phases"] = phases
                entry["full_rate_share"] = t_full / t_n
                entry["mean_cycles_per_valu"] = t_cyc / t_n
                entry["weighting"] = "phases weighted by measured SQ_INSTS_VALU per wave (cuts build)"
            else:
                entry["full_rate_share"] = entry["static"]["full_rate_share"]
                entry["mean_cycles_per_valu"] = entry["static"]["mean_cycles_per_valu"]
                entry["weighting"] = "static mix of the whole kernel"
            entry["issue_cost_vs_all_half_rate"] = entry["mean_cycles_per_valu"] / half
            res["kernels"][name] = entry
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
