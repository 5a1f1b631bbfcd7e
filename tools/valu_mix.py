#!/usr/bin/env python3
"""Opcode-class mix of the vector instructions in the LOOP bodies of a kernel (static count over every basic block that lies
inside a backward branch's range), priced with the measured per-class issue costs of profiles/r03_valu_rate.txt (8 wavefronts
per SIMD column).  CPU only: compiles the .hip sources to ISA.  Output: profiles/r05_valu_mix.json, read by bench.py for the
"measured basis" of valu_issue_frac.  usage: python tools/valu_mix.py"""
import json
import pathlib
import re
import subprocess
import tempfile
from collections import Counter

ROOT = pathlib.Path(__file__).resolve().parent.parent
CSRC = ROOT / "ml-pgdvs_amd" / "csrc"
# ns per wave64 instruction per SIMD at 8 wavefronts per SIMD (profiles/r03_valu_rate.txt)
COST = {"minmax": 1.80, "fma": 1.70, "pk_fma": 1.88, "cmp_cndmask": 0.69, "trans": 3.42, "simple": 1.10}
KERNELS = {"raster_tile": ("raster.hip", r"raster_tile_kernel"), "grid_query_tpq": ("knn_grid.hip", r"grid_query_tpq_kernelILi51E"),
           "agg_push0": ("static_agg.hip", r"agg_push_kernelILi1024E"), "agg_step": ("static_agg.hip", r"agg_step_kernel")}


def classify(op):
    if re.match(r"v_(min|max|med3|min3|max3)", op):
        return "minmax"
    if op.startswith("v_pk_fma") or op.startswith("v_pk_mul") or op.startswith("v_pk_add"):
        return "pk_fma"
    if re.match(r"v_(fma|fmac|mad|mac)", op):
        return "fma"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "cmp_cndmask"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
        return "trans"
    return "simple"


def loop_mix(asm, name_re):
    m = re.search(r"^(_ZN\S*" + name_re + r"\S*):\s*(;.*)?$", asm, re.M)
    assert m, name_re
    body = asm[m.end():asm.index(".Lfunc_end", m.end())]
    lines = body.splitlines()
    label_at = {}
    for i, ln in enumerate(lines):
        t = ln.strip()
        if re.match(r"^\.LBB\S+:", t):
            label_at[t.split(":")[0]] = i
    in_loop = [False] * len(lines)
    for i, ln in enumerate(lines):
        t = ln.strip().split()
        if t and t[0].startswith(("s_cbranch", "s_branch")) and t[-1] in label_at and label_at[t[-1]] <= i:
            for k in range(label_at[t[-1]], i + 1):
                in_loop[k] = True
    c = Counter()
    for i, ln in enumerate(lines):
        t = ln.strip()
        if in_loop[i] and t.startswith("v_") and not t.startswith("v_mfma"):
            c[classify(t.split()[0])] += 1
    return c


def main():
    out = {"cost_ns_per_wave_instruction_per_simd": COST, "cost_source": "profiles/r03_valu_rate.txt, 8 wavefronts per SIMD",
           "method": "static count of the vector instructions inside loop bodies (ranges of backward branches) of the compiled kernel",
           "kernels": {}}
    with tempfile.TemporaryDirectory() as td:
        asm_of = {}
        for k, (src, name_re) in KERNELS.items():
            if src not in asm_of:
                s = pathlib.Path(td) / (src + ".s")
                subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                                "-munsafe-fp-atomics", "-x", "hip", "-S", "--cuda-device-only", str(CSRC / src), "-o", str(s)],
                               check=True, stderr=subprocess.DEVNULL)
                asm_of[src] = s.read_text()
            c = loop_mix(asm_of[src], name_re)
            n = sum(c.values())
            mix = {kk: round(v / n, 4) for kk, v in sorted(c.items())}
            ns = sum(COST[kk] * v for kk, v in c.items()) / n
            out["kernels"][k] = {"loop_vector_instructions": n, "mix": mix, "ns_per_wave_instruction": round(ns, 4),
                                 "nominal_ns_per_wave_instruction": round(2 / 2.4, 4)}
            print(k, n, mix, round(ns, 3))
    (ROOT / "profiles" / "r05_valu_mix.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
