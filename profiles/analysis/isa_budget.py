"""Instruction budget of a kernel's basic blocks from the ISA listing `hipcc --save-temps` leaves behind.

    cd /tmp/rt && hipcc --offload-arch=gfx950 <Makefile flags> -c render.hip --save-temps
    python profiles/analysis/isa_budget.py render-hip-amdgcn-amd-amdhsa-gfx950.s 'k_render_fwd3<true, true, true, true>' [min_instr]

Per basic block (label to label): instruction counts by issue class, and the block's estimated SIMD issue cycles with the
per-class costs measured on this part (profiles/probes/README.md: v_fma_f32-class 2.5 cycles per wave64 instruction,
compare / select / move / integer / convert 4, DPP 6, v_exp / v_rcp / v_log 8, permlane swaps 10, LDS and VMEM issue 4,
scalar 1 -- the scalar unit runs beside the vector one).  Blocks with fewer than `min_instr` instructions are folded into
the total only.  Classes:
  fma2   v_fma / v_fmac / v_mul / v_add / v_sub / v_min / v_max / v_med3 on f32 (incl. the clamp forms)
  alu4   v_cmp*, v_cndmask, v_mov, integer and bit ops, conversions, v_perm, readlane / readfirstlane, mbcnt
  trans  v_exp, v_rcp, v_log, v_sqrt, v_rsq
  dpp    any VALU instruction with a DPP modifier;  swap  v_permlane*_swap
  mfma   matrix-core instructions (issue beside the vector ALU)
  lds    ds_*;  vmem  global_* / buffer_* / flat_*;  salu  s_* except waitcnt / nop / branches;  ctl  s_waitcnt, s_nop, branches, barriers
"""
import re
import subprocess
import sys

COST = {"fma2": 2.5, "alu4": 4.0, "trans": 8.0, "dpp": 6.0, "swap": 10.0, "mfma": 0.0, "lds": 4.0, "vmem": 4.0, "salu": 0.0, "ctl": 0.0}
FMA2 = re.compile(r"^v_(fma|fmac|mul|add|sub|subrev|min|max|med3|mac|mad)_(f32|legacy_f32)")


def classify(line):
    op = line.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_permlane") and "swap" in op:
        return "swap"
    if op.startswith("v_"):
        if "dpp" in op or " row_" in line or "quad_perm" in line or "wave_sh" in line or "row_bcast" in line:
            return "dpp"
        if re.match(r"^v_(exp|rcp|log|sqrt|rsq)_", op):
            return "trans"
        if FMA2.match(op):
            return "fma2"
        return "alu4"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_sleep", "s_setprio")):
            return "ctl"
        return "salu"
    return "ctl"


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2]
    min_instr = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    names = re.findall(r"^(_Z\S+):\s*(?:;.*)?$", text, re.M)
    target = None
    for n in names:
        d = demangle(n)
        if d.replace("void cgs::", "").startswith(want):
            target = n
            break
    if target is None:
        sys.exit(f"no kernel matching {want!r}; have: " + ", ".join(sorted({re.sub(r'\\(.*', '', demangle(n)) for n in names})))
    i = text.index("\n" + target + ":")
    j = text.index(".Lfunc_end", i)
    blocks, cur, label = [], {}, "entry"
    order = []
    for ln in text[i:j].splitlines()[2:]:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) and not re.match(r"^\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", s)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1) + (" " + m.group(2).strip("; ").strip() if m.group(2) else ""), {}
            continue
        if not ln.startswith("\t"):
            continue
        c = classify(s)
        cur[c] = cur.get(c, 0) + 1
    blocks.append((label, cur))
    cols = ["fma2", "alu4", "trans", "dpp", "swap", "mfma", "lds", "vmem", "salu", "ctl"]
    print(f"kernel {demangle(target).split('(')[0]}")
    print(f"{'block':46s} " + " ".join(f"{c:>5s}" for c in cols) + "   vec  est.cycles")
    tot = {c: 0 for c in cols}
    for label, cnt in blocks:
        n = sum(cnt.values())
        for c in cols:
            tot[c] += cnt.get(c, 0)
        if n < min_instr:
            continue
        vec = sum(cnt.get(c, 0) for c in ("fma2", "alu4", "trans", "dpp", "swap"))
        cyc = sum(cnt.get(c, 0) * COST[c] for c in cols)
        print(f"{label[:46]:46s} " + " ".join(f"{cnt.get(c, 0):5d}" for c in cols) + f" {vec:5d} {cyc:9.0f}")
    vec = sum(tot[c] for c in ("fma2", "alu4", "trans", "dpp", "swap"))
    print(f"{'TOTAL (static, whole kernel)':46s} " + " ".join(f"{tot[c]:5d}" for c in cols) + f" {vec:5d} {sum(tot[c] * COST[c] for c in cols):9.0f}")


main()
