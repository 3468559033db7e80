#!/usr/bin/env python3
"""Static instruction mix of a kernel of odr-audioenc_amd/toolame_hip.gfx950.s (`make -C odr-audioenc_amd/csrc asm`), by class.

  tools/isa_mix.py [kernel-substring] [--loops N] [--stages]

* whole-kernel table: instructions per class {fp64 arithmetic, conversions, moves, selects, compares, shifts / address,
  integer + bit operations, lane reads / DPP, LDS, VMEM, SALU, waits / nops};
* --stages: the same per region between two TL_STAMP sites (s_memtime ... global_store offset: 8 * stamp index), in program
  order -- the stereo path of tl_frame_kernel<1> is the run of regions 15, 8, 24.., 9.., 16.., 23 (model phase), 31, 0..7 (encoder);
* --loops N: the N largest innermost loops (the compiler marks them "Inner Loop Header"), each with its class mix and the
  number of constants it re-materialises per trip (v_mov_b32 of a literal: what `-disable-machine-licm` leaves inside loops).

The DYNAMIC mix per stage comes from counters (tools/class_budget.sh -> profiles/class_budget_r04.txt); this tool says what the
instructions are.  Weights: profiles/instr_rates_r04.txt (tools/instr_rates.hip)."""
import collections
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ASM = ROOT / "odr-audioenc_amd" / "toolame_hip.gfx950.s"

F64 = ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64", "v_max_f64", "v_min_f64", "v_ldexp_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
       "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_fract_f64", "v_floor_f64", "v_trunc_f64", "v_rndne_f64", "v_frexp_mant_f64")


def classify(op):
    if op.startswith(F64): return "f64"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith(("v_mov_b32_dpp", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")) or "_dpp" in op or "_sdwa" in op: return "lane"
    if op.startswith(("v_mov", "v_accvgpr", "v_pk_mov")): return "mov"
    if op.startswith("v_cndmask"): return "select"
    if op.startswith(("v_cmp", "v_cmpx")): return "cmp"
    if op.startswith(("v_lshl", "v_lshr", "v_ashr", "v_lshl_add", "v_lshl_or", "v_mad_u64", "v_mad_i64")): return "shift/addr"
    if op.startswith("v_"): return "int/bit"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait/nop"
    if op.startswith("s_"): return "salu"
    return "?"


CLASSES = ["f64", "cvt", "mov", "select", "cmp", "shift/addr", "int/bit", "lane", "lds", "vmem", "salu", "wait/nop"]
VALU = CLASSES[:8]


def kernel_lines(sub):
    lines = ASM.read_text().splitlines()
    out, on = [], False
    for ln in lines:
        if re.match(r"^_Z\w+:", ln):
            on = sub in ln
            if on: name = ln.split(":")[0]
        elif ln.startswith(".Lfunc_end"):
            if on: return name, out
            on = False
        if on: out.append(ln)
    raise SystemExit(f"no kernel matching {sub!r} in {ASM}")


def instr(ln):
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\b(.*)", ln)
    return (m.group(1), m.group(2)) if m else None


def table(title, rows):
    print(title)
    print(f"  {'':38s}" + "".join(f"{c:>11s}" for c in CLASSES) + f"{'VALU':>8s}{'VALU/f64':>9s}")
    for name, cnt in rows:
        valu = sum(cnt[c] for c in VALU)
        ratio = f"{valu / cnt['f64']:9.2f}" if cnt["f64"] else "        -"
        print(f"  {name:38s}" + "".join(f"{cnt[c]:11d}" for c in CLASSES) + f"{valu:8d}{ratio}")


def main():
    args = sys.argv[1:]
    sub = next((a for a in args if not a.startswith("--")), "tl_frame_kernelILi1E")
    name, lines = kernel_lines(sub)
    total = collections.Counter()
    for ln in lines:
        i = instr(ln)
        if i: total[classify(i[0])] += 1
    table(f"# {name}: static instruction mix (lines of {ASM.name})", [("whole kernel", total)])
    if "--stages" in args:
        # a stamp site: s_memtime, then (under `lane 0 && stamps`) a global_store_dwordx2 with offset 8 * index; regions in program order
        rows, cur, label, pending = [], collections.Counter(), "kernel entry", False
        for ln in lines:
            i = instr(ln)
            if not i: continue
            if i[0] == "s_memtime": pending = True
            if pending and i[0] == "global_store_dwordx2":
                m = re.search(r"offset:(\d+)", i[1])
                idx = int(m.group(1)) // 8 if m else 0
                rows.append((label, cur)); cur = collections.Counter(); label = f"after stamp {idx}"; pending = False
                continue
            cur[classify(i[0])] += 1
        rows.append((label, cur))
        table("# regions between stamp sites, program order (one row per code copy: stereo / mono / dead-head paths follow each other)", [r for r in rows if sum(r[1].values()) > 40])
    if "--loops" in args:
        n = int(args[args.index("--loops") + 1])
        loops, cur, header = [], None, None
        for k, ln in enumerate(lines):
            if "Inner Loop Header" in ln:
                cur = {"line": k, "depth": re.search(r"Depth=(\d+)", ln).group(1), "cnt": collections.Counter(), "lit": 0, "label": header}
                continue
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m: header = m.group(1)
            if cur is None: continue
            i = instr(ln)
            if not i: continue
            cur["cnt"][classify(i[0])] += 1
            if i[0].startswith("v_mov_b32") and re.search(r",\s*(0x[0-9a-f]+|-?\d+)\s*$", i[1]): cur["lit"] += 1
            if i[0].startswith("s_cbranch") and cur["label"] and cur["label"] in i[1]:
                loops.append(cur); cur = None
        loops.sort(key=lambda l: -sum(l["cnt"][c] for c in VALU))
        rows = [(f"{l['label']} @{l['line']} depth {l['depth']} lit-mov {l['lit']}", l["cnt"]) for l in loops[:n]]
        table(f"# the {n} largest innermost loops (one trip each)", rows)


if __name__ == "__main__":
    main()
