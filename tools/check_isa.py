#!/usr/bin/env python3
"""Guard of the build flags: look at the ISA that is actually INSIDE the linked library.

csrc/Makefile builds the hot kernels with compiler internals (-disable-machine-licm, the load-store-opt subtarget feature switched
off through -Xclang, the IR load/store vectorizer off).  A ROCm update can silently drop one of them; the symptoms are known --
`ds_read2_b64` / `ds_write2_b64` formed out of adjacent 8-byte LDS accesses (8 LDS cycles instead of 2 + 2), more registers than
a SIMD can hold three times, an LDS block that no longer fits a CU -- so the build checks for the symptoms instead of trusting
the flags.  The library's gfx950 code objects are pulled out of the .so (llvm-objdump --offloading), their kernel metadata read
(llvm-readelf --notes) and their text disassembled; the limits below fail the build (exit 1).

    python tools/check_isa.py odr-audioenc_amd/libtoolame_dab_hip.so [--json build/isa/summary.json] [--no-fail]

`make` in odr-audioenc_amd/csrc runs this after the link; tests/test_abi_symbols.py::test_isa_guard runs it on the CPU box too.
"""
import json
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")

# kernel-name pattern -> limits.  `pairs2` = ds_read2_b64 + ds_write2_b64 (+ st64 forms) in the kernel's text.
# The persistent encode kernels run 12 waves per CU: 3 per SIMD needs <= 168 VGPRs (512 / 3, granule 8), no vector spills, and the
# one workgroup's LDS must fit the CU's 160 KB.  No 2-address LDS forms at all in these kernels: with the three flags of csrc/Makefile
# (SI load/store optimizer off, IR load/store vectorizer off, SLP vectorizer off) the count is 0; with the SLP vectorizer alone left on
# it was 140 of the 3.3 k 8-byte reads of tl_frame_kernel<1> (4 %, - 1 %), with the compiler merging freely three quarters of them
# (round 4).  `pairs2_frac` = 2-address forms / all 8-byte-or-wider LDS accesses.
LIMITS = [
    (re.compile(r"tl_frame_kernel"), dict(vgpr=168, lds=163840, vgpr_spill=0, pairs2_frac=0.01)),
    (re.compile(r"tl_main_kernel"), dict(vgpr=168, lds=163840, vgpr_spill=0, pairs2_frac=0.0)),
    (re.compile(r"tl_psy2_kernel"), dict(vgpr=168, lds=163840, vgpr_spill=0)),       # keeps the vectorizer (csrc/Makefile): b128 forms expected
]


def run(*a, cwd=None):
    return subprocess.run([str(x) for x in a], cwd=cwd, check=True, capture_output=True, text=True).stdout


def extract(so: Path, work: Path):
    local = work / so.name
    shutil.copy(so, local)
    run(LLVM / "llvm-objdump", "--offloading", local.name, cwd=work)
    return sorted(p for p in work.iterdir() if "amdgcn" in p.name)


def kernels_of(co: Path):
    """name -> metadata of one code object"""
    notes = run(LLVM / "llvm-readelf", "--notes", co)
    out, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- .") and k in ("agpr_count", "args") and cur.get("name"):
            out[cur["name"]] = cur
            cur = {}
        cur[k] = v
        if k == "wavefront_size" and cur.get("name"):
            out[cur["name"]] = cur
            cur = {}
    if cur.get("name"):
        out[cur["name"]] = cur
    return out


def text_counts(co: Path):
    """kernel -> {instruction mnemonic: count} for the LDS forms the guard cares about, plus the total"""
    dis = run(LLVM / "llvm-objdump", "-d", co)
    counts, k = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            k = m.group(1)
            counts[k] = {"insts": 0}
            continue
        if k is None:
            continue
        t = line.split()
        if not t:
            continue
        op = t[0]
        counts[k]["insts"] += 1
        if op.startswith("ds_read2") or op.startswith("ds_write2") or op in ("ds_read_b128", "ds_write_b128", "ds_read_b64", "ds_write_b64"):
            counts[k][op] = counts[k].get(op, 0) + 1
    return counts


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = Path(args[0] if args else Path(__file__).resolve().parent.parent / "odr-audioenc_amd" / "libtoolame_dab_hip.so")
    js = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if "--json" in sys.argv:
        args = [a for a in args if a != js]
        so = Path(args[0]) if args else so
    summary, bad = {}, []
    with tempfile.TemporaryDirectory() as td:
        for co in extract(so, Path(td)):
            meta, txt = kernels_of(co), text_counts(co)
            for name, m in meta.items():
                c = txt.get(name, {})
                pairs2 = sum(v for k, v in c.items() if k.startswith("ds_read2") or k.startswith("ds_write2"))
                rec = dict(vgpr=int(m.get("vgpr_count", 0)), sgpr=int(m.get("sgpr_count", 0)), vgpr_spill=int(m.get("vgpr_spill_count", 0)),
                           sgpr_spill=int(m.get("sgpr_spill_count", 0)), lds=int(m.get("group_segment_fixed_size", 0)),
                           scratch=int(m.get("private_segment_fixed_size", 0)), insts=c.get("insts", 0), pairs2=pairs2,
                           b128=c.get("ds_read_b128", 0) + c.get("ds_write_b128", 0), b64=c.get("ds_read_b64", 0) + c.get("ds_write_b64", 0))
                wide = rec["pairs2"] + rec["b128"] + rec["b64"]
                rec["pairs2_frac"] = round(rec["pairs2"] / wide, 4) if wide else 0.0
                summary[name] = rec
                for pat, lim in LIMITS:
                    if not pat.search(name):
                        continue
                    for key, mx in lim.items():
                        if mx is not None and rec[key] > mx:
                            bad.append(f"{name}: {key} = {rec[key]} > {mx}")
    w = max(len(k) for k in summary) if summary else 10
    print(f"{'kernel':<{w}}  vgpr sgpr vsp ssp     lds scratch   insts  ds2 b128")
    for k, r in summary.items():
        print(f"{k:<{w}}  {r['vgpr']:4d} {r['sgpr']:4d} {r['vgpr_spill']:3d} {r['sgpr_spill']:3d} {r['lds']:7d} {r['scratch']:7d} {r['insts']:7d} {r['pairs2']:4d} {r['b128']:4d}")
    if js:
        Path(js).parent.mkdir(parents=True, exist_ok=True)
        Path(js).write_text(json.dumps(summary, indent=1, sort_keys=True))
    if not summary:
        bad.append("no gfx950 kernels found in " + str(so))
    for b in bad:
        print("ISA GUARD:", b, file=sys.stderr)
    if bad and "--no-fail" not in sys.argv:
        sys.exit(1)


if __name__ == "__main__":
    main()
