"""The C-ABI library loads without a GPU and exports every function include/toolame_batch.h declares (the nine reference
symbols of libtoolame-dab.sym among them); creating an encoder without a GPU fails loudly -- there is no CPU fallback."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF_SYMS = ["toolame_init", "toolame_finish", "toolame_enable_byteswap", "toolame_set_channel_mode", "toolame_set_psy_model",
            "toolame_set_bitrate", "toolame_set_samplerate", "toolame_set_pad", "toolame_encode_frame"]   # libtoolame-dab.sym


def _declared():
    src = (ROOT / "include" / "toolame_batch.h").read_text()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:toolame|tlb)_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_reference_abi():
    names = _declared()
    for s in REF_SYMS:
        assert s in names


def test_library_exports_every_declared_symbol():
    import odr_audioenc_amd as M
    if not M.LIB_PATH.exists():
        M.build()
    lib = C.CDLL(str(M.LIB_PATH))
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    import torch
    import odr_audioenc_amd as M
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(M.ToolameError) as e:
        M.Batch([M.StreamConfig()])
    assert e.value.code == 16


def test_product_does_not_reference_oracle():
    """nothing under the product package includes, links or loads oracle/ or tests/emu"""
    for p in (ROOT / "odr-audioenc_amd").rglob("*"):
        if p.suffix in (".h", ".hip", ".cpp", ".py", ".inc") or p.name == "Makefile":
            t = p.read_text(errors="ignore")
            assert "oracle/" not in t.replace("oracle/mp2_oracle.c:psy3_run", "") and "mp2_emu" not in t and "libmp2oracle" not in t, p


def _exported():
    import subprocess
    import odr_audioenc_amd as M
    if not M.LIB_PATH.exists():
        M.build()
    out = subprocess.run(["nm", "-D", "--defined-only", str(M.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    return [ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in "TtDdBbRr"]


def test_export_list_is_the_reference_sym_file_plus_tlb():
    """The WHOLE dynamic symbol table (`nm -D --defined-only`, every symbol type) is the nine names of libtoolame-dab.sym
    plus the tlb_* functions include/toolame_batch.h declares -- no kernel stubs, no mangled C++ names, no shim state
    (csrc/exports.map is the linker's version script)."""
    import subprocess
    import odr_audioenc_amd as M
    if not M.LIB_PATH.exists():
        M.build()
    out = subprocess.run(["nm", "-D", "--defined-only", str(M.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    names = sorted(ln.split()[-1] for ln in out.splitlines() if ln.split())
    assert names == sorted(list(REF_SYMS) + [n for n in _declared() if n.startswith("tlb_")]), \
        sorted(set(names) ^ set(list(REF_SYMS) + [n for n in _declared() if n.startswith("tlb_")]))


REF_HEADER = Path("/root/reference/libtoolame-dab/toolame.h")


@pytest.mark.skipif(not REF_HEADER.exists(), reason="the reference tree is only present in the build container")
def test_one_translation_unit_with_the_reference_header(tmp_path):
    """include/toolame_batch.h and the reference's own toolame.h in ONE C translation unit: a conflicting prototype fails to
    compile.  The TU takes the address of all nine functions with the reference's types and links against the product .so."""
    import subprocess
    import odr_audioenc_amd as M
    src = tmp_path / "both.c"
    src.write_text('#include "%s"\n#include "%s"\n' % (REF_HEADER, ROOT / "include" / "toolame_batch.h") + r"""
#include <stdio.h>
int main(void)
{
    int (*a)(void) = toolame_init; int (*b)(unsigned char *, size_t) = toolame_finish; int (*c)(void) = toolame_enable_byteswap;
    int (*d)(const char) = toolame_set_channel_mode; int (*e)(int) = toolame_set_psy_model; int (*f)(int) = toolame_set_bitrate;
    int (*g)(long) = toolame_set_samplerate; int (*h)(int) = toolame_set_pad;
    int (*i)(short [2][1152], unsigned char *, size_t, unsigned char *, size_t) = toolame_encode_frame;
    /* setter semantics that need no GPU: same call order as src/odr-audioenc.cpp:687-722 */
    if (a() || g(48000) || e(1) || d('j')) return 2;
    if (f(100) == 0) return 3;                 /* 100 kbps is not a Layer II rate: refused at the setter (toolame.c:212-237) */
    if (f(128) != 0 || h(58) != 0) return 4;
    if (g(24000) || f(384) == 0) return 5;     /* 384 kbps is not an MPEG-2 LSF rate */
    if (f(64) != 0) return 6;
    if (e(4) == 0 || d('x') == 0 || g(12345) == 0 || h(-1) == 0) return 7;
    printf("%p%p%p%p%p%p%p%p%p\n", (void *)a, (void *)b, (void *)c, (void *)d, (void *)e, (void *)f, (void *)g, (void *)h, (void *)i);
    return 0;
}
""")
    exe = tmp_path / "both"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src), str(M.LIB_PATH), "-Wl,-rpath," + str(M.LIB_PATH.parent),
                        "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)


def test_reference_send_schedule_against_the_reference_s_own_burst_lengths():
    """tlb_reference_send_schedule (host arithmetic, no GPU): how many 3*bitrate-byte units odr-audioenc's send loop
    (src/odr-audioenc.cpp:1208-1225: `while (toolame_buffer.size() > frame_len)`) sends during each call, derived from the
    configuration alone -- against the same loop run over the per-call return lengths the REAL reference produced (the `lens` of every
    golden case without X-PAD; tests/golden/make_golden.py)."""
    import numpy as np
    import odr_audioenc_amd as M
    from conftest import golden_cases
    if not M.LIB_PATH.exists():
        M.build()
    lib = C.CDLL(str(M.LIB_PATH))

    class Cfg(C.Structure):
        _fields_ = [("samplerate", C.c_long), ("mode", C.c_char), ("bitrate", C.c_int), ("psy_model", C.c_int), ("pad_len", C.c_int)]
    lib.tlb_reference_send_schedule.argtypes = [C.POINTER(Cfg), C.c_int, C.c_void_p]
    seen = 0
    for p in golden_cases():
        g = np.load(p)
        fs, mode, kbps, psy, kind, seed, pad_len, nframes = (int(v) for v in g["cfg"])
        if pad_len:
            continue
        lens = [int(x) for x in g["lens"]][:nframes]                 # per toolame_encode_frame call (the last entry is toolame_finish)
        unit, held, want = 3 * kbps, 0, []
        for n in lens:
            held += n
            k = 0
            while held > unit:
                held -= unit; k += 1
            want.append(k)
        got = np.zeros(nframes, dtype=np.int32)
        rest = lib.tlb_reference_send_schedule(C.byref(Cfg(fs, bytes([mode]), kbps, min(psy, 3), 0)), nframes, got.ctypes.data)
        assert rest == held and list(got) == want, p.stem
        seen += 1
    assert seen > 80
    # and over a long run at the metric's configuration: bursts of 9-10 units every ten calls, one unit always held
    got = np.zeros(2000, dtype=np.int32)
    rest = lib.tlb_reference_send_schedule(C.byref(Cfg(48000, b"s", 128, 1, 0)), 2000, got.ctypes.data)
    assert rest > 0 and set(got[got > 0]) <= {9, 10} and 200 <= int((got > 0).sum()) <= 210 and int(got.sum()) * 384 + rest <= 2000 * 384


def test_selfcheck_libm_says_this_host_is_the_pinned_one():
    """tlb_selfcheck_libm (host-only): the restated glibc routines against THIS host's libm; the build container and the GPU box run
    Ubuntu glibc 2.35 on FMA-capable CPUs, where the answer is 0 differing results (a deployment elsewhere calls it to find out)."""
    import odr_audioenc_amd as M
    if not M.LIB_PATH.exists():
        M.build()
    lib = C.CDLL(str(M.LIB_PATH))
    lib.tlb_selfcheck_libm.restype = C.c_long
    lib.tlb_selfcheck_libm.argtypes = [C.c_long]
    assert lib.tlb_selfcheck_libm(200000) == 0


def test_isa_guard_on_the_linked_library():
    """tools/check_isa.py (run by csrc/Makefile after every link) on the library as it ships: the persistent encode kernels fit three
    waves per SIMD (<= 168 VGPRs, no vector spills), their workgroup fits a CU's LDS, and the compiler has not merged adjacent 8-byte
    LDS accesses into ds_read2_b64 / ds_write2_b64 -- the symptoms of the build flags being dropped."""
    import json
    import subprocess
    import sys
    import odr_audioenc_amd as M
    if not M.LIB_PATH.exists():
        M.build()
    js = ROOT / "build" / "isa" / "test_summary.json"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_isa.py"), str(M.LIB_PATH), "--json", str(js)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    summary = json.loads(js.read_text())
    fk = [v for k, v in summary.items() if "tl_frame_kernel" in k]
    assert len(fk) == 6 and all(v["vgpr"] <= 168 and v["vgpr_spill"] == 0 and v["lds"] <= 163840 and v["pairs2_frac"] <= 0.01 for v in fk)
    # the version string names the toolchain the checked code objects came from: HIP major.minor.patch (this image: 7.x) and the clang version
    import re
    v = M.load_library().tlb_version().decode()
    m = re.search(r"built with HIP (\d+)\.(\d+)\.(\d+), clang (\d+)\.", v)
    assert m and int(m.group(1)) >= 7 and "ISA guard passed" in v, v
