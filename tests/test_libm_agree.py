"""csrc/tl_libm.h (the device's log / log10 / exp / pow / sincos / atan2) against THIS machine's libm, bit for bit.

The oracle is the reference linked against the host's glibc, so these functions are part of what "the reference's result"
means (third-party arithmetic).  tools/libm_agree.cpp draws arguments over the encoder's ranges, every
binade and the branch points of each routine; the long run (>= 1e8 arguments per function) is kept in profiles/, the suite
runs 2 M per function.  A libm other than glibc 2.35's dbl-64 routines on an FMA machine fails here -- which is the point:
it would also change the oracle's bytes on degenerate signals."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_tl_libm_equals_host_libm(tmp_path):
    exe = tmp_path / "libm_agree"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-pthread", "-I", str(ROOT / "odr-audioenc_amd" / "csrc"),
                    str(ROOT / "tools" / "libm_agree.cpp"), "-o", str(exe), "-lm"], check=True)
    r = subprocess.run([str(exe), "2", "4", "7"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count(" 0 differ") >= 16, r.stdout


def test_tables_are_this_libms_tables():
    """tl_libm_tables.inc is what tools/extract_libm_tables.py reads out of the libm.so.6 the oracle links against."""
    import hashlib
    import re
    import struct
    inc = (ROOT / "odr-audioenc_amd" / "csrc" / "tl_libm_tables.inc").read_text()
    libm = Path("/lib/x86_64-linux-gnu/libm.so.6").read_bytes()
    if hashlib.sha256(libm).hexdigest() not in inc:
        import pytest
        pytest.skip("another libm build: the agreement test above is the authority")
    found = re.findall(r"libm\.so\.6 \.rodata (0x[0-9a-f]+)\nTLM_TABLE_QUAL uint64_t (\w+)\[(\d+)\]", inc)
    assert len(found) == 10
    for addr, name, n in ((int(a, 16), nm, int(k)) for a, nm, k in found):
        body = inc[inc.index(name + "["):]
        body = body[body.index("{") + 1: body.index("}")]
        words = [int(w, 16) for w in re.findall(r"0x[0-9a-f]{16}", body)]
        assert len(words) == n and tuple(words) == struct.unpack_from(f"<{n}Q", libm, addr), name


def test_log10_of_the_energy_floor_is_exactly_minus_twenty():
    """csrc/mp2_wave.h tl_power_db() folds the reference's `energy < 1E-20 ? -200 + POWERNORM : 10 log10(energy) + POWERNORM`
    (psycho_1.c:243-246, psycho_3.c:152-160) into 10 log10(max(energy, 1E-20)) + POWERNORM.  That is the same function exactly if
    log10 of the double 1E-20 is -20.0 -- in the host's libm (what the reference runs) and in the restatement the device runs."""
    import math
    import sys
    sys.path.insert(0, str(ROOT / "tests"))
    import emulib as E
    assert math.log10(1e-20) == -20.0
    assert E.lib().emu_log10_pn(1e-20) == -20.0
    assert 10 * math.log10(1e-20) + 90.3090 == -200.0 + 90.3090


def test_log_pn_of_exact_powers_of_two_needs_no_special_case():
    """tlm_log_pn (the device's straight-line log) has no `x == 1.0 ? 0.0` line: e_log.c's early return exists for the directed rounding
    modes, to nearest the near-1 polynomial of 1.0 is +0.0 itself.  log10 hands log the mantissa 1.0 for every power of two >= 1 (and 0.5
    doubled -- also 1.0 -- never: negative exponents get the mantissa in [0.5, 1)), so the sign of that zero reaches the result."""
    import math
    import struct
    import sys
    sys.path.insert(0, str(ROOT / "tests"))
    import emulib as E
    L = E.lib()
    for k in list(range(-60, 61)) + [-1022, 1023]:
        x = math.ldexp(1.0, k)
        got, want = L.emu_log10_pn(x), math.log10(x)
        assert struct.pack("<d", got) == struct.pack("<d", want), (k, got, want)
    for x in (1.0, 0.5, 2.0, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, 10.0, 100.0, 1e-20):
        got, want = L.emu_log10_pn(x), math.log10(x)
        assert struct.pack("<d", got) == struct.pack("<d", want), (x, got, want)

