"""EDI AF-packet step on the CPU: the device source (csrc/edi_af.h, emulated lane loop) against the golden
vectors made with the reference's own TagItems/TagPacket/AFPacket classes, and -- where oracle/_ref/libedi_ref.so exists --
those classes against the same vectors."""
from pathlib import Path

import numpy as np
import pytest

import edilib as E

G = np.load(Path(__file__).resolve().parent / "golden" / "edi_cases.npz")


def _check(name, pkts, plen, st):
    assert (plen == G[name + "_len"]).all()
    assert (pkts[:16] == G[name + "_head"]).all()
    assert E.digest(pkts, plen) == bytes(G[name + "_sha"]).hex()
    assert st.tobytes() == G[name + "_state"].tobytes()


@pytest.mark.parametrize("name", [c[0] for c in E.CASES])
def test_emulated_device_code_vs_golden(name):
    frames, levels, fb, st = E.case_inputs(name)
    _check(name, *E.emu_af(frames, levels, fb, st))


@pytest.mark.parametrize("name", [c[0] for c in E.CASES])
def test_reference_classes_vs_golden(name):
    if E.ref_lib() is None:
        pytest.skip("oracle/_ref/libedi_ref.so not built (no /root/reference here)")
    frames, levels, fb, st = E.case_inputs(name)
    _check(name, *E.ref_af(frames, levels, fb, st))


def test_packet_structure():
    """independent reading of one packet: AF header, tag names and lengths, CRC-16/CCITT (TS 102 821)"""
    frames, levels, fb, st = E.case_inputs("tist_wrap")
    pkts, plen, _ = E.emu_af(frames, levels, fb, st)
    p = pkts[0, 0, : plen[0, 0]].tobytes()
    assert p[:2] == b"AF" and p[8] == 0x90 and p[9:10] == b"T"
    taglen = int.from_bytes(p[2:6], "big")
    assert len(p) == 10 + taglen + 2 and int.from_bytes(p[6:8], "big") == 65530
    crc = 0xffff
    for byte in p[:-2]:
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xffff if crc & 0x8000 else (crc << 1) & 0xffff
    assert (crc ^ 0xffff) == int.from_bytes(p[-2:], "big")
    pos, names = 10, []
    while pos < len(p) - 2:
        bits = int.from_bytes(p[pos + 4: pos + 8], "big")
        names.append(p[pos: pos + 4])
        pos += 8 + bits // 8
    assert pos == len(p) - 2 and names == [b"*ptr", b"dsti", b"ss\x00\x01", b"ODRa"]
    assert p[10 + 16 + 8 + 2 + 8 + 8 + 3: 10 + 16 + 8 + 2 + 8 + 8 + 3 + 384] == frames[0, 0, :384].tobytes()
