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


PG = np.load(Path(__file__).resolve().parent / "golden" / "edi_pft_cases.npz")


def _check_pft(name, frags, flen, nfrag, ps):
    assert (nfrag == PG[name + "_n"]).all() and (flen == PG[name + "_len"]).all()
    assert (frags[:2] == PG[name + "_head"]).all()
    assert E.pft_digest(frags, flen, nfrag) == bytes(PG[name + "_sha"]).hex()
    assert (ps == PG[name + "_pseq"]).all()


@pytest.mark.parametrize("name", [c[0] for c in E.PFT_CASES])
def test_pft_emulated_device_code_vs_golden(name):
    af, af_len, pseq, kw = E.pft_case_inputs(name)
    _check_pft(name, *E.emu_pft(af, af_len, pseq, **kw))


@pytest.mark.parametrize("name", [c[0] for c in E.PFT_CASES])
def test_pft_reference_rs_vs_golden(name):
    if E.pft_ref_lib() is None:
        pytest.skip("oracle/_ref/libpft_ref.so not built (no /root/reference here)")
    af, af_len, pseq, kw = E.pft_case_inputs(name)
    _check_pft(name, *E.ref_pft(af, af_len, pseq, **kw))


def test_pft_recovers_the_af_packet():
    """size-independent property: de-interleaving the fragments and dropping the parity gives the AF packet back, and every
    255-byte codeword (chunk, zero padding, parity) has zero syndromes at alpha^1..alpha^48 (TS 102 821, 7.2)"""
    af, af_len, pseq, kw = E.pft_case_inputs("fec2")
    frags, flen, nfrag, _ = E.emu_pft(af, af_len, pseq, **kw)
    ex, sr = [0] * 510, 1
    for i in range(255):
        ex[i] = ex[i + 255] = sr
        sr <<= 1
        if sr & 0x100:
            sr ^= 0x11d
    lg = {ex[i]: i for i in range(255)}
    for f, s in ((0, 0), (7, 1), (39, 0)):
        l, n = int(af_len[f, s]), int(nfrag[f, s])
        hdr = 16                                                   # PF header with RSk/RSz, no address part
        fsz = int(flen[f, s, 0]) - hdr
        k, z = int(frags[f, s, 0, 12]), int(frags[f, s, 0, 13])
        c = (l + z) // k
        block = bytearray(n * fsz)
        for i in range(n):
            assert frags[f, s, i, :2].tobytes() == b"PF" and int.from_bytes(frags[f, s, i, 4:7].tobytes(), "big") == i
            for j in range(fsz):
                block[j * n + i] = int(frags[f, s, i, hdr + j])
        got = b"".join(bytes(block[ci * (k + 48): ci * (k + 48) + k]) for ci in range(c))
        assert got[:l] == af[f, s, :l].tobytes() and got[l:] == bytes(z)
        for ci in range(c):
            cw = list(block[ci * (k + 48): ci * (k + 48) + k]) + [0] * (207 - k) + list(block[ci * (k + 48) + k: (ci + 1) * (k + 48)])
            for root in range(1, 49):                              # Horner evaluation of the codeword polynomial at alpha^root
                acc = 0
                for byte in cw:
                    acc = (ex[(lg[acc] + root) % 255] if acc else 0) ^ byte
                assert acc == 0, (f, s, ci, root)
