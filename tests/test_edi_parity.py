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
    _check(name, *E.emu_af(frames, levels, fb, st, unit_bytes=E.case_unit_bytes(name)))


@pytest.mark.parametrize("name", [c[0] for c in E.CASES])
def test_reference_classes_vs_golden(name):
    if E.ref_lib() is None:
        pytest.skip("oracle/_ref/libedi_ref.so not built (no /root/reference here)")
    frames, levels, fb, st = E.case_inputs(name)
    _check(name, *E.ref_af(frames, levels, fb, st, unit_bytes=E.case_unit_bytes(name)))


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


@pytest.mark.parametrize("name", [c[0] for c in E.PFT_CASES])
def test_pft_fragments_decode_with_the_reference_rs_decoder(name):
    """Receiver-side pin of the fragment layout (contrib/edioutput/PFT.cpp itself is not buildable here): the emulated device
    fragments, with `fec` of them dropped, go through a TS 102 821 receiver whose Reed-Solomon decoder is the reference's own
    contrib/fec/decode_rs_char.c, and the AF packet (LEN and CRC checked) comes back bit-exactly."""
    if E.pft_ref_lib() is None:
        pytest.skip("oracle/_ref/libpft_ref.so not built (no /root/reference here)")
    af, af_len, pseq, kw = E.pft_case_inputs(name)
    frags, flen, nfrag, _ = E.emu_pft(af, af_len, pseq, **kw)
    E.check_reassembly(af, af_len, frags, flen, nfrag, kw["fec"])


def test_pft_absent_and_oversize_packets():
    """an AF length of 0 (or below) is an ABSENT packet -- the surplus unit slots of a stream with fewer units per frame than
    its batch -- and takes no Pseq; a length beyond the slot is dropped but counted; neither makes fragments (no division by
    a zero chunk count, no copy past the staging buffer) and the packets around them only shift their sequence numbers"""
    af, af_len, pseq, kw = E.pft_case_inputs("fec2")
    good = E.emu_pft(af, af_len, pseq, **kw)
    bad_len = af_len.copy()
    bad_len[3, 0], bad_len[5, 1], bad_len[9, 0] = 0, af.shape[2] + 4, -7
    frags, flen, nfrag, ps = E.emu_pft(af, bad_len, pseq, **kw)
    assert nfrag[3, 0] == 0 and nfrag[5, 1] == 0 and nfrag[9, 0] == 0
    assert ps[0] == np.uint16(good[3][0] - 2) and ps[1] == good[3][1]          # stream 0 lost two sequence numbers, stream 1 none
    n = af.shape[0]
    for s, absent in ((0, [3, 9]), (1, [])):
        for f in range(n):
            if nfrag[f, s] == 0:
                continue
            shift = sum(1 for a in absent if a < f)
            assert nfrag[f, s] == good[2][f, s]
            want = good[0][f, s].copy()
            got = frags[f, s]
            for i in range(nfrag[f, s]):                               # same fragments, Pseq lower by the absent packets before
                ps_w = int.from_bytes(want[i, 2:4].tobytes(), "big")
                assert int.from_bytes(got[i, 2:4].tobytes(), "big") == (ps_w - shift) % 65536
            hdr = 16
            assert (got[:, hdr:] == want[:, hdr:]).all()
    for fec in (0, 2):
        k2 = dict(kw, fec=fec)
        _, _, n2, p2 = E.emu_pft(af[:4], np.zeros((4, af.shape[1]), dtype=np.int32), pseq, **k2)
        assert (n2 == 0).all() and (p2 == pseq).all()


def test_lsf_units_feed_the_pft_layer():
    """the mixed 48 k / 24 k / 16 k case end to end on the emulated device code: AF packets in slot order (absent slots of
    the streams with fewer units) -> PFT fragments; every present packet reassembles through the reference's RS decoder and
    each stream's Pseq advanced by its own number of units"""
    if E.pft_ref_lib() is None:
        pytest.skip("oracle/_ref/libpft_ref.so not built (no /root/reference here)")
    frames, levels, fb, st = E.case_inputs("lsf_mixed")
    frames, levels = frames[:12], levels[:12]
    ub = E.case_unit_bytes("lsf_mixed")
    pkts, plen, _ = E.emu_af(frames, levels, fb, st, unit_bytes=ub)
    assert pkts.shape[0] == 12 * 3 and (plen[:, 0].reshape(12, 3)[:, 1:] == 0).all() and (plen[:, 3] > 0).all()
    pseq = np.array([65530, 7, 100, 0], dtype=np.uint16)
    frags, flen, nfrag, ps = E.emu_pft(pkts, plen, pseq, fec=2)
    assert list((ps - pseq).astype(np.uint16)) == [12, 24, 24, 36]
    assert ((nfrag > 0) == (plen > 0)).all()
    E.check_reassembly(pkts, plen, frags, flen, nfrag, 2, present_only=True)
