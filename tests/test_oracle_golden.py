"""The oracle (oracle/mp2_oracle.c) against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  Bit-exact: bytes, burst cadence, every integer tap, and the fp64
taps compared as raw bit patterns.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oraclelib as O
from conftest import golden_cases
from pcmgen import gen_pcm


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("path", golden_cases(), ids=lambda p: p.stem)
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    fs, mode, kbps, psy, kind, seed, pad_len, nframes = (int(v) for v in g["cfg"])
    pcm = gen_pcm(seed, kind, 0, nframes)
    e = O.OracleEncoder(samplerate=fs, mode=chr(mode), kbps=kbps, psy=psy, pad_len=pad_len)
    nch, sbl = e.nch, e.sblimit
    chunks, lens = [], []
    big = {int(f): i for i, f in enumerate(g["big_tap_frames"])} if "big_tap_frames" in g else {}
    for i in range(nframes):
        if "xpad" in g:
            b = e.encode(pcm[i], bytes(g["xpad"][i]), int(g["xpad_len"][i]))
        else:
            b = e.encode(pcm[i])
        chunks.append(b)
        lens.append(len(b))
        t = e.taps()
        assert np.array_equal(t["scalar"][:nch, :, :sbl], g["scalar"][i][:nch, :, :sbl]), ("scalar", i)
        assert np.array_equal(t["scfsi"][:nch, :sbl], g["scfsi"][i][:nch, :sbl]), ("scfsi", i)
        assert np.array_equal(t["bit_alloc"][:nch], g["bit_alloc"][i][:nch]), ("bit_alloc", i)
        assert np.array_equal(_bits(t["max_sc"][:nch]), _bits(g["max_sc"][i][:nch])), ("max_sc", i)
        nsmr = sbl if psy == 1 else 32               # psy 1 writes only sblimit entries
        assert np.array_equal(_bits(t["smr"][:nch, :nsmr]), _bits(g["smr"][i][:nch, :nsmr])), ("smr", i)
        assert (t["mode"], t["mode_ext"]) == (int(g["mode"][i]), int(g["mode_ext"][i])), ("mode", i)
        if chr(mode) == "j":
            assert np.array_equal(t["j_scale"][:, :sbl], g["j_scale"][i][:, :sbl]), ("j_scale", i)
        if i in big:
            assert np.array_equal(_bits(t["sb_sample"][:nch]), _bits(g["sb_sample"][big[i]][:nch])), ("sb_sample", i)
            assert np.array_equal(t["subband"][:nch], g["subband"][big[i]][:nch]), ("subband", i)
    b = e.finish()
    chunks.append(b)
    lens.append(len(b))
    assert lens == list(g["lens"])
    assert b"".join(chunks) == g["data"].tobytes()
    if fs in (44100, 22050):                                 # frames of two lengths: some carry a padding slot
        assert nframes * e.frame_bytes <= len(g["data"]) <= nframes * (e.frame_bytes + 1)
        raw = g["data"].tobytes()
        pads = sum((raw[o + 2] >> 1) & 1 for o in _frame_offsets(raw, e.frame_bytes))
        assert len(g["data"]) == nframes * e.frame_bytes + pads
    else:
        assert len(g["data"]) == nframes * e.frame_bytes       # whole frames, incl. the finish() tail


def _frame_offsets(data, whole):
    """start of every frame of a padded-rate stream: header bit 9 (byte 2, bit 1) says whether the frame is one byte longer"""
    o, out = 0, []
    while o < len(data):
        assert data[o] == 0xff and (data[o + 1] & 0xf0) == 0xf0, o
        out.append(o)
        o += whole + ((data[o + 2] >> 1) & 1)
    return out


def test_tables_match_reference(golden_dir):
    g = np.load(golden_dir / "tables_48k.npz")
    e = O.OracleEncoder(psy=3)
    L = O.lib()
    L.mp2o_get_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    buf = np.zeros(1024, dtype=np.float64)
    for name in ("enwindow", "scalefactor", "dct", "dbtable", "p3_bark", "p3_ath", "p3_cbidx", "p3_subset"):
        n = L.mp2o_get_table(e.h, name.encode(), buf.ctypes.data, 1024)
        ref = np.asarray(g[name], dtype=np.float64).ravel()
        assert n == len(ref), name
        if name in ("p3_bark", "p3_ath"):           # index 0 is never written by the reference
            assert np.array_equal(_bits(buf[1:n]), _bits(ref[1:])), name
        else:
            assert np.array_equal(_bits(buf[:n]), _bits(ref)), name
    assert np.array_equal(_bits(g["multiple"]), _bits(g["scalefactor"]))


@pytest.mark.parametrize("fs", (48000, 44100, 32000, 24000, 22050, 16000))
def test_shared_tables_match_reference_memory(golden_dir, fs):
    """csrc/mp2_tables.inc is shared by the product and the oracle, so end-to-end bytes alone would not notice a table both got
    wrong where no test signal reaches.  Every such table -- psy-1 threshold (line, bark, hear) and critical-band tables and the
    psy-2 absolute threshold of all six sample rates, the allocation tables -- against what the REFERENCE holds in memory after
    its own init code ran (tests/golden/make_golden.py make_rate_tables; critband.h, freqtable.h, absthr.h, encode_new.c:16-100)."""
    g = np.load(golden_dir / "tables_rates.npz")
    e = O.OracleEncoder(samplerate=fs, kbps=128 if fs >= 32000 else 64, psy=1)
    L = O.lib()
    L.mp2o_get_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    buf = np.zeros(1024, dtype=np.float64)
    names = ["p1_line", "p1_bark", "p1_hear", "p1_cbound", "p2_absthr"] + sorted(k for k in g.files if k.startswith("alloc_"))
    for name in names:
        ref = np.asarray(g[name if name.startswith("alloc_") else f"{name}_{fs}"], dtype=np.float64).ravel()
        n = L.mp2o_get_table(e.h, name.encode(), buf.ctypes.data, 1024)
        assert n == len(ref), (name, n, len(ref))
        assert np.array_equal(_bits(buf[:n]), _bits(ref)), name
    assert len(g["p2_absthr_48000"]) == 513 and len(names) == 14


@pytest.mark.parametrize("psy", [2, 4])
@pytest.mark.parametrize("fs", [48000, 44100, 32000, 24000, 22050, 16000])
def test_psy2_psy4_derived_tables_match_reference_memory(golden_dir, fs, psy):
    """The partition map, lines per partition, bark value / normalisation / tone-masking-noise value per partition, spreading function and
    analysis window of psy 2 and psy 4 are built by init code that the oracle AND the product restate (oracle/mp2_oracle_psy2.inc,
    mp2_oracle_psy4.inc; csrc/mp2_host.cpp tl_build_psy2_tables / tl_build_psy4_tables) -- two copies of one text, which end-to-end bytes
    alone would not tell from the reference's.  Both copies against the arrays the REFERENCE's own psycho_2_init / psycho_4_init filled
    (file-scope pointers made visible by oracle/Makefile TABLE_TAPS, read through in tests/golden/make_golden.py), bit for bit."""
    import emulib
    g = np.load(golden_dir / "tables_rates.npz")
    pre = f"p{psy}"
    ref = {k: np.asarray(g[f"{pre}_{k}_{fs}"]) for k in ("partition", "numlines", "cbval", "rnorm", "tmn", "s", "window")}
    # ---- the oracle's copy
    e = O.OracleEncoder(samplerate=fs, kbps=128 if fs >= 32000 else 64, psy=psy)
    L = O.lib()
    L.mp2o_get_table.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    buf = np.zeros(64 * 64, dtype=np.float64)
    names = list(ref) + (["ath", "bark", "minval"] if psy == 4 else ["bmax"])
    for k in names:
        want = np.asarray(g[f"{pre}_{k}" if k in ("bmax", "minval") else f"{pre}_{k}_{fs}"], dtype=np.float64).ravel()
        n = L.mp2o_get_table(e.h, f"{pre}_{k}".encode(), buf.ctypes.data, buf.size)
        assert n == len(want), (k, n, len(want))
        assert np.array_equal(_bits(buf[:n]), _bits(want)), (fs, psy, k)
    # ---- the product's copy (host code of the device path, compiled into the emulation library)
    E = emulib.lib()
    E.emu_psy2_table.argtypes = [C.c_long, C.c_int, C.c_char_p, C.c_void_p, C.c_int]
    def prod(name):
        n = E.emu_psy2_table(fs, psy, name.encode(), buf.ctypes.data, buf.size)
        assert n > 0, name
        return buf[:n].copy()
    assert np.array_equal(_bits(prod("window")), _bits(ref["window"]))
    assert np.array_equal(_bits(prod("s")), _bits(ref["s"].ravel()))
    assert np.array_equal(_bits(prod("tmn")), _bits(ref["tmn"]))
    assert np.array_equal(prod("partition").astype(np.int32), ref["partition"])
    npart = int(ref["partition"][512]) + 1
    assert int(prod("npart")[0]) == npart
    lo, hi = prod("part_lo").astype(int), prod("part_hi").astype(int)
    for j in range(npart):
        lines = np.nonzero(ref["partition"] == j)[0]
        assert (lo[j], hi[j]) == (lines[0], lines[-1] + 1) and hi[j] - lo[j] == ref["numlines"][j], j
    # rnorm[j] * numlines[j], the divisor of psycho_2.c:200-204 (0 where the reference skips the division)
    den = np.where((ref["rnorm"] != 0) & (ref["numlines"] != 0), ref["rnorm"] * ref["numlines"], 0.0)
    assert np.array_equal(_bits(prod("den")), _bits(den))
    if psy == 2:       # bmax[(int)(cbval + 0.5)], psycho_2.c:188-189
        assert np.array_equal(_bits(prod("bmaxk")), _bits(np.asarray(g["p2_bmax"])[(ref["cbval"] + 0.5).astype(int)]))
        assert np.array_equal(_bits(prod("absthr")), _bits(np.asarray(g[f"p2_absthr_{fs}"])))
    else:              # minval[(int)cbval], psycho_4.c:262; the threshold in quiet as energy (ath[], :352-361)
        assert np.array_equal(_bits(prod("bmaxk")), _bits(np.asarray(g["p4_minval"])[ref["cbval"].astype(int)]))
        assert np.array_equal(_bits(prod("absthr")), _bits(np.asarray(g[f"p4_ath_{fs}"])))


def test_burst_cadence_128k():
    """0 bytes for 10 calls, 3708 on the 11th, ... and finish() flushes the rest."""
    pcm = gen_pcm(1, 0, 0, 24)
    data, lens = O.oracle_stream(pcm, psy=1)
    assert lens[:10] == [0] * 10 and lens[10] == 3708 and lens[20] == 3708
    assert sum(lens) == 24 * 384 == len(data)


def test_header_bytes_128k_stereo():
    """Appendix A: FF FC 84 00 for 48 kHz / 128 kbps / mode 's'."""
    data, _ = O.oracle_stream(gen_pcm(2, 0, 0, 3), psy=1)
    for f in range(3):
        assert data[f * 384: f * 384 + 4] == bytes([0xFF, 0xFC, 0x84, 0x00])


def test_illegal_configs_rejected():
    for kw in (dict(samplerate=44000), dict(kbps=100), dict(mode="x"), dict(psy=7), dict(pad_len=-1)):
        with pytest.raises(ValueError):
            O.OracleEncoder(**kw)


def test_silence_psy3_is_defined():
    """The reference segfaults here (psycho_3.c:299, (int)(0/0) index); the oracle defines it."""
    data, _ = O.oracle_stream(gen_pcm(0, 1, 0, 4), psy=3)
    assert len(data) == 4 * 384


def test_baseline_configs0_2000_frames_oracle_equals_live_reference():
    """BASELINE configs[0] / SURVEY 8(d) cfg1: ONE stream, 2000 frames, seed 0, 48 kHz stereo 128 kbps, psycho_1 -- the bit-exact
    gate on the CPU: the restatement against the reference compiled from its own sources (oracle/_ref), bytes and burst lengths.
    Skipped where the reference build is not present (it cannot travel as source)."""
    if not O.REF_SO.exists():
        pytest.skip("oracle/_ref/libtoolame_ref.so not built here")
    from pcmgen import gen_pcm
    pcm = gen_pcm(0, 0, 0, 2000)
    for mode in ("s", "j"):
        ref = O.reference_stream(pcm, mode=mode, psy=1)
        data, lens = O.oracle_stream(pcm, mode=mode, psy=1)
        assert data == ref["data"] and lens == ref["lens"], mode
        assert len(data) == 2000 * 384


def test_spreading_bands_hold_every_nonzero_coefficient_of_the_partitions_that_exist():
    """Round 6: the psy-2 kernel sums the spreading function over each partition's BAND (TlPsy2Tables::s_band, csrc/mp2_host.cpp tl_psy2_band) instead
    of over all 64 columns.  For every table the device path builds (six rates x models 2 / 4): the window [band_lo, band_lo + window) of row j holds
    EXACTLY row j's coefficients, every coefficient outside it that belongs to a partition that exists is zero (so the sums lose nothing: a zero
    coefficient adds +0 to a sum of non-negative terms), the window stays inside the 64 columns and is a whole number of batches."""
    import emulib
    E = emulib.lib()
    E.emu_psy2_table.argtypes = [C.c_long, C.c_int, C.c_char_p, C.c_void_p, C.c_int]

    def tab(fs, psy, name, n):
        buf = np.zeros(n)
        got = E.emu_psy2_table(fs, psy, name.encode(), buf.ctypes.data, buf.size)
        assert got > 0, name
        return buf[:got]
    for psy in (2, 4):
        for fs in (48000, 44100, 32000, 24000, 22050, 16000):
            s = tab(fs, psy, "s", 4096).reshape(64, 64)
            npart, w = int(tab(fs, psy, "npart", 4)[0]), int(tab(fs, psy, "band_w", 4)[0])
            lo = tab(fs, psy, "band_lo", 64).astype(int)
            sb = tab(fs, psy, "s_band", 48 * 64).reshape(64, 48)
            window = (w + 7) // 8 * 8
            assert 0 < w <= window <= 48 and (psy == 4 or w > 32), (psy, fs, w)
            for j in range(64):
                assert 0 <= lo[j] and lo[j] + window <= 64
                assert np.array_equal(sb[j, :window].view(np.uint64), s[j, lo[j]:lo[j] + window].view(np.uint64)), (psy, fs, j)
                assert not sb[j, window:].any()
                outside = np.ones(64, bool)
                outside[lo[j]:lo[j] + window] = False
                outside[npart:] = False                          # columns of partitions that do not exist multiply an exact +0 grouped energy
                assert not s[j, outside].any(), (psy, fs, j)


def test_transform_dealing_covers_every_butterfly_once_and_spreads_over_the_lds_banks():
    """The host deals the general butterflies of the transform's passes k = 4, 6, 8 to 128 slots (two per lane; csrc/mp2_host.cpp tl_build_tables,
    TlTables::fht_fg_lane).  (1) Every (block, i) of a pass appears exactly once (the table build aborts otherwise; here from the offsets themselves).
    (2) The property the dealing exists for (MI355X_MICROARCH.md, LDS): a ds_read_b64 is served per 32-lane half with bank = double index mod 32, a
    ds_write_b64 per 16 lanes with bank = double index mod 16 -- in passes k = 4 and k = 6 no two lanes of such a group share a bank, for any of a
    butterfly's eight points (the exchange partners are the base offsets XOR a constant: a permutation of the banks)."""
    import emulib
    E = emulib.lib()
    E.emu_fht_dealing.argtypes = [C.c_void_p]
    buf = np.zeros(3 * 128, dtype=np.uint32)
    assert E.emu_fht_dealing(buf.ctypes.data) == 384
    fg = buf.reshape(3, 128)
    unfx = lambda a: next(j for j in range(1024) if (j ^ (j >> 5)) == a)          # the layout map j -> j ^ (j >> 5), inverted
    for p, K in enumerate((4, 6, 8)):
        k1, kx = 1 << K, (1 << K) >> 1
        seen = set()
        for g in range(128):
            v = int(fg[p, g])
            if v == 0xFFFFFFFF or (p == 2 and g == 127):
                continue
            f0, g0 = unfx((v & 0xFFFF) >> 3), unfx((v >> 16) >> 3)
            blk, i = f0 // (4 * k1), f0 % (4 * k1)
            assert 1 <= i < kx and g0 == blk * 4 * k1 + k1 - i, (p, g, f0, g0)
            assert (blk, i) not in seen
            seen.add((blk, i))
        assert len(seen) == (128 // kx) * (kx - 1), (p, len(seen))
        assert int(fg[2, 127]) == ((128 ^ (128 >> 5)) << 3) << 16                 # k = 8's slot 127: the trivial butterfly, f0 = 0, g0 = kx
        if p == 2:
            continue
        for half in (0, 16):                                                      # f0 in the low 16 bits, g0 in the high
            for it in range(2):
                idx = [(int(fg[p, 64 * it + lane]) >> half & 0xFFFF) >> 3 if int(fg[p, 64 * it + lane]) != 0xFFFFFFFF else None for lane in range(64)]
                for lo in (0, 32):                                                # reads: 32-lane halves, 32 bank pairs
                    banks = [a % 32 for a in idx[lo:lo + 32] if a is not None]
                    assert len(banks) == len(set(banks)), (p, it, lo, "read")
                for lo in range(0, 64, 16):                                       # writes: 16-lane groups, 16 bank pairs
                    banks = [a % 16 for a in idx[lo:lo + 16] if a is not None]
                    if p == 0 and it == 1 and lo == 32:                          # k = 4's last 16 slots hold i = 7 of all sixteen blocks: two lanes per bank pair
                        assert max(banks.count(b) for b in set(banks)) <= 2
                    else:
                        assert len(banks) == len(set(banks)), (p, it, lo, "write")
