"""Host logic: the wave-per-stream kernel source (csrc/mp2_wave.h) executed by the lane-loop
emulation (tests/emu) against the oracle and the golden vectors -- bit-exact bytes, integer taps and
fp64 filterbank taps; SMR raw-bit equal (the device's transcendentals are glibc's, csrc/tl_libm.h).  CPU only."""
import math

import numpy as np
import pytest

import emulib as E
import oraclelib as O
from conftest import golden_cases
from pcmgen import gen_pcm


def _emu_stream(pcm, xpad=None, xpad_len=None, chunks=(None,), **cfg):
    b = E.EmuBatch([cfg])
    out, pos, taps = b"", 0, []
    n = pcm.shape[0]
    sizes = [n] if chunks == (None,) else list(chunks)
    for c in sizes:
        xp = xpad[pos:pos + c, None] if xpad is not None else None
        xl = xpad_len[pos:pos + c, None] if xpad is not None else None
        got, t = b.encode(pcm[pos:pos + c, None], xp, xl, want_taps=True)
        out += got[0]
        taps.append(t[:, 0])
        pos += c
    out += b.flush()[0]
    b.close()
    return out, np.concatenate(taps)


@pytest.mark.parametrize("path", golden_cases(), ids=lambda p: p.stem)
def test_emulated_kernel_matches_reference_golden(path):
    g = np.load(path)
    fs, mode, kbps, psy, kind, seed, pad_len, nframes = (int(v) for v in g["cfg"])
    pcm = gen_pcm(seed, kind, 0, nframes)
    xp = xl = None
    if "xpad" in g:
        xp = np.stack([E.pack_xpad(g["xpad"][i], int(g["xpad_len"][i]), pad_len) for i in range(nframes)])
        xl = g["xpad_len"].astype(np.int32)
    out, taps = _emu_stream(pcm, xp, xl, chunks=(1, 3, nframes - 4), samplerate=fs, mode=chr(mode), kbps=kbps, psy=psy,
                            pad_len=pad_len)
    assert out == g["data"].tobytes()
    nch = 1 if chr(mode) == "m" else 2
    for i in range(nframes):
        assert np.array_equal(taps[i]["scalar"][:nch], g["scalar"][i][:nch])
        assert np.array_equal(taps[i]["scfsi"][:nch], g["scfsi"][i][:nch])
        assert np.array_equal(taps[i]["bit_alloc"][:nch], g["bit_alloc"][i][:nch])
        assert (int(taps[i]["mode"]), int(taps[i]["mode_ext"])) == (int(g["mode"][i]), int(g["mode_ext"][i]))
        nsmr = 27 if psy == 1 else 32
        assert np.array_equal(taps[i]["smr"][:nch, :nsmr].view(np.uint64), g["smr"][i][:nch, :nsmr].view(np.uint64)), i
    if "sb_sample" in g:
        for k, f in enumerate(g["big_tap_frames"]):
            assert np.array_equal(taps[int(f)]["sb_sample"][:nch].view(np.uint64), g["sb_sample"][k][:nch].view(np.uint64))


FUZZ = [(psy, mode, fs, kbps) for psy in (1, 3, 0, 2) for (mode, fs, kbps) in
        (("s", 48000, 128), ("j", 48000, 128), ("j", 48000, 96), ("m", 48000, 64), ("s", 32000, 192), ("j", 24000, 64),
         ("m", 16000, 32), ("d", 48000, 256), ("j", 48000, 64), ("s", 48000, 384))]


@pytest.mark.parametrize("psy,mode,fs,kbps", FUZZ)
def test_emulated_kernel_vs_oracle_fuzz(psy, mode, fs, kbps):
    """8 signal kinds x 10 frames per configuration, several streams in one batch."""
    kinds = [k for k in range(8) if not (psy == 3 and k in (1, 3))] + ([1, 3] if psy == 3 else [])
    nframes = 10
    pcms = [gen_pcm(300 + 17 * k + psy, k, 0, nframes) for k in kinds]
    b = E.EmuBatch([dict(samplerate=fs, mode=mode, kbps=kbps, psy=psy)] * len(kinds))
    got, _ = b.encode(np.stack(pcms, axis=1))
    tail = b.flush()
    for s, k in enumerate(kinds):
        ref, _ = O.oracle_stream(pcms[s], samplerate=fs, mode=mode, kbps=kbps, psy=psy)
        assert got[s] + tail[s] == ref, (k,)
    b.close()


def test_known_bad_streams_of_round2():
    """The streams on which round 2's transcendentals (fdlibm forms, <= 1 ulp from glibc) cost a frame: the kernel source with
    glibc's own arithmetic (csrc/tl_libm.h) equals the oracle on all of them.  (The list and how it was found:
    tests/test_hip_parity.py KNOWN_BAD_R02.)"""
    from test_hip_parity import KNOWN_BAD_R02
    for fs, mode, kbps, psy, kind, seed in KNOWN_BAD_R02:
        pcm = gen_pcm(seed, kind, 0, 12)
        out, _ = _emu_stream(pcm, chunks=(1, 3, 8), samplerate=fs, mode=mode, kbps=kbps, psy=psy)
        assert out == O.oracle_stream(pcm, samplerate=fs, mode=mode, kbps=kbps, psy=psy)[0], (fs, mode, kbps, psy, kind, seed)


def test_edge_spectra_all_models():
    """tools/fuzz_psy2_edge.py in the suite: DC, Nyquist, lone impulses, exact-bin sinusoids and squares, one-LSB noise, sparse
    frames ... -- spectra that reach the special branches of the restated glibc routines (zero operands and extreme ratios in
    atan2, sincos at multiples of pi/4, the 0.0005 energy clamp) and the degenerate paths of the tone labelling -- every
    model except psy 3 (whose reference crashes on silent spectra), kernel source (emulation) against the oracle."""
    import sys
    sys.path.insert(0, str(E.ROOT / "tools"))
    import fuzz_psy2_edge as Fz
    n, bad = Fz.run(per_kind=16, seed=5, models=(2, 4, 2, 4, 1, 0), workers=4)
    assert n == 160 and bad == []


def test_emulated_log10_pow10_accuracy():
    """csrc/tl_libm.h against this machine's glibc: the same bits over the encoder's ranges (the wide sweep over every
    function is tests/test_libm_agree.py / tools/libm_agree.cpp)."""
    import math
    L = E.lib()
    rng = np.random.default_rng(0)
    worst = 0
    for x in np.concatenate([10 ** rng.uniform(-20, 3, 20000), 1 + rng.uniform(-0.1, 0.1, 5000)]):
        a, b = L.emu_log10(float(x)), math.log10(float(x))
        worst = max(worst, abs(int(np.float64(a).view(np.int64)) - int(np.float64(b).view(np.int64))))
    assert worst == 0
    # the straight-line variant used on clamped spectra is the same function on positive normals, bit for bit
    for x in np.concatenate([10 ** rng.uniform(-20, 12, 20000), 1 + rng.uniform(-0.1, 0.1, 5000), [1.0, 1e-20, 0.5, 2.0]]):
        assert np.float64(L.emu_log10_pn(float(x))).view(np.int64) == np.float64(L.emu_log10(float(x))).view(np.int64)
    worst = 0
    for x in rng.uniform(-20, 25, 20000):
        a, b = L.emu_pow10(float(x)), math.pow(10.0, float(x))
        worst = max(worst, abs(int(np.float64(a).view(np.int64)) - int(np.float64(b).view(np.int64))))
    assert worst == 0


def test_put_bits48():
    """tl_put_bits48 (a field of up to 48 bits as three OR-ed words) == writing the field bit by bit, for every start offset
    inside a word and random field lengths, including fields that end exactly on a word boundary."""
    import ctypes as C
    L = E.lib()
    L.emu_put_bits48_check.restype = C.c_long
    L.emu_put_bits48_check.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int]
    rng = np.random.default_rng(5)
    for start in range(0, 64):
        n = 600
        lens = rng.integers(1, 49, n).astype(np.int32)
        if start % 3 == 0:
            lens[::5] = 32 - (start % 32) if start % 32 else 32       # some fields ending on / spanning whole words
        vals = rng.integers(0, 1 << 62, n, dtype=np.uint64)
        assert L.emu_put_bits48_check(vals.ctypes.data, lens.ctypes.data, n, start) == 0


def test_scalefactor_index():
    """tl_sf_index (exponent bracket + three table reads) == the reference's binary search (encode_new.c:208-218): every table
    entry and its neighbours, powers of two and their neighbours, random magnitudes over the whole range, zero and denormals."""
    import ctypes as C
    L = E.lib()
    L.emu_sf_index_check.restype = C.c_long
    L.emu_sf_index_check.argtypes = [C.c_void_p, C.c_long]
    tab = (C.c_double * 64)()
    L.emu_scalefactors(tab)
    sf = np.array(tab)
    rng = np.random.default_rng(11)
    p2 = 2.0 ** np.arange(-80, 1)
    vals = np.concatenate([sf, np.nextafter(sf, 0), np.nextafter(sf, 4), p2, np.nextafter(p2, 0), np.nextafter(p2, 4),
                           10.0 ** rng.uniform(-25, 0.3, 400000), rng.uniform(0, 2, 400000), [0.0, 5e-324, 1e-310, 1e-300, 1.999999]])
    vals = np.ascontiguousarray(vals[vals < 2.0])
    assert L.emu_sf_index_check(vals.ctypes.data, len(vals)) == 0


def test_snr_columns_increase():
    """bits_for_nonoise as a count (joint-stereo trials) and the allocation rounds rely on it (encode_new.c:16-27,96-100)."""
    assert E.lib().emu_snr_monotone() == 1


def test_quantiser_division():
    """tl_div_by (reciprocal + two fused corrections) == IEEE division for every scalefactor divisor: random dividends over
    the subband-sample range, dividends that put the quotient next to a representable number, and dividends that put it
    next to a rounding midpoint (the hard cases of any division algorithm)."""
    import ctypes as C
    L = E.lib()
    L.emu_div_by_check.restype = C.c_long
    L.emu_div_by_check.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    tab = (C.c_double * 64)()
    L.emu_scalefactors(tab)                                         # the encoder's own table (encode_new.c:65-83)
    sf = np.array(tab)
    assert sf[0] == 2.0 and sf[63] == 1e-20
    # Markstein's exception: no divisor has an all-ones significand
    assert not np.any((sf.view(np.uint64) & np.uint64((1 << 52) - 1)) == np.uint64((1 << 52) - 1))
    rng = np.random.default_rng(7)
    n = 100000

    def hardest(d):
        """Dividends whose quotient by d lies within ~2^-106 (relative) of a rounding midpoint: S * 2^53 = (2M+1) * D + t
        for small t, solved for the odd 2M+1 modulo a power of two (D = integer significand of d)."""
        m, e = math.frexp(d)
        D = int(m * (1 << 53))
        z = (D & -D).bit_length() - 1
        Dp, mod = D >> z, 1 << (53 - z)
        inv = pow(Dp, -1, mod)
        out = []
        for tp in range(-15, 16, 2):
            r = (-tp * inv) % mod
            for k in range(0, 1 << z if z < 6 else 64):
                o = r + k * mod
                if not (o & 1):
                    continue
                o |= 1 << 53                                        # 2M+1 in [2^53, 2^54)
                num = o * D + (tp << z)
                if num % (1 << 53):
                    continue
                S = num >> 53
                while S >= (1 << 53) and not (S & 1):
                    S >>= 1
                if (1 << 52) <= S < (1 << 53):
                    out.append(math.ldexp(float(S), -52))
        return out

    nhard = 0
    for d in np.concatenate([sf, np.arange(1.0, 201.0)]):           # scalefactors; critical-band widths (psy 1 noise weights)
        s_rand = rng.uniform(-2.5, 2.5, n) * 10.0 ** rng.uniform(-12, 0, n)
        q = rng.uniform(0.5, 4.0, n) * rng.choice([-1.0, 1.0], n)
        near_rep = np.nextafter(q * d, rng.choice([-np.inf, np.inf], n))      # quotient within an ulp of q
        half = (np.nextafter(np.abs(q), np.inf) - np.abs(q)) / 2
        mid = (np.abs(q).astype(np.longdouble) + half.astype(np.longdouble)) * np.sign(q).astype(np.longdouble)
        near_mid = (mid * np.longdouble(d)).astype(np.float64)              # quotient within an ulp-fraction of a midpoint
        hard = np.array(hardest(float(d)), dtype=np.float64)
        nhard += len(hard)
        s_all = np.ascontiguousarray(np.concatenate([s_rand, q * d, near_rep, near_mid, np.nextafter(near_mid, np.inf),
                                                     np.nextafter(near_mid, -np.inf), hard, -hard]))
        d_all = np.full(s_all.shape, d)
        assert L.emu_div_by_check(s_all.ctypes.data, d_all.ctypes.data, len(s_all)) == 0
    assert nhard > 1000


def test_emulated_illegal_configs():
    for kw in (dict(samplerate=11025), dict(kbps=100), dict(mode="x"), dict(psy=5), dict(psy=-1), dict(pad_len=-1), dict(pad_len=999)):
        with pytest.raises(ValueError):
            E.EmuBatch([kw])


def test_xpad_length_contract():
    """d_xpad_len is 0 or 2..pad_len (include/toolame_batch.h).  A value above the stream's toolame_set_pad() length, above the
    record size, or 1, and any value on a stream created with pad_len 0, makes the frame carry no PAD: the bytes equal an
    encode without PAD (nothing is written outside the frame).  A pad length that leaves no room for header, CRC and bit
    allocation is refused when the stream is created."""
    nf = 4
    pcm = gen_pcm(77, 0, 0, nf)
    rng = np.random.default_rng(9)
    xp = rng.integers(0, 256, size=(nf, E.TL_MAX_XPAD), dtype=np.uint8)
    for cfg, bad in ((dict(mode="j", psy=1, pad_len=20), (21, 58, 200, 201, 100000, 1, -3)),
                     (dict(mode="s", psy=3, pad_len=0), (2, 16, 58, 200, 4096)),
                     (dict(mode="m", psy=1, kbps=8, samplerate=24000, pad_len=0), (58, 48, 44, 2))):
        ref, _ = _emu_stream(pcm, **cfg)
        for v in bad:
            xl = np.full(nf, v, dtype=np.int32)
            got, _ = _emu_stream(pcm, xp, xl, **cfg)
            assert got == ref, (cfg, v)
    # lengths inside the contract still carry their bytes (and differ from the PAD-less stream)
    cfg = dict(mode="j", psy=1, pad_len=20)
    ref, _ = _emu_stream(pcm, **cfg)
    got, _ = _emu_stream(pcm, xp, np.full(nf, 20, dtype=np.int32), **cfg)
    assert got != ref and len(got) == len(ref)
    e = O.OracleEncoder(mode="j", psy=1, pad_len=20)          # reference layout: padlen + 1 bytes, the last one = the valid length
    want = b"".join(e.encode(pcm[i], bytes(xp[i, :20]) + b"\x14", 20) for i in range(nf)) + e.finish()
    e.close()
    assert got == want
    for kw in (dict(samplerate=24000, mode="m", kbps=8, pad_len=58), dict(samplerate=24000, mode="m", kbps=8, pad_len=40),
               dict(samplerate=48000, mode="m", kbps=32, pad_len=90), dict(samplerate=16000, mode="m", kbps=8, pad_len=66)):
        with pytest.raises(ValueError):
            E.EmuBatch([kw])
    E.EmuBatch([dict(samplerate=24000, mode="m", kbps=8, pad_len=16)]).close()


@pytest.mark.parametrize("fs,mode,kbps", [(44100, "s", 128), (44100, "j", 192), (44100, "m", 64), (44100, "d", 384), (44100, "s", 96),
                                          (22050, "s", 64), (22050, "m", 32), (22050, "j", 160), (22050, "m", 8)])
def test_padding_slot_rates_vs_oracle(fs, mode, kbps):
    """44.1 and 22.05 kHz: frames of two lengths (padding slots, availbits.c:49-62; header bit 9).  Every psy model, frames fed
    in ragged chunks (the slot recurrence's state passes from call to call), bytes against the oracle."""
    nf = 14
    for psy in (0, 1, 2, 3):
        pcm = gen_pcm(900 + psy + kbps, (0, 7, 5, 4)[psy], 0, nf)
        ref, _ = O.oracle_stream(pcm, samplerate=fs, mode=mode, kbps=kbps, psy=psy)
        got, _ = _emu_stream(pcm, chunks=(1, 2, 4, nf - 7), samplerate=fs, mode=mode, kbps=kbps, psy=psy)
        assert got == ref, (fs, mode, kbps, psy)
        if psy == 1:
            b = E.EmuBatch([dict(samplerate=fs, mode=mode, kbps=kbps, psy=psy)])
            whole = b.frame_bytes[0]
            assert len(ref) > nf * whole or (1152 * kbps * 125) % fs == 0      # some frames carry the extra slot
            b.close()


def test_mono_streams_share_waves_in_pairs():
    """Two mono streams of one configuration are encoded by ONE wave (csrc/mp2_wave.h tl_encode_pair: lane = 2*sb + unit; the batch
    pairs consecutive mono streams of a configuration).  A batch with pairs of every psy model -- padded rates, LSF, X-PAD of different
    lengths per stream and frame, an odd stream left alone, a stereo stream in between -- against the oracle stream by stream, frames
    fed in ragged chunks; and the same batch with pairing switched off gives the same bytes."""
    import os
    cfgs = ([dict(samplerate=48000, mode="m", kbps=64, psy=1)] * 5 + [dict(samplerate=48000, mode="j", kbps=128, psy=1)] +
            [dict(samplerate=24000, mode="m", kbps=32, psy=3)] * 3 + [dict(samplerate=44100, mode="m", kbps=64, psy=1)] * 2 +
            [dict(samplerate=48000, mode="m", kbps=96, psy=0)] * 2 + [dict(samplerate=32000, mode="m", kbps=64, psy=4)] * 3 +
            [dict(samplerate=48000, mode="m", kbps=128, psy=2)] * 2 + [dict(samplerate=22050, mode="m", kbps=32, psy=3)] * 2 +
            [dict(samplerate=48000, mode="m", kbps=80, psy=3, pad_len=24)] * 4 + [dict(samplerate=16000, mode="m", kbps=8, psy=1)] * 2)
    ns, nf = len(cfgs), 9
    rng = np.random.default_rng(31)
    pcm = np.stack([gen_pcm(3300 + s, (0, 7, 4, 5, 0, 2)[s % 6], 0, nf) for s in range(ns)], axis=1)
    xp = rng.integers(0, 256, size=(nf, ns, E.TL_MAX_XPAD), dtype=np.uint8)
    xl = np.zeros((nf, ns), dtype=np.int32)
    for s, c in enumerate(cfgs):
        if c.get("pad_len"):
            xl[:, s] = rng.choice([0, 2, 6, 24, 17], size=nf)
    want = []
    for s, c in enumerate(cfgs):
        e = O.OracleEncoder(**c)
        pl = c.get("pad_len", 0)
        chunks = []
        for f in range(nf):
            n = int(xl[f, s])
            # reference layout: pad_len + 1 bytes, the n bytes that are sent sit at [pad_len - n, pad_len), the last byte = the valid length
            chunks.append(e.encode(pcm[f, s], bytes(pl - n) + bytes(xp[f, s, :n]) + bytes([n]), n) if pl else e.encode(pcm[f, s]))
        want.append(b"".join(chunks) + e.finish())
        e.close()
    def run():
        b = E.EmuBatch(cfgs)
        got, pos = [b""] * ns, 0
        for n in (1, 3, 2, nf - 6):
            g, _ = b.encode(pcm[pos:pos + n], xp[pos:pos + n], xl[pos:pos + n])
            got = [a + c for a, c in zip(got, g)]
            pos += n
        tail = b.flush()
        npair = b.pair_units()
        b.close()
        return [a + c for a, c in zip(got, tail)], npair
    got, npair = run()
    # 5 + 3 + 2 + 2 + 3 + 2 + 2 + 4 + 2 mono streams in groups of one configuration: 2 + 1 + 1 + 1 + 1 + 1 + 1 + 2 + 1 pairs, every frame of them
    assert npair == 11 * nf, npair
    for s in range(ns):
        assert got[s] == want[s], (s, cfgs[s])
    os.environ["EMU_NO_PAIRS"] = "1"
    try:
        alone, none = run()
        assert none == 0 and alone == got
    finally:
        del os.environ["EMU_NO_PAIRS"]


def test_mono_pairs_random_configurations_equal_lone_streams():
    """Random mono configurations (rates, bitrates, psy models, X-PAD) x signal kinds -- silence, impulses and square waves with their
    allocation ties among them -- as pairs of streams per wave (tl_encode_pair, tl_allocate_pair: both units' greedy loops at once)
    against the same batch with every stream alone (tl_allocate): the bytes must not depend on the pairing."""
    import os
    v1m = (32, 48, 56, 64, 80, 96, 112, 128, 160, 192); v2 = (8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160)
    rng = np.random.default_rng(77)
    nf, cfgs = 4, []
    for _ in range(16):
        fs = int(rng.choice([48000, 44100, 32000, 24000, 22050, 16000]))
        c = dict(samplerate=fs, mode="m", kbps=int(rng.choice(v1m if fs >= 32000 else v2)), psy=int(rng.choice([0, 1, 1, 3, 2, 4])))
        if rng.random() < 0.3: c["pad_len"] = 24
        cfgs += [c] * 2
    ns = len(cfgs)
    kinds = [int(k) if not (cfgs[s]["psy"] == 3 and k in (1, 3)) else 0 for s, k in enumerate(rng.integers(0, 8, size=ns))]
    pcm = np.stack([gen_pcm(int(rng.integers(1 << 30)), kinds[s], 0, nf) for s in range(ns)], axis=1)
    xp = rng.integers(0, 256, size=(nf, ns, E.TL_MAX_XPAD), dtype=np.uint8)
    xl = np.zeros((nf, ns), dtype=np.int32)
    for s, c in enumerate(cfgs):
        if c.get("pad_len"): xl[:, s] = rng.choice([0, 2, 6, 24, 17], size=nf)
    def run():
        b = E.EmuBatch(cfgs)
        g, _ = b.encode(pcm, xp, xl)
        t, n = b.flush(), b.pair_units()
        b.close()
        return [a + c for a, c in zip(g, t)], n
    paired, npair = run()
    assert npair == 16 * nf
    os.environ["EMU_NO_PAIRS"] = "1"
    try:
        alone, none = run()
    finally:
        del os.environ["EMU_NO_PAIRS"]
    assert none == 0
    for s in range(ns):
        assert paired[s] == alone[s], (s, cfgs[s], kinds[s])


def test_power_spectrum_deferral_runs_full():
    """The power spectrum computes the logarithm's table branch for all 512 lines and files the lines e_log.c sends through its close-to-1
    branch (mp2_psy13.h: tl_power_db_main / tl_power_near1); 64 filed lines are processed at once.  On ordinary signals a channel files
    ~47 lines, so the pass that runs while the line loop is still going never happens.  A lone impulse gives every line the SAME energy:
    an amplitude whose energy has a mantissa just above 1 files all 512 lines of a frame -- eight full passes.  The kernel source has to
    take that path (counter) and still equal the oracle."""
    L = E.lib()
    import ctypes
    L.emu_near1_full_flushes.restype = ctypes.c_long
    nframes, hits = 3, 0
    for amp in range(2000, 32000, 37):
        pcm = np.zeros((nframes, 2, 1152), dtype=np.int16)
        pcm[1, :, 700] = amp                                          # one sample, both channels, inside frame 1's analysis window
        before = L.emu_near1_full_flushes()
        out, _ = _emu_stream(pcm, chunks=(nframes,), samplerate=48000, mode="s", kbps=128, psy=1)
        if L.emu_near1_full_flushes() - before >= 8:
            hits += 1
            assert out == O.oracle_stream(pcm, samplerate=48000, mode="s", kbps=128, psy=1)[0], amp
            if hits >= 3:
                break
    assert hits >= 3

