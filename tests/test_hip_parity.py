"""Parity tests proper: the HIP path (through the C-ABI of include/toolame_batch.h) against the
golden vectors of the real reference and against the oracle, on a real MI355X.  Bit-exact for
bytes and every integer tap; fp64 filterbank taps compared as raw bits; the device's transcendentals are
glibc 2.35's operation for operation (csrc/tl_libm.h), so every model -- 2 and 4 included -- is held to the oracle's
bytes on degenerate signals too, and bit-exact against the host emulation of the same kernel source."""
from pathlib import Path

import numpy as np
import pytest

import emulib as E
import oraclelib as O
from conftest import golden_cases
from framecheck import crc16_frame_ok as _crc16_frame_ok
from pcmgen import gen_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import odr_audioenc_amd as mod
    mod.load_library()
    return mod


def _xpad_arrays(g, nframes, pad_len):
    xp = np.zeros((nframes, 1, E.TL_MAX_XPAD), dtype=np.uint8)
    xl = np.zeros((nframes, 1), dtype=np.int32)
    for i in range(nframes):
        xl[i, 0] = int(g["xpad_len"][i])
        xp[i, 0] = E.pack_xpad(g["xpad"][i], int(g["xpad_len"][i]), pad_len)
    return xp, xl


def test_golden_all_cases_one_batch(M):
    """Every golden case as one stream of a single mixed-configuration batch (mixed rates, modes,
    bitrates, psy models, X-PAD) -- bytes must equal the reference's output."""
    cases = [np.load(p) for p in golden_cases()]
    cfgs, pcms, has_xpad = [], [], False
    nframes = int(cases[0]["cfg"][7])
    for g in cases:
        fs, mode, kbps, psy, kind, seed, pad_len, nf = (int(v) for v in g["cfg"])
        assert nf == nframes
        cfgs.append(M.StreamConfig(samplerate=fs, mode=chr(mode), bitrate=kbps, psy_model=psy, pad_len=pad_len))
        pcms.append(gen_pcm(seed, kind, 0, nframes))
        has_xpad |= pad_len > 0
    pcm = np.stack(pcms, axis=1)
    xp = np.zeros((nframes, len(cases), E.TL_MAX_XPAD), dtype=np.uint8)
    xl = np.zeros((nframes, len(cases)), dtype=np.int32)
    for s, g in enumerate(cases):
        if "xpad" in g:
            a, b = _xpad_arrays(g, nframes, int(g["cfg"][6]))
            xp[:, s] = a[:, 0]
            xl[:, s] = b[:, 0]
    b = M.Batch(cfgs)
    got, taps = b.encode(pcm, xp, xl, want_taps=True)
    tail = b.flush()
    for s, (g, p) in enumerate(zip(cases, golden_cases())):
        assert got[s] + tail[s] == g["data"].tobytes(), p.stem
        nch = 1 if chr(int(g["cfg"][1])) == "m" else 2
        for f in range(nframes):
            t = taps[f, s]
            assert np.array_equal(t["bit_alloc"][:nch], g["bit_alloc"][f][:nch]), (p.stem, f)
            assert np.array_equal(t["scfsi"][:nch], g["scfsi"][f][:nch]), (p.stem, f)
    b.close()


def test_filterbank_taps_bit_exact(M):
    """sb_sample (fp64) and quantised samples against the reference's taps, raw bits."""
    for p in golden_cases():
        g = np.load(p)
        if "sb_sample" not in g:
            continue
        fs, mode, kbps, psy, kind, seed, pad_len, nframes = (int(v) for v in g["cfg"])
        nch = 1 if chr(mode) == "m" else 2
        pcm = gen_pcm(seed, kind, 0, nframes)[:, None]
        b = M.Batch([M.StreamConfig(samplerate=fs, mode=chr(mode), bitrate=kbps, psy_model=psy, pad_len=pad_len)])
        xp, xl = _xpad_arrays(g, nframes, pad_len) if "xpad" in g else (None, None)
        _, taps = b.encode(pcm, xp, xl, want_taps=True)
        for i, f in enumerate(g["big_tap_frames"]):
            t = taps[int(f), 0]
            assert np.array_equal(t["sb_sample"][:nch].view(np.uint64), g["sb_sample"][i][:nch].view(np.uint64)), p.stem
            # the reference only rewrites subband[] where it transmits samples (stale values elsewhere)
            for ch in range(nch):
                own = (t["bit_alloc"][ch] != 0) & ((np.arange(32) < int(t["jsbound"])) | (ch == 0))
                assert np.array_equal(t["subband"][ch][..., own], g["subband"][i][ch][..., own]), p.stem
            sbl = 32
            assert np.array_equal(t["scalar"][:nch], g["scalar"][int(f)][:nch]), p.stem
            assert np.array_equal(np.ascontiguousarray(t["smr"][:nch, :27]).view(np.uint64), np.ascontiguousarray(g["smr"][int(f)][:nch, :27]).view(np.uint64)), p.stem   # raw bits
        b.close()


def test_device_equals_host_emulation_bitwise(M):
    """Same kernel source, two compilers: gfx950 vs the lane-loop emulation. Everything bit-identical,
    including SMR (shared deterministic log10/pow10)."""
    cfgs = [dict(psy=1, mode="j"), dict(psy=3, mode="s"), dict(psy=1, mode="m", kbps=64, samplerate=32000),
            dict(psy=3, mode="j", kbps=192), dict(psy=0, mode="d"), dict(psy=3, mode="m", kbps=64, samplerate=24000),
            dict(psy=2, mode="j"), dict(psy=2, mode="s", kbps=192, samplerate=32000)]
    nframes = 6
    pcm = np.stack([gen_pcm(50 + s, [0, 7, 4, 5, 2, 6, 0, 7][s], 0, nframes) for s in range(len(cfgs))], axis=1)
    e = E.EmuBatch(cfgs)
    eg, et = e.encode(pcm, want_taps=True)
    b = M.Batch([M.StreamConfig(samplerate=c.get("samplerate", 48000), mode=c["mode"], bitrate=c.get("kbps", 128),
                                psy_model=c["psy"]) for c in cfgs])
    dg, dt = b.encode(pcm, want_taps=True)
    assert dg == eg and b.flush() == e.flush()
    for name in ("sb_sample", "smr", "max_sc"):
        assert np.array_equal(dt[name].view(np.uint64), et[name].view(np.uint64)), name
    for name in ("subband", "scalar", "scalar_pre", "j_scale", "scfsi", "bit_alloc", "adb_left", "mode", "mode_ext", "crc16", "scfcrc"):
        assert np.array_equal(dt[name], et[name]), name
    b.close()
    e.close()


@pytest.mark.parametrize("psy,mode,nstreams,nframes", [(1, "j", 96, 12), (3, "s", 96, 12), (1, "s", 64, 24), (2, "j", 64, 12)])
def test_many_streams_vs_oracle(M, psy, mode, nstreams, nframes):
    """stream i uses seed i; frames fed in ragged chunks (1, 2, 5, rest) to exercise the
    state hand-over between launches."""
    pcm = np.stack([gen_pcm(1000 + s, s % 8 if not (psy == 3 and s % 8 in (1, 3)) else 0, 0, nframes) for s in range(nstreams)], axis=1)
    b = M.Batch([M.StreamConfig(mode=mode, psy_model=psy)] * nstreams)
    chunks, pos = [b""] * nstreams, 0
    for n in (1, 2, 5, nframes - 8):
        got, _ = b.encode(pcm[pos:pos + n])
        chunks = [a + c for a, c in zip(chunks, got)]
        pos += n
    tail = b.flush()
    for s in range(nstreams):
        ref, _ = O.oracle_stream(pcm[:, s], mode=mode, psy=psy)
        assert chunks[s] + tail[s] == ref, s
    b.close()


@pytest.mark.parametrize("psy", [0, 1, 2, 3])
def test_kernel_variants_agree(M, psy):
    """The host picks kernel VARIANTS from the composition of a batch (csrc/tlb_batch.cpp): a list of two-channel streams only runs the
    kernels built with the channel count as a constant, a batch with one configuration passes no configuration-index table, a launch over
    every stream of the batch passes no stream list.  The same stereo streams must come out byte for byte the same (a) alone in a batch --
    every shortcut taken --, (b) next to one mono stream of another bitrate -- the generic kernels with pairs code, an index table -- and
    (c) next to one stream of another psy model -- two lists, each with its stream list; and all of it equals the oracle."""
    n, nframes = 24, 9
    pcm = np.stack([gen_pcm(4000 + s, s % 8 if not (psy == 3 and s % 8 in (1, 3)) else 0, 0, nframes) for s in range(n)], axis=1)
    extra = gen_pcm(4999, 2, 0, nframes)[:, None]
    base = [M.StreamConfig(mode="j" if s % 2 else "s", psy_model=psy) for s in range(n)] if psy != 3 else [M.StreamConfig(mode="s", psy_model=3)] * n

    def run(cfgs, data):
        b = M.Batch(cfgs)
        a, _ = b.encode(data[:4]); c, _ = b.encode(data[4:])
        t = b.flush()
        b.close()
        return [x + y + z for x, y, z in zip(a, c, t)]

    alone = run([M.StreamConfig(mode="s", psy_model=psy)] * n, pcm)                      # one configuration, one list, stereo only
    two_cfg = run(base, pcm)                                                             # (for psy != 3: two configurations -> an index table)
    with_mono = run([M.StreamConfig(mode="s", psy_model=psy)] * n + [M.StreamConfig(mode="m", bitrate=64, psy_model=psy)], np.concatenate([pcm, extra], axis=1))
    other = 0 if psy != 0 else 1
    with_other = run([M.StreamConfig(mode="s", psy_model=psy)] * n + [M.StreamConfig(mode="s", psy_model=other)], np.concatenate([pcm, extra], axis=1))
    assert alone == with_mono[:n] == with_other[:n]
    for s in range(n):
        ref, _ = O.oracle_stream(pcm[:, s], mode="s", psy=psy)
        assert alone[s] == ref, s
        ref2, _ = O.oracle_stream(pcm[:, s], mode=base[s].mode, psy=psy)
        assert two_cfg[s] == ref2, s
    refm, _ = O.oracle_stream(extra[:, 0], mode="m", kbps=64, psy=psy)
    assert with_mono[n] == refm
    refo, _ = O.oracle_stream(extra[:, 0], mode="s", psy=other)
    assert with_other[n] == refo


def test_power_spectrum_deferral_runs_full_on_the_device(M):
    """tests/test_emu_parity.py::test_power_spectrum_deferral_runs_full on the device: lone impulses whose energy files all 512 lines of a
    frame for the logarithm's close-to-1 branch (the amplitudes are found with the emulation's counter), next to ordinary streams in one batch."""
    import ctypes
    L = E.lib()
    L.emu_near1_full_flushes.restype = ctypes.c_long
    nframes, amps = 3, []
    for amp in range(2000, 32000, 37):
        pcm = np.zeros((nframes, 1, 2, 1152), dtype=np.int16)
        pcm[1, 0, :, 700] = amp
        before = L.emu_near1_full_flushes()
        eb = E.EmuBatch([dict(psy=1)])
        eb.encode(pcm)
        eb.close()
        if L.emu_near1_full_flushes() - before >= 8:
            amps.append(amp)
            if len(amps) >= 6:
                break
    assert len(amps) >= 3
    pcms = []
    for k, amp in enumerate(amps):
        p = np.zeros((nframes, 2, 1152), dtype=np.int16)
        p[1, :, 700] = amp
        if k % 2:
            p[1, 1, 700] = 0                                          # one channel only: the other one is digital silence
        pcms.append(p)
    pcms += [gen_pcm(77 + k, 0, 0, nframes) for k in range(3)]
    b = M.Batch([M.StreamConfig(mode="s", psy_model=1)] * len(pcms))
    got, _ = b.encode(np.stack(pcms, axis=1))
    tail = b.flush()
    b.close()
    for s, p in enumerate(pcms):
        assert got[s] + tail[s] == O.oracle_stream(p, mode="s", psy=1)[0], s


@pytest.mark.parametrize("nstreams", [1, 7, 9, 37, 3100])
def test_unit_lists_odd_shapes(M, nstreams):
    """The (stream, frame) units of a launch come off eight per-XCD lists (stream k on list k % 8; a wave's first unit is its rank,
    the rest off the list heads, then the other lists): every unit must be encoded exactly once whatever the shape -- fewer streams
    than lists, fewer units than waves, more units than waves; launches of 1, 2 and 5 frames.  Models 1 (one kernel) and 0 / 2
    (encode kernel alone / after the psy-2 kernel); every stream against the oracle up to 37 streams, 40 sampled ones of 3100 --
    plus, for every stream, the sync word of every frame (a unit that was never encoded leaves zeros) and equality with the
    stream 64 before it (the signals repeat every 64 streams)."""
    nframes = 8
    for psy, mode in ((1, "j"), (0, "s"), (2, "s")) if nstreams <= 37 else ((1, "s"),):
        base = [gen_pcm(3000 + s, s % 8, 0, nframes) for s in range(min(nstreams, 64))]
        pcm = np.stack([base[s % 64] for s in range(nstreams)], axis=1)
        b = M.Batch([M.StreamConfig(mode=mode, psy_model=psy)] * nstreams)
        chunks, pos = [b""] * nstreams, 0
        for n in (1, 2, 5):
            got, _ = b.encode(pcm[pos:pos + n])
            chunks = [a + c for a, c in zip(chunks, got)]
            pos += n
        tail = b.flush()
        b.close()
        refs = {}
        for s in (range(nstreams) if nstreams <= 37 else list(range(0, nstreams, 80)) + [nstreams - 1]):
            if s % 64 not in refs:
                refs[s % 64] = O.oracle_stream(pcm[:, s], mode=mode, psy=psy)[0]
            assert chunks[s] + tail[s] == refs[s % 64], (psy, nstreams, s)
        for s in range(nstreams):
            data = chunks[s] + tail[s]
            assert len(data) == 384 * nframes, (psy, nstreams, s)
            assert all(data[384 * f:384 * f + 2] == b"\xff\xfc" for f in range(nframes)), (psy, nstreams, s)
            if s >= 64:
                assert data == chunks[s - 64] + tail[s - 64], (psy, nstreams, s)


@pytest.mark.parametrize("nstreams,launches", [(1, (40, 3, 57)), (2, (33, 1)), (5, (7, 1, 12)), (3100, (5, 4))])
def test_psy2_runs_of_frames_odd_shapes(M, nstreams, launches):
    """Models 2 and 4: the psy-2 kernel's units are RUNS of frames of one channel (whole chains for the complete rounds of waves, the
    last round's chains cut into runs that seed themselves from the PCM before them; csrc/mp2_host.cpp tl_psy2_plan).  Shapes that
    make the planner cut differently -- one stream with many frames (2 chains on 3072 waves), a handful, mono and stereo mixed, more
    chains than waves -- over several launches (the state passes through the stream's record between launches): bytes against the oracle."""
    nframes = sum(launches)
    for psy in (2, 4):
        cfgs = [M.StreamConfig(mode=("s", "m", "j")[s % 3] if nstreams <= 5 else "s", bitrate=128 if s % 3 != 1 or nstreams > 5 else 64, psy_model=psy) for s in range(nstreams)]
        base = [gen_pcm(8100 + s, (0, 7, 4, 5, 2)[s % 5], 0, nframes) for s in range(min(nstreams, 64))]
        pcm = np.stack([base[s % 64] for s in range(nstreams)], axis=1)
        b = M.Batch(cfgs)
        chunks, pos = [b""] * nstreams, 0
        for n in launches:
            got, _ = b.encode(pcm[pos:pos + n])
            chunks = [a + c for a, c in zip(chunks, got)]
            pos += n
        tail = b.flush()
        b.close()
        refs = {}
        for s in (range(nstreams) if nstreams <= 5 else list(range(0, nstreams, 97)) + [nstreams - 1]):
            c = cfgs[s]
            key = (s % 64, c.mode, c.bitrate)
            if key not in refs:
                refs[key] = O.oracle_stream(pcm[:, s], mode=c.mode, kbps=c.bitrate, psy=psy)[0]
            assert chunks[s] + tail[s] == refs[key], (psy, nstreams, s)
        if nstreams > 64:
            for s in range(64, nstreams):
                assert chunks[s] + tail[s] == chunks[s - 64] + tail[s - 64], (psy, s)


@pytest.mark.parametrize("nframes", [5, 6, 9, 13])
def test_host_path_chunking(M, nframes):
    """tlb_encode_host cuts a big call (>= 8 MB of PCM) into up to four chunks of whole frames whose copies and kernels overlap;
    frame counts that do not divide evenly (5 = 2 + 2 + 1: three chunks) must come out like the same frames fed one per call
    (never chunked).  1900 streams x 4.6 KB = 8.7 MB per frame."""
    nstreams = 1900
    base = [gen_pcm(5000 + s, s % 8, 0, nframes) for s in range(64)]
    pcm = np.stack([base[s % 64] for s in range(nstreams)], axis=1)
    b1 = M.Batch([M.StreamConfig(mode="j", psy_model=1)] * nstreams)
    whole, _ = b1.encode(pcm)
    t1 = b1.flush()
    b1.close()
    b2 = M.Batch([M.StreamConfig(mode="j", psy_model=1)] * nstreams)
    parts = [b""] * nstreams
    for f in range(nframes):
        got, _ = b2.encode(pcm[f:f + 1])
        parts = [a + c for a, c in zip(parts, got)]
    t2 = b2.flush()
    b2.close()
    assert whole == parts and t1 == t2
    ref = O.oracle_stream(pcm[:, 5], mode="j", psy=1)[0]
    assert whole[5] + t1[5] == ref


@pytest.mark.parametrize("seed", range(8))
def test_api_shapes_fuzz(M, seed):
    """Random batch sizes and call patterns: the same frames through calls of random lengths (some big enough for the chunked
    host path) and through one call per frame must agree byte for byte, and two random streams must equal the oracle."""
    rng = np.random.default_rng(1000 + seed)
    nstreams = int(rng.choice([1, 2, 5, 8, 13, 100, 700, 2500]))
    psy = int(rng.choice([0, 1, 1, 2, 3, 4]))
    mode = str(rng.choice(["s", "j"]))
    kbps = int(rng.choice([128, 192]))
    lens = []
    while sum(lens) < 14:
        lens.append(int(rng.integers(1, 8)))
    nframes = sum(lens)
    base = [gen_pcm(7000 + 10 * seed + s, (s + seed) % 8 if not (psy == 3 and (s + seed) % 8 in (1, 3)) else 0, 0, nframes) for s in range(min(nstreams, 16))]
    pcm = np.stack([base[s % 16] for s in range(nstreams)], axis=1)
    cfg = M.StreamConfig(mode=mode, bitrate=kbps, psy_model=psy)
    b1, b2 = M.Batch([cfg] * nstreams), M.Batch([cfg] * nstreams)
    a, pos = [b""] * nstreams, 0
    for n in lens:
        got, _ = b1.encode(pcm[pos:pos + n])
        a = [x + y for x, y in zip(a, got)]
        pos += n
    c = [b""] * nstreams
    for f in range(nframes):
        got, _ = b2.encode(pcm[f:f + 1])
        c = [x + y for x, y in zip(c, got)]
    ta, tc = b1.flush(), b2.flush()
    b1.close()
    b2.close()
    assert a == c and ta == tc, (nstreams, psy, mode, lens)
    for s in {0, nstreams - 1}:
        assert a[s] + ta[s] == O.oracle_stream(pcm[:, s], mode=mode, kbps=kbps, psy=psy)[0], (nstreams, psy, mode, s)


def _against_oracle(M, jobs, nframes, chunks=None):
    """jobs = [(fs, mode, kbps, psy, kind, seed)]: one mixed batch on the device, every stream against the oracle (references
    computed on all host cores); returns the jobs whose bytes differ."""
    from concurrent.futures import ThreadPoolExecutor
    pcms = [gen_pcm(seed, kind, 0, nframes) for (_, _, _, _, kind, seed) in jobs]

    def ref(i):
        fs, mode, kbps, psy, _, _ = jobs[i]
        return O.oracle_stream(pcms[i], samplerate=fs, mode=mode, kbps=kbps, psy=psy)[0]
    O.lib()
    with ThreadPoolExecutor(16) as ex:                   # the oracle is a ctypes call: the GIL is released inside it
        refs = list(ex.map(ref, range(len(jobs))))
    b = M.Batch([M.StreamConfig(samplerate=fs, mode=mode, bitrate=kbps, psy_model=psy) for (fs, mode, kbps, psy, _, _) in jobs])
    pcm = np.stack(pcms, axis=1)
    out, pos = [b""] * len(jobs), 0
    for n in (chunks or [nframes]):
        got, _ = b.encode(pcm[pos:pos + n])
        out = [x + y for x, y in zip(out, got)]
        pos += n
    tail = b.flush()
    b.close()
    return [j for j, o, t, r in zip(jobs, out, tail, refs) if o + t != r]


# The complete set of streams on which round 2's device path (fdlibm-style transcendentals, <= 1 ulp from glibc) differed from
# the oracle: found again by running round 2's kernel source (host emulation, bit-equal to the device) over every rate / mode /
# bitrate x psy 2, 4 x {lone impulse, full-scale square waves of period 2..64} -- the two signal classes of all 23 mismatching
# streams of the round-2 GPU soak (profiles/soak_r02.txt).  (fs, mode, kbps, psy, kind, seed): kind 2 = square wave of period
# 2 + seed % 63 samples, kind 3 = impulse.
KNOWN_BAD_R02 = [(48000, "s", 192, 4, 2, 14), (48000, "s", 384, 4, 2, 14), (48000, "j", 128, 4, 2, 14), (48000, "j", 192, 4, 2, 14),
                 (48000, "j", 384, 4, 2, 14), (48000, "d", 192, 4, 2, 14), (48000, "d", 384, 4, 2, 14), (44100, "s", 384, 2, 3, 1)]


def test_known_bad_streams_of_round2(M):
    """Every stream that differed in round 2 now equals the oracle byte for byte, whole and in ragged chunks."""
    assert _against_oracle(M, KNOWN_BAD_R02, 12) == []
    assert _against_oracle(M, KNOWN_BAD_R02, 12, chunks=[1, 3, 8]) == []


def test_degenerate_signals_psy2_psy4(M):
    """The sweep the known-bad set came from, on the device: every (rate, mode, bitrate) x psy 2, 4 x lone impulse, and at
    48 / 44.1 / 32 kHz x full-scale square waves of every period 2..64 -- 2900 streams whose spectra are hundreds of lines of
    nearly equal level, where one ulp of a logarithm decides a tone test or an allocation tie."""
    rates = {48000: [(m, k) for m in "sjd" for k in (64, 96, 128, 160, 192, 256, 384)] + [("m", k) for k in (32, 48, 64, 96, 128, 192)],
             32000: [("s", 128), ("j", 192), ("m", 64), ("m", 96), ("d", 256)],
             24000: [("s", 64), ("j", 96), ("m", 32), ("m", 64), ("s", 128)],
             16000: [("m", 24), ("s", 48), ("j", 64)],
             44100: [("s", 128), ("j", 192), ("m", 64), ("d", 256), ("s", 384), ("j", 96)],
             22050: [("m", 32), ("s", 64), ("j", 160), ("m", 8), ("s", 128)]}
    jobs = []
    for fs, lst in rates.items():
        for mode, kbps in lst:
            for psy in (2, 4):
                jobs.append((fs, mode, kbps, psy, 3, 1))
                if fs in (48000, 44100, 32000):
                    jobs += [(fs, mode, kbps, psy, 2, P - 2) for P in range(2, 65)]
    assert len(jobs) > 2900
    assert _against_oracle(M, jobs, 6) == []


@pytest.mark.parametrize("seed", [201, 202])
def test_soak_slice_all_models(M, seed):
    """A slice of tools/soak_gpu.py inside the suite: 1536 streams of random legal configurations x all five psy models x
    all eight signal kinds, 8 frames in ragged chunks, against the oracle.  No model is exempt."""
    rng = np.random.default_rng(seed)
    rates = {48000: [(m, k) for m in "sjdm" for k in ((64, 96, 128, 160, 192, 256, 384) if m != "m" else (32, 48, 64, 96, 128, 192))],
             32000: [("s", 128), ("j", 192), ("m", 64)], 24000: [("s", 64), ("j", 96), ("m", 32)], 16000: [("m", 24), ("s", 48)],
             44100: [("s", 128), ("j", 192), ("s", 384)], 22050: [("m", 32), ("s", 64), ("j", 160)]}
    combos = [(fs, m, k) for fs, lst in rates.items() for m, k in lst]
    jobs = []
    while len(jobs) < 1536:
        fs, mode, kbps = combos[rng.integers(len(combos))]
        psy = int(rng.choice([0, 1, 2, 2, 3, 4, 4]))
        kind = int(rng.integers(8))
        if psy == 3 and kind in (1, 3):
            kind = 0                                      # (covered by test_psy3_silence_and_impulse)
        jobs.append((fs, mode, kbps, psy, kind, int(rng.integers(1 << 30))))
    assert _against_oracle(M, jobs, 8, chunks=[1, 3, 4]) == []


def test_configuration_sweep_vs_oracle(M):
    """Every legal (sample rate, mode, bitrate) x psy model as ONE mixed batch, a different signal per stream, against the
    oracle byte for byte (SURVEY 8d cfg5 generalised: mixed configurations share a launch)."""
    rates = {48000: [(m, k) for m in "sjdm" for k in ((64, 96, 128, 160, 192, 256, 384) if m != "m" else (32, 48, 64, 96, 128, 192))],
             32000: [("s", 128), ("j", 192), ("m", 64), ("m", 96), ("d", 256)],
             24000: [("s", 64), ("j", 96), ("m", 32), ("m", 64), ("s", 128)],
             16000: [("m", 24), ("s", 48), ("j", 64)],
             44100: [("s", 128), ("j", 192), ("m", 64), ("d", 256), ("s", 384)],      # frames of two lengths (padding slots)
             22050: [("m", 32), ("s", 64), ("j", 160), ("m", 8)]}
    nframes, cfgs, pcms, refs = 5, [], [], []
    n = 0
    for fs, lst in rates.items():
        for mode, kbps in lst:
            for psy in (0, 1, 2, 3, 4):
                kind = (0, 3, 5, 7)[n % 4] if psy != 3 else (0, 5)[n % 2]      # psy 3 + silence-like kinds crash the reference
                pcm = gen_pcm(31 + n, kind, 0, nframes)
                n += 1
                try:
                    ref, _ = O.oracle_stream(pcm, samplerate=fs, mode=mode, kbps=kbps, psy=psy)
                except Exception:
                    continue                                                   # combination the reference rejects
                cfgs.append(M.StreamConfig(samplerate=fs, mode=mode, bitrate=kbps, psy_model=psy))
                pcms.append(pcm)
                refs.append(ref)
    assert len(cfgs) > 150
    b = M.Batch(cfgs)
    got, _ = b.encode(np.stack(pcms, axis=1))
    tail = b.flush()
    bad = [(c.samplerate, c.mode, c.bitrate, c.psy_model) for c, g, t, r in zip(cfgs, got, tail, refs) if g + t != r]
    assert not bad, bad
    b.close()


def _full_size(M, nstreams, nframes, psy, nbase, oracle_samples):
    base = np.stack([gen_pcm(s, 0, 0, nframes) for s in range(nbase)], axis=1)
    pcm = np.tile(base, (1, nstreams // nbase, 1, 1))
    b = M.Batch([M.StreamConfig(mode="s", psy_model=psy)] * nstreams)
    got, _ = b.encode(pcm)
    tail = b.flush()
    full = [g + t for g, t in zip(got, tail)]
    for s in range(nstreams):
        assert full[s] == full[s % nbase], s                          # identical inputs, identical frames, wherever the wave ran
        assert full[s][:2] == b"\xff\xfc" and len(full[s]) == nframes * 384
    for s in range(nbase):                                            # every distinct frame of the batch: CRC-16 recomputed from its bytes
        for f in range(nframes):
            assert _crc16_frame_ok(full[s][384 * f: 384 * (f + 1)]), (s, f)
    for s in oracle_samples:
        ref, _ = O.oracle_stream(pcm[:, s], mode="s", psy=psy)
        assert full[s] == ref
    b.close()


def test_baseline_configs4_mixed_batch(M):
    """BASELINE configs[4]: 32 kHz mono 64 kbps and 48 kHz stereo 192 kbps streams INTERLEAVED in one batch, psy model 4 (and
    the same with model 2, which the reference's own setter can select), against the oracle byte for byte; fed in two calls."""
    nstreams, nframes = 64, 10
    for psy in (4, 2):
        cfgs = [M.StreamConfig(samplerate=32000, mode="m", bitrate=64, psy_model=psy) if s % 2 == 0
                else M.StreamConfig(samplerate=48000, mode="s", bitrate=192, psy_model=psy) for s in range(nstreams)]
        pcm = np.stack([gen_pcm(7000 + s, (0, 7)[(s // 2) % 2], 0, nframes) for s in range(nstreams)], axis=1)
        b = M.Batch(cfgs)
        g1, _ = b.encode(pcm[:4])
        g2, _ = b.encode(pcm[4:])
        tail = b.flush()
        for s, c in enumerate(cfgs):
            ref, _ = O.oracle_stream(pcm[:, s], samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=psy)
            assert g1[s] + g2[s] + tail[s] == ref, (psy, s)
        b.close()


def test_one_stream_many_frames(M):
    """Frames of one stream are independent units for the kernels of models 1 and 3: ONE stream with 600 frames in one call
    (then 7 more in a second call, then the flush) equals the oracle, which encodes them one after the other."""
    nframes = 607
    for psy, mode, fs, kbps in ((1, "j", 48000, 128), (3, "s", 44100, 192)):
        pcm = gen_pcm(9000 + psy, 0, 0, nframes)[:, None]
        b = M.Batch([M.StreamConfig(samplerate=fs, mode=mode, bitrate=kbps, psy_model=psy)])
        g1, _ = b.encode(pcm[:600])
        g2, _ = b.encode(pcm[600:])
        ref, _ = O.oracle_stream(pcm[:, 0], samplerate=fs, mode=mode, kbps=kbps, psy=psy)
        assert g1[0] + g2[0] + b.flush()[0] == ref, (psy, mode, fs)
        b.close()


def test_baseline_configs0_2000_frames(M):
    """BASELINE configs[0] / SURVEY 8(d) cfg1 on the device: ONE stream, 2000 frames, seed 0, psycho_1, 's' and 'j', in one call
    (2000 independent units) -- every byte against the oracle (itself gated against the live reference on the same input,
    tests/test_oracle_golden.py)."""
    pcm = gen_pcm(0, 0, 0, 2000)[:, None]
    for mode in ("s", "j"):
        b = M.Batch([M.StreamConfig(mode=mode, psy_model=1)])
        got, _ = b.encode(pcm)
        tail = b.flush()
        b.close()
        assert got[0] + tail[0] == O.oracle_stream(pcm[:, 0], mode=mode, psy=1)[0], mode


def test_full_size_properties(M):
    """BASELINE configs[1] size (4096 streams, psy 1): size-independent properties instead of a full oracle run -- identical
    inputs give identical frames wherever they sit in the batch, every frame starts with the sync header and has the right
    length, the CRC-16 of every distinct frame is recomputed from its bytes, and a sample of streams equals the oracle."""
    _full_size(M, 4096, 3, 1, 64, (0, 17, 63))


def test_full_size_properties_configs2(M):
    """BASELINE configs[2] size: 16384 streams x psy 3, the same properties."""
    _full_size(M, 16384, 3, 3, 128, (0, 77, 127))


def test_full_population_configs3_on_one_device(M):
    """The WHOLE population of BASELINE configs[3] -- 131 072 streams, psy 3 -- on one device in one batch (the shape
    only bench.py had run): the same size-independent properties, three sampled streams against the oracle."""
    _full_size(M, 131072, 2, 3, 128, (0, 77, 127))


def test_full_population_configs4_share_on_one_device(M):
    """One GPU's share of BASELINE configs[4] at full size: 16 384 streams, 32 kHz mono 64 kbps / 48 kHz stereo 192 kbps interleaved,
    psy 4, two calls + flush.  Identical inputs give identical bytes wherever the wave ran (mono streams run two to a wave), every frame
    has its sync word and length, four sampled streams (both kinds, first and last block) equal the oracle."""
    nstreams, nframes, nbase = 16384, 3, 128
    cfgs = [M.StreamConfig(samplerate=32000, mode="m", bitrate=64, psy_model=4) if s % 2 == 0
            else M.StreamConfig(samplerate=48000, mode="s", bitrate=192, psy_model=4) for s in range(nstreams)]
    base = np.stack([gen_pcm(8200 + s, (0, 7)[(s // 2) % 2], 0, nframes) for s in range(nbase)], axis=1)
    pcm = np.tile(base, (1, nstreams // nbase, 1, 1))
    b = M.Batch(cfgs)
    g1, _ = b.encode(pcm[:1])
    g2, _ = b.encode(pcm[1:])
    tail = b.flush()
    b.close()
    full = [a + c + d for a, c, d in zip(g1, g2, tail)]
    for s in range(nstreams):
        assert full[s] == full[s % nbase], s
        fb = 288 if s % 2 == 0 else 576
        assert len(full[s]) == nframes * fb and all(full[s][fb * f: fb * f + 2] == b"\xff\xfc" for f in range(nframes)), s
    for s in (0, 1, nbase - 2, nbase - 1):
        c = cfgs[s]
        assert full[s] == O.oracle_stream(pcm[:, s], samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=4)[0], s


def test_psy3_silence_and_impulse_device_equals_emulation(M):
    """psy 3 on digital silence and on a lone impulse -- inputs the reference itself cannot encode (psycho_3.c:299 indexes
    with (int)(0.0/0.0)) and the commonest real input of a fleet of radio streams -- mixed into one batch with healthy
    streams and with psy 1 on the same signals: device bytes and every tap equal the host emulation of the same kernel
    source bit for bit (a NaN-to-int conversion or a -0.0 that behaves differently on gfx950 would show here), and the
    emulation equals the oracle's defined behaviour."""
    kinds = [1, 3, 0, 1, 3, 6, 1, 3, 7, 1]
    cfgs = [dict(psy=3, mode="s"), dict(psy=3, mode="s"), dict(psy=3, mode="s"), dict(psy=3, mode="j"), dict(psy=3, mode="j"),
            dict(psy=3, mode="j"), dict(psy=3, mode="m", kbps=64), dict(psy=3, mode="m", kbps=64, samplerate=24000), dict(psy=1, mode="j"),
            dict(psy=1, mode="s")]
    nframes = 8
    pcm = np.stack([gen_pcm(200 + s, kinds[s], 0, nframes) for s in range(len(cfgs))], axis=1)
    pcm[4:, 0] = gen_pcm(300, 0, 4, nframes - 4)                      # silence, then programme: the state carries over
    pcm[4:, 2] = 0                                                    # programme, then silence
    e = E.EmuBatch(cfgs)
    eg, et = e.encode(pcm, want_taps=True)
    b = M.Batch([M.StreamConfig(samplerate=c.get("samplerate", 48000), mode=c["mode"], bitrate=c.get("kbps", 128), psy_model=c["psy"]) for c in cfgs])
    dg, dt = b.encode(pcm, want_taps=True)
    assert dg == eg and b.flush() == e.flush()
    for name in ("sb_sample", "smr", "max_sc"):
        assert np.array_equal(dt[name].view(np.uint64), et[name].view(np.uint64)), name
    for name in ("subband", "scalar", "scalar_pre", "j_scale", "scfsi", "bit_alloc", "adb_left", "mode", "mode_ext", "crc16", "scfcrc"):
        assert np.array_equal(dt[name], et[name]), name
    assert np.isfinite(dt["smr"]).all()
    for s, c in enumerate(cfgs):
        ref, _ = O.oracle_stream(pcm[:, s], samplerate=c.get("samplerate", 48000), mode=c["mode"], kbps=c.get("kbps", 128), psy=c["psy"])
        assert dg[s] + b.flush()[s] == ref, s
    b.close()
    e.close()


def test_xpad_length_contract_on_device(M):
    """d_xpad_len outside 0 / 2..pad_len: the frame carries no PAD and nothing is written outside the stream's working set --
    the neighbours in the same workgroup keep producing the oracle's bytes (an over-long length used to reach
    negative word offsets in LDS)."""
    nf, ns = 4, 8
    cfgs = [M.StreamConfig(mode="j", psy_model=1, pad_len=20), M.StreamConfig(mode="s", psy_model=3, pad_len=0),
            M.StreamConfig(samplerate=24000, mode="m", bitrate=8, psy_model=1, pad_len=0), M.StreamConfig(mode="j", psy_model=1, pad_len=58)] * 2
    pcm = np.stack([gen_pcm(400 + s, 0, 0, nf) for s in range(ns)], axis=1)
    rng = np.random.default_rng(11)
    xp = rng.integers(0, 256, size=(nf, ns, E.TL_MAX_XPAD), dtype=np.uint8)
    xl = np.zeros((nf, ns), dtype=np.int32)
    xl[:, 0], xl[:, 1], xl[:, 2], xl[:, 3] = 58, 58, 58, 58           # only stream 3 (and 7, below) may carry 58 bytes
    xl[:, 4], xl[:, 5], xl[:, 6], xl[:, 7] = 100000, 2, 48, 58
    b = M.Batch(cfgs)
    got, _ = b.encode(pcm, xp, xl)
    tail = b.flush()
    for s, c in enumerate(cfgs):
        legal = 2 <= xl[0, s] <= c.pad_len
        e = O.OracleEncoder(samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=c.psy_model, pad_len=c.pad_len)
        want = b"".join(e.encode(pcm[i, s], (bytes(xp[i, s, :c.pad_len]) + bytes([c.pad_len])) if legal else None, c.pad_len if legal else 0)
                        for i in range(nf)) + e.finish()
        e.close()
        assert got[s] + tail[s] == want, s
    b.close()
    with pytest.raises(M.ToolameError):
        M.Batch([M.StreamConfig(samplerate=24000, mode="m", bitrate=8, pad_len=58)])


@pytest.mark.parametrize("path", [p for p in golden_cases() if not p.stem.startswith("p4_")], ids=lambda p: p.stem)
def test_legacy_abi_burst_cadence(M, path):
    """The nine reference symbols on EVERY golden case the legacy setters can express (psy 0..3; all rates, modes, bitrates,
    X-PAD): same call order as src/odr-audioenc.cpp:687-722, same bursty return lengths as the real reference (golden `lens`,
    bitstream.c:46-71), same bytes."""
    import ctypes as C
    g = np.load(path)
    fs, mode, kbps, psy, kind, seed, pad_len, nframes = (int(v) for v in g["cfg"])
    L = M.legacy_api()
    assert L.toolame_init() == 0
    assert L.toolame_set_samplerate(fs) == 0
    assert L.toolame_set_psy_model(psy) == 0
    assert L.toolame_set_channel_mode(bytes([mode])) == 0
    assert L.toolame_set_bitrate(kbps) == 0
    assert L.toolame_set_pad(pad_len) == 0
    assert L.toolame_set_psy_model(4) != 0 and L.toolame_set_channel_mode(b"x") != 0
    pcm = gen_pcm(seed, kind, 0, nframes)
    out = (C.c_ubyte * 4096)()
    chunks, lens = [], []
    for i in range(nframes):
        buf = np.ascontiguousarray(pcm[i])
        if "xpad" in g:
            xd = np.ascontiguousarray(g["xpad"][i], dtype=np.uint8)  # the reference's layout: padlen + 1 bytes
            n = L.toolame_encode_frame(buf.ctypes.data, xd.ctypes.data, int(g["xpad_len"][i]), out, 4096)
        else:
            n = L.toolame_encode_frame(buf.ctypes.data, None, 0, out, 4096)
        chunks.append(bytes(out[:n]))
        lens.append(n)
    n = L.toolame_finish(out, 4096)
    chunks.append(bytes(out[:n]))
    lens.append(n)
    assert lens == list(g["lens"])
    assert b"".join(chunks) == g["data"].tobytes()


@pytest.mark.parametrize("fs,mode,kbps,psy,nframes", [(48000, "j", 128, 1, 700), (22050, "m", 8, 1, 400), (24000, "s", 64, 3, 250),
                                                      (44100, "s", 384, 2, 60), (48000, "s", 192, 0, 300)])
def test_legacy_abi_deferred_launches(M, fs, mode, kbps, psy, nframes):
    """The shim files frames away on the calls that return nothing and encodes them in one launch when a burst is due
    (toolame_hip.hip, Legacy): over hundreds of frames -- many bursts, up to 78 deferred frames per launch at 8 kbps, frames of
    two lengths at 44.1 / 22.05 kHz, psy 2's chained state across deferred frames -- every call returns what the oracle's
    bit-buffer emulation returns (bitstream.c:46-71), byte for byte, and toolame_finish hands out the rest."""
    import ctypes as C
    L = M.legacy_api()
    assert L.toolame_init() == 0 and L.toolame_set_samplerate(fs) == 0 and L.toolame_set_psy_model(psy) == 0
    assert L.toolame_set_channel_mode(mode.encode()) == 0 and L.toolame_set_bitrate(kbps) == 0 and L.toolame_set_pad(0) == 0
    pcm = gen_pcm(77, 0, 0, nframes)
    ref, ref_lens = O.oracle_stream(pcm, samplerate=fs, mode=mode, kbps=kbps, psy=psy)
    out = (C.c_ubyte * 4096)()
    got, lens = [], []
    for i in range(nframes):
        n = L.toolame_encode_frame(np.ascontiguousarray(pcm[i]).ctypes.data, None, 0, out, 4096)
        got.append(bytes(out[:n]))
        lens.append(n)
    n = L.toolame_finish(out, 4096)
    got.append(bytes(out[:n]))
    lens.append(n)
    assert lens == ref_lens and b"".join(got) == ref
    assert sum(1 for x in lens[:-1] if x) >= 2                     # several bursts were crossed


def test_legacy_setters_validate(M):
    """toolame_set_bitrate refuses an illegal rate at the setter, for the MPEG version the sample rate selected
    (toolame.c:212-237 -> BitrateIndex, common.c:95-116), and a too-small output buffer truncates with a message
    (bitstream.c:54-58)."""
    L = M.legacy_api()
    assert L.toolame_init() == 0 and L.toolame_set_samplerate(48000) == 0 and L.toolame_set_channel_mode(b"s") == 0
    assert L.toolame_set_bitrate(100) != 0 and L.toolame_set_bitrate(144) != 0      # 144 is an LSF rate only
    assert L.toolame_set_bitrate(0) == 0 and L.toolame_set_bitrate(128) == 0
    assert L.toolame_set_samplerate(24000) == 0
    assert L.toolame_set_bitrate(384) != 0 and L.toolame_set_bitrate(144) == 0
    assert L.toolame_set_samplerate(11025) != 0 and L.toolame_set_pad(-2) != 0
    assert L.toolame_init() == 0


def test_errors(M):
    for kw in (dict(samplerate=11025), dict(samplerate=12345), dict(bitrate=100), dict(mode="x"), dict(psy_model=7),
               dict(pad_len=-1)):
        with pytest.raises(M.ToolameError):
            M.Batch([M.StreamConfig(**kw)])


def test_ingest_gain_peak_deinterleave(M):
    """the caller's gain / peak / de-interleave glue as a device pre-kernel, against the oracle
    (src/odr-audioenc.cpp:1030-1051,1139-1152), then straight into the encoder."""
    rng = np.random.default_rng(3)
    cfgs = [M.StreamConfig(mode="j"), M.StreamConfig(mode="m", bitrate=64), M.StreamConfig(mode="s"), M.StreamConfig(mode="m", bitrate=64)]
    gains = [0.0, -6.0, 3.5, 20.0]                       # +20 dB overflows int16: the reference wraps, so do we
    nf = 3
    raw = rng.integers(-32768, 32768, size=(nf, len(cfgs), 2304), dtype=np.int16)
    raw[0, 0, :8] = [32767, -32768, 0, 1, -1, 12345, -12345, 2]
    raw[2, 2] = 0                                        # digital silence -> peaks 0 (the caller's silence test)
    b = M.Batch(cfgs)
    for s, g in enumerate(gains):
        b.set_gain_db(g, s)
    pcm, peaks = b.ingest(raw)
    L = O.lib()
    for f in range(nf):
        for s, c in enumerate(cfgs):
            out = np.zeros((2, 1152), dtype=np.int16)
            pk = np.zeros(2, dtype=np.int16)
            src = np.ascontiguousarray(raw[f, s])
            L.mp2o_ingest(src.ctypes.data, 1 if c.mode == "m" else 2, gains[s], out.ctypes.data, pk.ctypes.data)
            assert np.array_equal(pcm[f, s], out), (f, s)
            assert np.array_equal(peaks[f, s], pk), (f, s)
    assert tuple(peaks[2, 2]) == (0, 0)
    # silence accounting of the caller (odr-audioenc.cpp:1053-1079) on these peaks plus a silent tail, against the oracle
    import ctypes as C
    pk2 = np.concatenate([peaks, np.zeros((5, len(cfgs), 2), dtype=np.int16)])
    pk2[5, 1] = (0, 3)                                     # one non-silent frame in the tail of stream 1 resets its counter
    ms = np.array([7, 0, 100, 0], dtype=np.uint32)
    want = ms.copy()
    for f in range(pk2.shape[0]):
        for s, c in enumerate(cfgs):
            want[s] = L.mp2o_silence_ms(int(want[s]), np.ascontiguousarray(pk2[f, s]).ctypes.data, 1 if c.mode == "m" else 2, c.samplerate)
    HL = M.load_library()
    HL.tlb_silence_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    assert HL.tlb_silence_host(b.h, np.ascontiguousarray(pk2).ctypes.data, pk2.shape[0], ms.ctypes.data) == 0
    assert np.array_equal(ms, want) and ms[2] == 24 * 6 and ms[1] == 24 * 2, (ms, want)
    got, _ = b.encode(pcm)
    tail = b.flush()
    for s, c in enumerate(cfgs):
        ref, _ = O.oracle_stream(pcm[:, s], mode=c.mode, kbps=c.bitrate, psy=c.psy_model)
        assert got[s] + tail[s] == ref
    b.close()


def test_zmq_frame_header(M):
    """SURVEY 8f N2 (ZeroMQ part): struct zmq_frame_header_t + frame, src/Outputs.h:76-89, Outputs.cpp:101-138."""
    import ctypes as C
    import struct
    cfgs = [M.StreamConfig(mode="j"), M.StreamConfig(mode="s", bitrate=192), M.StreamConfig(mode="m", bitrate=64)]
    b = M.Batch(cfgs)
    L = M.load_library()
    nf = 2
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(nf, 3, b.out_stride), dtype=np.uint8)
    peaks = rng.integers(-5, 32768, size=(nf, 3, 2)).astype(np.int16)
    ms = L.tlb_zmq_msg_stride(b.h)
    assert ms == 12 + b.out_stride
    msgs = np.zeros((nf, 3, ms), dtype=np.uint8)
    assert L.tlb_zmq_frame_host(b.h, frames.ctypes.data, peaks.ctypes.data, nf, msgs.ctypes.data) == 0
    for f in range(nf):
        for s in range(3):
            n = b.frame_bytes[s]
            want = struct.pack("<HHIhh", 1, 2, n, int(peaks[f, s, 0]), int(peaks[f, s, 1])) + frames[f, s, :n].tobytes()
            assert msgs[f, s, :12 + n].tobytes() == want, (f, s)
    b.close()


def test_edi_af_packets(M):
    """SURVEY 8f N2 (EDI part): AF packets through the C-ABI equal the golden vectors made with the reference's own
    TagItems/TagPacket/AFPacket classes (tests/golden/make_golden_edi.py), sender state included."""
    import edilib as E
    g = np.load(Path(__file__).resolve().parent / "golden" / "edi_cases.npz")
    kb = {96: 32, 288: 96, 384: 128, 576: 192}                  # frame bytes -> kbps at 48 kHz
    for name, *_ in E.CASES:
        frames, levels, fb, st = E.case_inputs(name)
        if name in E.CASE_STREAMS:                              # mixed 48 k / 24 k / 16 k: two or three 24-ms units per LSF frame
            b = M.Batch([M.StreamConfig(samplerate=r, mode=m, bitrate=k) for r, k, m in E.CASE_STREAMS[name]])
            assert b.max_upf == 3 and b.units_per_frame == [1, 2, 2, 3] and list(b.unit_bytes) == list(E.case_unit_bytes(name))
        else:
            b = M.Batch([M.StreamConfig(mode="s" if n > 96 else "m", bitrate=kb[int(n)]) for n in fb])
        assert list(b.frame_bytes) == [int(n) for n in fb] and b.out_stride == frames.shape[2]
        state = st.astype(M.EDI_STATE_DTYPE)
        pkts, plen = b.edi_af(frames, levels, state, E.VERSION)
        assert (plen == g[name + "_len"]).all(), name
        assert (pkts[:16] == g[name + "_head"]).all(), name
        assert E.digest(pkts, plen) == bytes(g[name + "_sha"]).hex(), name
        assert state.tobytes() == g[name + "_state"].tobytes(), name
        if name in E.CASE_STREAMS:                              # the ZeroMQ messages of the same batch: one per unit
            import struct
            msgs = b.zmq_frames(frames[:8], levels[:8])
            for f in range(8):
                for s2 in range(b.nstreams):
                    U, upf = b.unit_bytes[s2], b.units_per_frame[s2]
                    for u in range(b.max_upf):
                        m = msgs[f * b.max_upf + u, s2]
                        if u >= upf:
                            assert not m[:12].any(), (f, s2, u)
                            continue
                        want = struct.pack("<HHIhh", 1, 2, U, int(levels[f, s2, 0]), int(levels[f, s2, 1])) + frames[f, s2, u * U:(u + 1) * U].tobytes()
                        assert m[:12 + U].tobytes() == want, (f, s2, u)
        # two calls of half the frames == one call (the state carries everything)
        half = frames.shape[0] // 2
        state2 = st.astype(M.EDI_STATE_DTYPE)
        p1, l1 = b.edi_af(frames[:half], levels[:half] if levels is not None else None, state2, E.VERSION)
        p2, l2 = b.edi_af(frames[half:], levels[half:] if levels is not None else None, state2, E.VERSION)
        assert (np.concatenate([p1, p2]) == pkts).all() and (np.concatenate([l1, l2]) == plen).all(), name
        b.close()
    # tlb_edi_state_init == the first-call branch of EDI::write_frame
    for now, delay, tist in ((1700000000, 250, 1), (1600000123, 0, 0), (1751234567, 1015, 1)):
        assert M.edi_state_init(2, now, delay, tist, 37).tobytes() == E.init_state(2, now, delay, tist, 37).astype(M.EDI_STATE_DTYPE).tobytes()


EDI_TICK_CASES = {
    "48k": [(48000, "j", 128, 1), (48000, "s", 192, 3), (48000, "m", 64, 1), (48000, "s", 128, 0), (48000, "m", 32, 2), (48000, "j", 96, 4)],
    "mixed_lsf": [(48000, "s", 128, 1), (24000, "s", 64, 1), (24000, "m", 32, 3), (16000, "m", 24, 1)],      # E.CASE_STREAMS shape: 1, 2, 2, 3 units per frame
}


@pytest.mark.parametrize("case", sorted(EDI_TICK_CASES))
@pytest.mark.parametrize("egress,ngroups", [("af", 1), ("af", 3), ("pft", 2), ("frames", 2), ("zmq", 2)])
def test_tick_pipeline_equals_stage_by_stage(M, case, egress, ngroups):
    """tlb_tick_run -- PCIe in, ingest, encode, EDI AF (PFT), PCIe out as ONE call per tick, streams split into groups on three
    HIP streams -- against the same stages called one by one through the golden-pinned entry points (tlb_ingest_host,
    tlb_encode_host, tlb_edi_af_host, tlb_edi_pft_host): identical packets / fragments / frames tick by tick, gain applied,
    the version packet (ODRv) included, the last frame through tlb_tick_finish."""
    streams = EDI_TICK_CASES[case]
    cfgs = [M.StreamConfig(samplerate=r, mode=m, bitrate=k, psy_model=p) for r, m, k, p in streams]
    ns, T = len(cfgs), 14
    rng = np.random.default_rng(11)
    inter = np.stack([np.stack([gen_pcm(900 + s, (0, 7, 5, 4)[s % 4], 0, T)[f].T.reshape(-1) for s in range(ns)]) for f in range(T)])   # [T, ns, 2304] L R L R
    inter[3:6, 1] = 0                                    # three silent frames in stream 1: its silence counter runs and resets
    gains = [0.0, -3.0, 0.0, 6.0, 0.0, -9.5][:ns]
    version = b"odr-audioenc_amd tick"
    import ctypes as C
    kw = dict(now_s=1712345678, delay_ms=370, tist=True, tai_utc_offset=37)
    pft = dict(fec=2, chunk_len=207, transport=True, addr_source=4711, dest_port=12000)
    # ---- stage by stage
    b = M.Batch(cfgs)
    for s, g in enumerate(gains):
        b.set_gain_db(g, s)
    pcm, peaks = b.ingest(inter)
    lens = np.zeros((T, ns), dtype=np.int32)
    frames = np.zeros((T, ns, b.out_stride), dtype=np.uint8)
    assert b.L.tlb_encode_host_len(b.h, pcm.ctypes.data, T, None, None, frames.ctypes.data, lens.ctypes.data, None) == 0
    last = np.zeros((1, ns, b.out_stride), dtype=np.uint8)
    assert b.L.tlb_flush_host(b.h, last[0].ctypes.data) == 0
    # what leaves on tick f: the frame in output slot f (input frame f-1) with THIS tick's peaks; the finish sends the pending frame with the last peaks
    fr = np.concatenate([frames[1:], last])
    lv = np.concatenate([peaks[1:], peaks[-1:]])
    want_frames = [[fr[f, s, :b.frame_bytes[s]].tobytes() for s in range(ns)] for f in range(T)]
    if egress == "zmq":
        msgs = b.zmq_frames(fr, lv)
    elif egress != "frames":
        state = M.edi_state_init(ns, kw["now_s"], kw["delay_ms"], kw["tist"], kw["tai_utc_offset"])
        pk, pl = b.edi_af(fr, lv, state, version)
        if egress == "pft":
            pseq = np.zeros(ns, dtype=np.uint16)
            fg, fl, nf = b.edi_pft(pk, pl, pseq, **pft)
    # ---- the pipeline
    t = M.Tick(cfgs, egress=egress, ngroups=ngroups, version=version, **kw, **(pft if egress == "pft" else {}))
    for s, g in enumerate(gains):
        t.set_gain_db(g, s)
    OL = O.lib()
    OL.mp2o_silence_ms.restype = C.c_uint
    OL.mp2o_silence_ms.argtypes = [C.c_uint, C.c_void_p, C.c_int, C.c_long]
    silence = [0] * ns
    for f in range(T + 1):
        if f < T:
            t.pcm[:] = inter[f]
            t.run()
            assert np.array_equal(t.peaks, peaks[f])
            for s in range(ns):
                silence[s] = OL.mp2o_silence_ms(silence[s], np.ascontiguousarray(t.peaks[s]).ctypes.data, 1 if cfgs[s].mode == "m" else 2, cfgs[s].samplerate)
            assert list(t.silence_ms) == silence, f
            if f == 0:
                assert all(not t.packets(s) and not t.fragments(s) and not t.frame(s) and not t.messages(s) for s in range(ns))
                continue
        else:
            t.finish()
        k = f - 1
        for s in range(ns):
            if egress == "frames":
                assert t.frame(s) == want_frames[k][s], (f, s)
            elif egress == "zmq":
                U = b.unit_bytes[s]
                assert t.messages(s) == [msgs[k * b.max_upf + u, s, :12 + U].tobytes() for u in range(b.units_per_frame[s])], (f, s)
            elif egress == "af":
                want = [pk[k * b.max_upf + u, s, :pl[k * b.max_upf + u, s]].tobytes() for u in range(b.units_per_frame[s])]
                assert t.packets(s) == want, (f, s)
            else:
                want = [[fg[k * b.max_upf + u, s, i, :fl[k * b.max_upf + u, s, i]].tobytes() for i in range(nf[k * b.max_upf + u, s])] for u in range(b.units_per_frame[s])]
                assert t.fragments(s) == want, (f, s)
    t.close()
    b.close()


def test_tick_pipeline_with_xpad(M):
    """The tick's X-PAD side input (SURVEY 8f N3 through tlb_tick_xpad / tlb_tick_xpad_len): raw frames out of the pipeline
    equal the oracle's byte stream, frame by frame, for streams with different pad_len and a changing X-PAD length per tick."""
    cfgs = [M.StreamConfig(mode="j", bitrate=128, psy_model=1, pad_len=58), M.StreamConfig(mode="s", bitrate=192, psy_model=3, pad_len=34),
            M.StreamConfig(mode="m", bitrate=64, psy_model=1, pad_len=0),
            M.StreamConfig(samplerate=44100, mode="s", bitrate=128, psy_model=1, pad_len=0)]       # frames of two lengths (padding slots)
    T, ns = 10, len(cfgs)
    rng = np.random.default_rng(21)
    pcm = [gen_pcm(500 + s, 0, 0, T) for s in range(ns)]
    lens = [[58, 10, 2, 0, 34, 58, 58, 0, 20, 58], [34, 0, 2, 34, 8, 34, 0, 34, 34, 6], [0] * T, [0] * T]
    full = [[bytes(rng.integers(0, 256, cfgs[s].pad_len + 1, dtype=np.uint8)) for _ in range(T)] for s in range(ns)]   # the reference's xpad_data layout
    refs = []
    for s, c in enumerate(cfgs):
        e = O.OracleEncoder(samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=c.psy_model, pad_len=c.pad_len)
        out = b"".join(e.encode(pcm[s][f], full[s][f], lens[s][f]) for f in range(T)) + e.finish()
        e.close()
        refs.append(out)
    t = M.Tick(cfgs, egress="frames", ngroups=2, with_xpad=True)
    got = [b""] * ns
    for f in range(T):
        for s in range(ns):
            t.pcm[s] = pcm[s][f].T.reshape(-1) if cfgs[s].mode != "m" else np.concatenate([pcm[s][f][0], np.zeros(1152, dtype=np.int16)])
            t.xpad[s] = E.pack_xpad(full[s][f], lens[s][f], cfgs[s].pad_len)
            t.xpad_len[s] = lens[s][f]
        t.run()
        got = [g + t.frame(s) for s, g in enumerate(got)]
    t.finish()
    got = [g + t.frame(s) for s, g in enumerate(got)]
    t.close()
    assert got == refs


def test_mono_streams_share_waves_in_pairs_on_the_device(M):
    """Two mono streams of one configuration are encoded by ONE wave (csrc/mp2_wave.h tl_encode_pair; kernel variants <.., true>): pairs of
    every psy model, padded rates, LSF, X-PAD lengths that differ per stream and frame, odd streams left alone, stereo streams in
    between -- device == host emulation == oracle stream by stream, frames in ragged launches."""
    cfgs = ([dict(samplerate=48000, mode="m", kbps=64, psy=1)] * 5 + [dict(samplerate=48000, mode="j", kbps=128, psy=1)] +
            [dict(samplerate=24000, mode="m", kbps=32, psy=3)] * 3 + [dict(samplerate=44100, mode="m", kbps=64, psy=1)] * 2 +
            [dict(samplerate=48000, mode="m", kbps=96, psy=0)] * 2 + [dict(samplerate=32000, mode="m", kbps=64, psy=4)] * 3 +
            [dict(samplerate=48000, mode="m", kbps=128, psy=2)] * 2 + [dict(samplerate=22050, mode="m", kbps=32, psy=3)] * 2 +
            [dict(samplerate=48000, mode="m", kbps=80, psy=3, pad_len=24)] * 4 + [dict(samplerate=16000, mode="m", kbps=8, psy=1)] * 2 +
            [dict(samplerate=48000, mode="s", kbps=192, psy=0)] + [dict(samplerate=48000, mode="m", kbps=192, psy=1)] * 36)
    ns, nf = len(cfgs), 9
    rng = np.random.default_rng(31)
    pcm = np.stack([gen_pcm(3300 + s, (0, 7, 4, 5, 0, 2)[s % 6], 0, nf) for s in range(ns)], axis=1)
    xp = rng.integers(0, 256, size=(nf, ns, E.TL_MAX_XPAD), dtype=np.uint8)
    xl = np.zeros((nf, ns), dtype=np.int32)
    for s, c in enumerate(cfgs):
        if c.get("pad_len"):
            xl[:, s] = rng.choice([0, 2, 6, 24, 17], size=nf)
    def run(b):
        got, pos = [b""] * ns, 0
        for n in (1, 3, 2, nf - 6):
            g, _ = b.encode(pcm[pos:pos + n], xp[pos:pos + n], xl[pos:pos + n])
            got = [a + c for a, c in zip(got, g)]
            pos += n
        tail = b.flush()
        b.close()
        return [a + c for a, c in zip(got, tail)]
    dev = run(M.Batch([M.StreamConfig(samplerate=c["samplerate"], mode=c["mode"], bitrate=c["kbps"], psy_model=c["psy"], pad_len=c.get("pad_len", 0)) for c in cfgs]))
    emu = run(E.EmuBatch(cfgs))
    assert dev == emu
    for s in list(range(27)) + [ns - 1]:
        c = cfgs[s]
        e = O.OracleEncoder(**c)
        pl = c.get("pad_len", 0)
        want = b"".join(e.encode(pcm[f, s], bytes(pl - int(xl[f, s])) + bytes(xp[f, s, :int(xl[f, s])]) + bytes([int(xl[f, s])]), int(xl[f, s])) if pl
                        else e.encode(pcm[f, s]) for f in range(nf)) + e.finish()
        e.close()
        assert dev[s] == want, (s, c)


# ---- life cycle of ONE stream inside a live batch (tlb_stream_reset / _finish / _reconfigure; toolame.c:120-166 per stream) ----
_LIFE_POOL = [(48000, "s", 128, 1), (48000, "j", 192, 3), (48000, "m", 64, 1), (24000, "m", 64, 3), (48000, "s", 160, 2), (48000, "j", 128, 4),
              (48000, "d", 96, 0), (44100, "s", 128, 1), (22050, "m", 32, 3), (32000, "m", 64, 4), (48000, "s", 384, 1), (16000, "m", 24, 2)]


def _life_oracle(pcm_seg, cfg):
    return O.oracle_stream(pcm_seg, samplerate=cfg.samplerate, mode=cfg.mode, kbps=cfg.bitrate, psy=cfg.psy_model)[0]


def test_stream_life_cycle_in_a_live_batch(M):
    """64 streams of twelve configurations (all psy models, padded rates, LSF) run for 14 frames in ragged calls; between calls single
    streams are reset (toolame_init), finished (toolame_finish: the pending frame comes back) or reconfigured (new rate / mode /
    bitrate / model).  Every stream equals the oracle: the untouched ones over the whole run, restarted ones per life, each life an
    encoder of its own that starts with its first frame."""
    rng = np.random.default_rng(5)
    ns, T = 64, 14
    cfgs = [M.StreamConfig(samplerate=r, mode=m, bitrate=k, psy_model=p) for r, m, k, p in (_LIFE_POOL[s % len(_LIFE_POOL)] for s in range(ns))]
    pcm = np.stack([gen_pcm(4200 + s, (0, 7, 4, 0)[s % 4], 0, T) for s in range(ns)], axis=1)
    new_cfg = lambda q: M.StreamConfig(samplerate=q[0], mode=q[1], bitrate=q[2], psy_model=q[3])
    # frame index -> events applied BEFORE that frame is fed
    events = {3: [("reset", 5, None), ("finish", 17, None)],
              6: [("reconf", 9, new_cfg((48000, "s", 224, 2))),          # a configuration the batch has not seen
                  ("reconf", 30, new_cfg(_LIFE_POOL[1])),                # one it has
                  ("reset", 5, None)],                                   # the same stream again
              7: [("finish", 63, None), ("reconf", 40, new_cfg((24000, "s", 160, 4)))],
              11: [("reconf", 9, new_cfg((48000, "m", 192, 1))), ("finish", 0, None)]}
    b = M.Batch(cfgs)
    lives = {s: [dict(cfg=cfgs[s], start=0, got=b"", tail=None)] for s in range(ns)}      # per stream: its lives
    f = 0
    for n in (3, 1, 2, 1, 4, 3):                                                         # call boundaries = every event frame
        for kind, s, c in events.get(f, []):
            cur = lives[s][-1]
            cur["end"] = f
            if kind == "reset":
                b.stream_reset(s); cur["tail"] = None
            elif kind == "finish":
                cur["tail"] = b.stream_finish(s)
            else:
                cur["tail"] = None
                b.stream_reconfigure(s, c)
            lives[s].append(dict(cfg=c or cur["cfg"], start=f, got=b"", tail=None))
        got, _ = b.encode(pcm[f:f + n])
        for s in range(ns):
            lives[s][-1]["got"] += got[s]
        f += n
    assert f == T
    tail = b.flush()
    for s in range(ns):
        lives[s][-1]["end"] = T; lives[s][-1]["tail"] = tail[s]
        for life in lives[s]:
            if life["end"] == life["start"]:
                assert life["got"] == b"" and not life["tail"]
                continue
            want = _life_oracle(pcm[life["start"]:life["end"], s], life["cfg"])
            have = life["got"] + (life["tail"] or b"")
            if life["tail"] is None:                                    # reset / reconfigured away: the pending frame was dropped
                assert len(have) < len(want) and want.startswith(have) and len(want) - len(have) <= 1729, (s, life["start"])
            else:
                assert have == want, (s, life["start"], life["cfg"])
    # refused reconfigurations change nothing: a frame longer than the batch's stride, an illegal bitrate
    with pytest.raises(M.ToolameError):
        b.stream_reconfigure(1, M.StreamConfig(samplerate=16000, mode="s", bitrate=160, psy_model=1))     # 1440-byte frames > stride
    with pytest.raises(M.ToolameError):
        b.stream_reconfigure(1, M.StreamConfig(samplerate=48000, mode="s", bitrate=100, psy_model=1))
    # tlb_reset: the whole batch as new (scratch included) -- the same PCM gives the same bytes as the first life of every stream
    b.reset()
    cur = [lives[s][-1]["cfg"] for s in range(ns)]
    got, _ = b.encode(pcm[:5])
    tail = b.flush()
    for s in range(0, ns, 7):
        assert got[s] + tail[s] == _life_oracle(pcm[:5, s], cur[s]), s
    b.close()


@pytest.mark.parametrize("egress,ngroups", [("af", 3), ("pft", 2), ("frames", 1)])
def test_tick_submit_wait_equals_run(M, egress, ngroups):
    """Ticks overlapped (tlb_tick_submit / tlb_tick_wait: the next tick is queued on the second set of pinned buffers before this one is
    waited for, its copy-in running under this tick's kernels and copy-out) give, tick by tick, exactly what tlb_tick_run gives."""
    streams = EDI_TICK_CASES[sorted(EDI_TICK_CASES)[-1]]
    cfgs = [M.StreamConfig(samplerate=r, mode=m, bitrate=k, psy_model=p) for r, m, k, p in streams]
    ns, T = len(cfgs), 11
    inter = np.stack([np.stack([gen_pcm(1900 + s, (0, 7, 5, 4)[s % 4], 0, T)[f].T.reshape(-1) for s in range(ns)]) for f in range(T)])
    kw = dict(egress=egress, ngroups=ngroups, version=b"v1", now_s=1712345678, delay_ms=370, tist=True)
    if egress == "pft":
        kw.update(fec=1, chunk_len=207)
    snap = lambda t: [(t.frame(s), t.packets(s), t.fragments(s), tuple(t.peaks[s]), int(t.silence_ms[s])) for s in range(ns)]
    a = M.Tick(cfgs, **kw)
    want = []
    for f in range(T):
        a.pcm[:] = inter[f]
        a.run()
        want.append(snap(a))
    a.finish()
    want.append(snap(a))
    b = M.Tick(cfgs, **kw)
    got = []
    b.pcm[:] = inter[0]
    b.submit()
    for f in range(1, T):
        b.pcm[:] = inter[f]                                # the OTHER input set: filled while tick f - 1 is on its way
        b.submit()
        b.wait()
        got.append(snap(b))
    with pytest.raises(M.ToolameError):
        b.wait(); b.wait()                                 # nothing left to wait for after the last one
    got.append(snap(b))
    b.finish()
    got.append(snap(b))
    assert len(got) == len(want) == T + 1
    for f in range(T + 1):
        assert got[f] == want[f], f
    a.close(); b.close()


@pytest.mark.parametrize("egress", ["frames", "af"])
def test_tick_stream_life_cycle(M, egress):
    """The same three operations on a tick object, between runs: the restarted stream's slots stay empty until its next frame is
    final, its frames equal a fresh encoder's, nobody else's output moves, and the EDI sender of the stream keeps counting (SEQ has
    no gap and no repeat)."""
    cfgs = [M.StreamConfig(samplerate=r, mode=m, bitrate=k, psy_model=p) for r, m, k, p in
            [(48000, "s", 128, 1), (48000, "j", 128, 3), (24000, "m", 64, 1), (48000, "s", 192, 2), (48000, "m", 96, 4), (16000, "m", 32, 3)]]
    ns, T = len(cfgs), 12
    pcm = np.stack([gen_pcm(5100 + s, 0, 0, T) for s in range(ns)], axis=1)               # [T, ns, 2, 1152]
    inter = np.ascontiguousarray(pcm.transpose(0, 1, 3, 2)).reshape(T, ns, 2304)          # L R L R ... (mono: the first 1152 values are channel 0)
    for s, c in enumerate(cfgs):
        if c.mode == "m":
            inter[:, s, :1152] = pcm[:, s, 0]
    events = {4: ("reset", 1, None), 6: ("reconf", 3, M.StreamConfig(samplerate=48000, mode="s", bitrate=160, psy_model=1)), 8: ("finish", 4, None)}
    t = M.Tick(cfgs, egress=egress, ngroups=2, version=b"v", now_s=1712345678)
    lives = {s: [dict(cfg=cfgs[s], start=0, got=b"", tail=None)] for s in range(ns)}
    seqs = {s: [] for s in range(ns)}
    for f in range(T):
        if f in events:
            kind, s, c = events[f]
            cur = lives[s][-1]; cur["end"] = f
            if kind == "reset":
                t.stream_reset(s)
            elif kind == "finish":
                cur["tail"] = t.stream_finish(s)
            else:
                t.stream_reconfigure(s, c)
            lives[s].append(dict(cfg=c or cur["cfg"], start=f, got=b"", tail=None))
        t.pcm[:] = inter[f]
        t.run()
        for s in range(ns):
            if egress == "frames":
                lives[s][-1]["got"] += t.frame(s)
            else:
                for pk in t.packets(s):
                    seqs[s].append(int.from_bytes(pk[6:8], "big"))                          # AF header: "AF", LEN(4), SEQ(2)
                    unit = 3 * lives[s][-1]["cfg"].bitrate
                    at = pk.find(b"ss\x00\x01")                                            # TAG item ss0001: name(4) + length(4) + 3 bytes header + the unit
                    lives[s][-1]["got"] += pk[at + 11:at + 11 + unit]
            if f in events and events[f][1] == s:
                assert not t.frame(s) and not t.packets(s), (f, s)                          # the run right after the restart has nothing for it
    for s in range(ns):
        lives[s][-1]["end"] = T
        for life in lives[s]:
            want = _life_oracle(pcm[life["start"]:life["end"], s], life["cfg"])
            have = life["got"] + (life["tail"] or b"")
            assert want.startswith(have) and (life["tail"] is None or have == want), (s, life["start"])
            assert len(want) - len(have) <= 1729
        if egress == "af" and seqs[s]:
            assert seqs[s] == [(seqs[s][0] + i) & 0xffff for i in range(len(seqs[s]))], s
    t.close()


def test_long_runs_carry_no_drift(M):
    """3000 frames of one stream per model (psy 1 joint, psy 3, psy 2 and psy 4 with their chained prediction state, a padded
    44.1 kHz stream) in 7 calls of ragged length: equal to the oracle to the last byte -- nothing accumulates over 72 s of audio."""
    jobs = [(48000, "j", 128, 1), (48000, "s", 192, 3), (48000, "s", 160, 2), (32000, "m", 64, 4), (44100, "s", 128, 1)]
    nframes = 3000
    cfgs = [M.StreamConfig(samplerate=fs, mode=m, bitrate=k, psy_model=p) for fs, m, k, p in jobs]
    pcm = np.stack([gen_pcm(4200 + s, (0, 7, 0, 5, 0)[s], 0, nframes) for s in range(len(jobs))], axis=1)
    b = M.Batch(cfgs)
    out, pos = [b""] * len(jobs), 0
    for n in (1, 7, 300, 692, 1000, 999, 1):
        got, _ = b.encode(pcm[pos:pos + n])
        out = [x + y for x, y in zip(out, got)]
        pos += n
    assert pos == nframes
    tail = b.flush()
    b.close()
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(5) as ex:
        refs = list(ex.map(lambda s: O.oracle_stream(pcm[:, s], samplerate=jobs[s][0], mode=jobs[s][1], kbps=jobs[s][2], psy=jobs[s][3])[0], range(len(jobs))))
    for s in range(len(jobs)):
        assert out[s] + tail[s] == refs[s], jobs[s]


def test_edi_pft_fragments(M):
    """SURVEY 8f N2 (PFT part): fragments through the C-ABI equal the golden vectors (reference Reed-Solomon + CRC code under
    the restated PFT.cpp logic, tests/golden/make_golden_edi.py); Pseq carried across calls."""
    import edilib as E
    g = np.load(Path(__file__).resolve().parent / "golden" / "edi_pft_cases.npz")
    for name, *_ in E.PFT_CASES:
        af, af_len, pseq, kw = E.pft_case_inputs(name)
        b = M.Batch([M.StreamConfig()] * af.shape[1])               # the PFT layer only needs the stream count
        ps = pseq.copy()
        frags, flen, nfrag = b.edi_pft(af, af_len, ps, **kw)
        assert (nfrag == g[name + "_n"]).all() and (flen == g[name + "_len"]).all(), name
        assert (frags[:2] == g[name + "_head"]).all(), name
        assert E.pft_digest(frags, flen, nfrag) == bytes(g[name + "_sha"]).hex(), name
        assert (ps == g[name + "_pseq"]).all(), name
        half = af.shape[0] // 2
        ps2 = pseq.copy()
        f1, l1, n1 = b.edi_pft(af[:half], af_len[:half], ps2, **kw)
        f2, l2, n2 = b.edi_pft(af[half:], af_len[half:], ps2, **kw)
        assert (np.concatenate([f1, f2]) == frags).all() and (np.concatenate([n1, n2]) == nfrag).all() and (ps2 == ps).all(), name
        # receiver side, on the DEVICE's fragments: `fec` of them lost, the reference's decode_rs_char.c corrects the erasures, the
        # AF packet (LEN, CRC) comes back bit-exactly
        if E.pft_ref_lib() is not None:
            E.check_reassembly(af, af_len, frags, flen, nfrag, kw["fec"])
        b.close()


def test_bench_two_ranks_on_the_gpu():
    """The multi-rank path of bench.py on real hardware (BASELINE configs[3]): `--gpus 2` starts two fresh rank
    processes that share this box's one GPU (gloo carries the barriers and the gather; with one GPU per rank it is RCCL), each
    encodes its own 16384 psy-3 streams, rank 0 prints the line.  What an 8-GPU driver run exercises has then already run on
    a GPU: process start before any GPU call, sharding, barriers, max over ranks, gather, the post-run oracle check per rank."""
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--no-also",
                        "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["world_size_observed"] == 2 and line["scaling"] == "weak"
    assert len(line["per_gpu_frames_per_s"]) == 2 and all(v > 0 for v in line["per_gpu_frames_per_s"])
    assert line["config"]["baseline_config"] == 3 and "configs[3]" in line["config"]["workload"] and line["config"]["streams_per_gpu"] == 16384
    assert line["value"] > 0 and line["steps"] == 3 and line["output_check"]["checked"] and line["output_check"]["per_rank_ok"] == [True, True]
    assert line["roofline"]["hbm"]["achieved"] > 0 and line["roofline"]["bound"] == "valu_fp64" and 0 < line["roofline"]["frac"] < 1
    # the line names the device of every rank, gathered over the collective -- here two ranks on ONE GPU, and it says so
    dev = line["devices_observed"]
    assert [d["rank"] for d in dev] == [0, 1] and all(d.get("name") and (d.get("uuid") or d.get("pci_bus_id")) for d in dev), dev
    assert line["distinct_devices_observed"] == 1 and len({d["pid"] for d in dev}) == 2


def test_bench_configs4_two_ranks_on_the_gpu():
    """BASELINE configs[4] as a multi-rank run (`bench.py --gpus 2 --config 4`): every rank encodes its own interleaved 32 kHz mono /
    48 kHz stereo batch with psy 4 (`value`) and with psy 2 (`also.configs4_psy2`), the oracle checks a mono and a stereo stream on
    EVERY rank, the line has per-GPU and aggregate rates for both models."""
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--config", "4", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--streams", "4096", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["world_size_observed"] == 2 and line["config"]["baseline_config"] == 4 and "configs[4]" in line["config"]["workload"]
    assert len(line["per_gpu_frames_per_s"]) == 2 and all(v > 0 for v in line["per_gpu_frames_per_s"]) and line["value"] > 0
    assert line["output_check"]["checked"] and line["output_check"]["per_rank_ok"] == [True, True]
    sib = line["also"]["configs4_psy2"]
    assert sib["value"] > 0 and len(sib["per_gpu_frames_per_s"]) == 2 and sib["output_check"]["per_rank_ok"] == [True, True]


def test_this_hosts_libm_is_the_one_tl_libm_restates(tmp_path):
    """The oracle on THIS box links this box's libm; csrc/tl_libm.h restates glibc 2.35's FMA-path routines.  Same sweep as
    tests/test_libm_agree.py (CPU suite, build container), run where the GPU tests run: 2 M arguments per function, 0 differ."""
    import subprocess
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "libm_agree"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-pthread", "-I", str(root / "odr-audioenc_amd" / "csrc"),
                    str(root / "tools" / "libm_agree.cpp"), "-o", str(exe), "-lm"], check=True)
    r = subprocess.run([str(exe), "2", "8", "11"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
