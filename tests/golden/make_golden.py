#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference
(oracle/_ref/libtoolame_ref.so, built by oracle/Makefile from /root/reference/libtoolame-dab/*.c)
on seeded integer PCM (tests/pcmgen.py).  Runs only in the build container; the .npz files are
committed and are the fixtures every parity test (oracle, HIP path) is anchored on.

Each case file holds: the configuration, the PCM seed/kind (PCM itself is regenerated), the
reference's output bytes (all toolame_encode_frame calls + toolame_finish), the per-call return
lengths (pins the 4096-byte burst cadence), and per-frame stage taps read from the
reference's own statics (toolame.c:96-115): scalar, scfsi, bit_alloc, smr, max_sc, mode/mode_ext,
plus sb_sample / quantised subband samples for selected frames.
"""
import ctypes as C
import pickle
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import oraclelib as O  # noqa: E402
from pcmgen import gen_pcm  # noqa: E402

NFRAMES = 16
BIG_TAP_FRAMES = (0, 5)

CONFIGS = [  # (samplerate, mode, kbps)
    (48000, "s", 128), (48000, "j", 128), (48000, "s", 192), (32000, "m", 64), (24000, "m", 64),
    (48000, "s", 96), (48000, "j", 192), (48000, "d", 128), (24000, "j", 64), (48000, "m", 64),
]


def cases():
    out = []
    for psy in (0, 1, 2, 3, 4):      # 4 = psycho_4, selected by writing the reference's `model` (see tests/oraclelib.py)
        for i, (fs, mode, kbps) in enumerate(CONFIGS):
            out.append(dict(name=f"p{psy}_{fs // 1000}k_{mode}_{kbps}_k0", samplerate=fs, mode=mode, kbps=kbps,
                            psy=psy, kind=0, seed=100 + i, pad_len=0))
    for psy in (1, 2, 3, 4):
        for mode in ("s", "j"):
            for kind in range(1, 8):
                if psy == 3 and kind in (1, 3):
                    continue  # the reference segfaults on digital silence with psy 3 (psycho_3.c:299)
                out.append(dict(name=f"p{psy}_48k_{mode}_128_k{kind}", samplerate=48000, mode=mode, kbps=128,
                                psy=psy, kind=kind, seed=7 + kind, pad_len=0))
    # 44.1 / 22.05 kHz: frames of two lengths (padding slots, availbits.c:49-62) -- the library encodes them, DAB does not use them
    for psy, fs, mode, kbps in ((1, 44100, "s", 128), (1, 44100, "j", 192), (3, 44100, "s", 128), (0, 44100, "m", 64), (2, 44100, "s", 128),
                                (1, 22050, "m", 32), (1, 22050, "s", 64), (3, 22050, "j", 96), (2, 22050, "m", 32)):
        out.append(dict(name=f"p{psy}_{fs // 1000}k_{mode}_{kbps}_k0", samplerate=fs, mode=mode, kbps=kbps, psy=psy, kind=0,
                        seed=300 + psy + kbps, pad_len=0))
    # round 6: the lowest LSF rate (16 kHz, sampling-frequency index 2 of common.c:118-144) end to end, and dual channel below 48 kHz
    for psy, fs, mode, kbps in ((1, 16000, "m", 32), (3, 16000, "s", 64), (2, 16000, "j", 48), (1, 24000, "d", 96), (4, 16000, "d", 80)):
        out.append(dict(name=f"p{psy}_{fs // 1000}k_{mode}_{kbps}_k0", samplerate=fs, mode=mode, kbps=kbps, psy=psy, kind=0,
                        seed=600 + psy + kbps, pad_len=0))
    out.append(dict(name="p1_48k_j_128_xpad", samplerate=48000, mode="j", kbps=128, psy=1, kind=0, seed=42, pad_len=58))
    out.append(dict(name="p3_48k_s_192_xpad", samplerate=48000, mode="s", kbps=192, psy=3, kind=0, seed=43, pad_len=58))
    # round 5: the DAB maximum (196 bytes), the largest length the caller accepts (255, src/odr-audioenc.cpp:566), and a small frame
    # whose PAD eats most of its bit budget (48 kHz mono 32 kbps: 96-byte frames, 80 of them PAD; `adb` of toolame.c:301)
    out.append(dict(name="p1_48k_s_192_xpad196", samplerate=48000, mode="s", kbps=192, psy=1, kind=0, seed=44, pad_len=196,
                    lens=[196, 10, 2, 0, 100, 196]))
    out.append(dict(name="p3_48k_j_192_xpad255", samplerate=48000, mode="j", kbps=192, psy=3, kind=0, seed=45, pad_len=255,
                    lens=[255, 201, 2, 0, 230, 255]))
    out.append(dict(name="p1_48k_m_32_xpad80", samplerate=48000, mode="m", kbps=32, psy=1, kind=0, seed=46, pad_len=80,
                    lens=[80, 10, 2, 0, 34, 80]))
    out.append(dict(name="p2_24k_m_32_xpad40", samplerate=24000, mode="m", kbps=32, psy=2, kind=0, seed=47, pad_len=40,
                    lens=[40, 10, 2, 0, 34, 40]))
    return out


def xpads_for(case):
    if not case["pad_len"]:
        return None
    rng = np.random.default_rng(case["seed"])
    lens = case.get("lens", [58, 10, 2, 0, 34, 58])
    return [(bytes(rng.integers(0, 256, case["pad_len"] + 1, dtype=np.uint8)), lens[i % 6]) for i in range(NFRAMES)]


_TABLE_CHILD = r"""
import ctypes as C, sys, numpy as np, pickle
L = C.CDLL(sys.argv[1])
L.toolame_set_samplerate.argtypes = [C.c_long]; L.toolame_set_channel_mode.argtypes = [C.c_char]
L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.toolame_init(); L.toolame_set_samplerate(48000); L.toolame_set_psy_model(int(sys.argv[3]))
L.toolame_set_channel_mode(b's'); L.toolame_set_bitrate(128); L.toolame_set_pad(0)
pcm = ((np.arange(2304) * 7919) % 2001 - 1000).astype(np.int16).reshape(2,1152); out = (C.c_ubyte*4096)()
L.toolame_encode_frame(pcm.ctypes.data, None, 0, out, 4096)
g = lambda name, ct, n: np.ctypeslib.as_array((ct*n).in_dll(L, name)).copy()
d = dict(enwindow=g('enwindow', C.c_double, 512), scalefactor=g('scalefactor', C.c_double, 64),
         multiple=g('multiple', C.c_double, 64))
m = (C.c_double*512)(); L.create_dct_matrix(m); d['dct'] = np.ctypeslib.as_array(m).copy()
if sys.argv[3] == '1':
    d['dbtable'] = g('dbtable', C.c_double, 1000)
else:
    d['p3_bark'] = g('bark', C.c_double, 513); d['p3_ath'] = g('ath', C.c_double, 513)
    nb = C.c_int.in_dll(L, 'cbands').value
    d['p3_cbidx'] = g('cbandindex', C.c_int, 32)[:nb+1]; d['p3_subset'] = g('freq_subset', C.c_int, 136)
pickle.dump(d, open(sys.argv[2], 'wb'))
"""


# Every table the product shares with the oracle through csrc/mp2_tables.inc, read out of the REFERENCE's memory after its own
# init code ran (oracle/Makefile TABLE_TAPS makes the statics visible): psy-1 threshold / critical-band tables and the psy-2
# absolute threshold per sample rate, the allocation tables once.
_RATE_CHILD = r"""
import ctypes as C, sys, numpy as np, pickle
L = C.CDLL(sys.argv[1]); fs = int(sys.argv[3]); psy = int(sys.argv[4])
L.toolame_set_samplerate.argtypes = [C.c_long]; L.toolame_set_channel_mode.argtypes = [C.c_char]
L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.toolame_init(); L.toolame_set_samplerate(fs); L.toolame_set_psy_model(min(psy, 3))
if psy > 3: C.c_int.in_dll(L, 'tlref_model').value = psy      # the setter refuses model 4 (toolame.c:204-207); `model` is exposed by oracle/Makefile
L.toolame_set_channel_mode(b's'); L.toolame_set_bitrate(128 if fs >= 32000 else 64); L.toolame_set_pad(0)
pcm = ((np.arange(2304) * 7919) % 2001 - 1000).astype(np.int16).reshape(2,1152); out = (C.c_ubyte*4096)()
L.toolame_encode_frame(pcm.ctypes.data, None, 0, out, 4096)
d = {}
if psy == 1:
    n = C.c_int.in_dll(L, 'sub_size').value; ncb = C.c_int.in_dll(L, 'crit_band').value
    class G(C.Structure): _fields_ = [('line', C.c_int), ('bark', C.c_double), ('hear', C.c_double), ('x', C.c_double)]
    ltg = C.cast(C.c_void_p.in_dll(L, 'tlref_tab_ltg').value, C.POINTER(G))
    d['p1_line'] = np.array([ltg[i].line for i in range(1, n)], dtype=np.int32)
    d['p1_bark'] = np.array([ltg[i].bark for i in range(1, n)]); d['p1_hear'] = np.array([ltg[i].hear for i in range(1, n)])
    cb = C.cast(C.c_void_p.in_dll(L, 'cbound').value, C.POINTER(C.c_int))
    d['p1_cbound'] = np.array([cb[i] for i in range(ncb)], dtype=np.int32)
    g = lambda name, ct, n: np.ctypeslib.as_array((ct*n).in_dll(L, 'tlref_tab_' + name)).copy()
    d.update(alloc_snr=g('SNR', C.c_double, 18), alloc_bits=g('bits', C.c_int, 18), alloc_group=g('group', C.c_int, 18),
             alloc_steps=g('steps', C.c_int, 18), alloc_steps2n=g('steps2n', C.c_int, 18), alloc_nbal=g('nbal', C.c_int, 9),
             alloc_table_sblimit=g('table_sblimit', C.c_int, 5), alloc_step_index=g('step_index', C.c_int, 144), alloc_line=g('line', C.c_int, 160))
else:
    # the DERIVED tables of psy 2 / psy 4: file-scope pointers the init code filled (psycho_2.c:337-403, psycho_4.c:330-413)
    pre = 'p2' if psy == 2 else 'p4'
    def through(name, ct, n):
        p = C.cast(C.c_void_p.in_dll(L, 'tlref_tab_%s_%s' % (pre, name)).value, C.POINTER(ct))
        return np.array([p[i] for i in range(n)], dtype=np.int32 if ct is C.c_int else np.float64)
    if psy == 2:
        a = C.cast(C.c_void_p.in_dll(L, 'tlref_tab_absthr').value, C.POINTER(C.c_double))
        d['p2_absthr'] = np.array([a[i] for i in range(513)])
        d['p2_bmax'] = np.ctypeslib.as_array((C.c_double * 27).in_dll(L, 'tlref_tab_p2_bmax')).copy()
    else:
        d['p4_minval'] = np.ctypeslib.as_array((C.c_double * 27).in_dll(L, 'tlref_tab_p4_minval')).copy()
        d['p4_ath'] = through('ath', C.c_double, 513); d['p4_bark'] = through('bark', C.c_double, 513)
    d[pre + '_partition'] = through('partition', C.c_int, 513); d[pre + '_numlines'] = through('numlines', C.c_int, 64)
    d[pre + '_cbval'] = through('cbval', C.c_double, 64); d[pre + '_rnorm'] = through('rnorm', C.c_double, 64)
    d[pre + '_tmn'] = through('tmn', C.c_double, 64); d[pre + '_s'] = through('s', C.c_double, 64 * 64).reshape(64, 64)
    d[pre + '_window'] = through('window', C.c_double, 1024)
pickle.dump(d, open(sys.argv[2], 'wb'))
"""
RATES = (48000, 44100, 32000, 24000, 22050, 16000)


def make_rate_tables():
    d = {}
    for fs in RATES:
        for psy in (1, 2, 4):
            tmp = HERE / f"_rt{fs}_{psy}.pkl"
            subprocess.run([sys.executable, "-c", _RATE_CHILD, str(O.REF_SO), str(tmp), str(fs), str(psy)], check=True, stderr=subprocess.DEVNULL)
            for k, v in pickle.load(open(tmp, "rb")).items():
                if k.startswith("alloc_") or k in ("p2_bmax", "p4_minval"):
                    assert k not in d or np.array_equal(d[k], v)
                    d[k] = v
                else:
                    d[f"{k}_{fs}"] = v
            tmp.unlink()
    np.savez_compressed(HERE / "tables_rates.npz", **d)


def make_tables():
    d = {}
    for psy in ("1", "3"):
        tmp = HERE / f"_tables{psy}.pkl"
        subprocess.run([sys.executable, "-c", _TABLE_CHILD, str(O.REF_SO), str(tmp), psy], check=True,
                       stderr=subprocess.DEVNULL)
        d.update(pickle.load(open(tmp, "rb")))
        tmp.unlink()
    np.savez_compressed(HERE / "tables_48k.npz", **d)


def main():
    if not O.REF_SO.exists():
        subprocess.run(["make", "-C", str(O.ORACLE_DIR), "ref"], check=True)
    if "--no-tables" not in sys.argv:
        make_tables()
        make_rate_tables()
    if "--tables-only" in sys.argv:
        return
    total = 0
    for case in cases():
        if "--only-missing" in sys.argv and (HERE / (case["name"] + ".npz")).exists():
            continue
        pcm = gen_pcm(case["seed"], case["kind"], 0, NFRAMES)
        xp = xpads_for(case)
        ref = O.reference_stream(pcm, samplerate=case["samplerate"], mode=case["mode"], kbps=case["kbps"],
                                 psy=case["psy"], pad_len=case["pad_len"], xpads=xp, tap_frames=range(NFRAMES))
        assert ref["rc"] == [0] * 6, ref["rc"]
        t = ref["taps"]
        arrs = dict(
            data=np.frombuffer(ref["data"], dtype=np.uint8), lens=np.array(ref["lens"], dtype=np.int32),
            scalar=np.stack([t[i]["scalar"] for i in range(NFRAMES)]).astype(np.uint8),
            j_scale=np.stack([t[i]["j_scale"] for i in range(NFRAMES)]).astype(np.uint8),
            scfsi=np.stack([t[i]["scfsi"] for i in range(NFRAMES)]).astype(np.uint8),
            bit_alloc=np.stack([t[i]["bit_alloc"] for i in range(NFRAMES)]).astype(np.uint8),
            smr=np.stack([t[i]["smr"] for i in range(NFRAMES)]),
            max_sc=np.stack([t[i]["max_sc"] for i in range(NFRAMES)]),
            mode=np.array([t[i]["mode"] for i in range(NFRAMES)], dtype=np.int8),
            mode_ext=np.array([t[i]["mode_ext"] for i in range(NFRAMES)], dtype=np.int8),
            cfg=np.array([case["samplerate"], ord(case["mode"]), case["kbps"], case["psy"], case["kind"],
                          case["seed"], case["pad_len"], NFRAMES], dtype=np.int64),
        )
        if case["psy"] == 1 and case["kind"] == 0:      # the 18 KB/frame taps only where they add coverage
            arrs["sb_sample"] = np.stack([t[i]["sb_sample"] for i in BIG_TAP_FRAMES])
            arrs["subband"] = np.stack([t[i]["subband"] for i in BIG_TAP_FRAMES]).astype(np.uint16)
            arrs["big_tap_frames"] = np.array(BIG_TAP_FRAMES, dtype=np.int32)
        if xp is not None:
            arrs["xpad"] = np.stack([np.frombuffer(x[0], dtype=np.uint8) for x in xp])
            arrs["xpad_len"] = np.array([x[1] for x in xp], dtype=np.int32)
        f = HERE / (case["name"] + ".npz")
        np.savez_compressed(f, **arrs)
        total += f.stat().st_size
    print(len(cases()), "cases,", total // 1024, "KiB")


if __name__ == "__main__":
    main()
