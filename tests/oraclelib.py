"""ctypes bindings for the TEST-ONLY oracle (oracle/libmp2oracle.so) and, when built, the real
reference (oracle/_ref/libtoolame_ref.so).  Never imported by the product package."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"


class Taps(C.Structure):
    _fields_ = [
        ("sb_sample", C.c_double * (2 * 3 * 12 * 32)),
        ("j_sample", C.c_double * (3 * 12 * 32)),
        ("scalar_pre", C.c_uint * (2 * 3 * 32)),
        ("scalar", C.c_uint * (2 * 3 * 32)),
        ("j_scale", C.c_uint * (3 * 32)),
        ("max_sc", C.c_double * (2 * 32)),
        ("smr", C.c_double * (2 * 32)),
        ("scfsi", C.c_uint * (2 * 32)),
        ("bit_alloc", C.c_uint * (2 * 32)),
        ("subband", C.c_uint * (2 * 3 * 12 * 32)),
        ("adb_left", C.c_int),
        ("mode", C.c_int), ("mode_ext", C.c_int), ("jsbound", C.c_int),
        ("crc16", C.c_uint),
        ("scfcrc", C.c_ubyte * 4),
    ]


_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR), "libmp2oracle.so"], check=True)


def lib():
    global _lib
    if _lib is None:
        so = ORACLE_DIR / "libmp2oracle.so"
        src = ORACLE_DIR / "mp2_oracle.c"
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            build_oracle()
        L = C.CDLL(str(so))
        L.mp2o_create.restype = C.c_void_p
        L.mp2o_create.argtypes = [C.c_long, C.c_char, C.c_int, C.c_int, C.c_int]
        L.mp2o_destroy.argtypes = [C.c_void_p]
        for f in ("mp2o_frame_bytes", "mp2o_nch", "mp2o_sblimit", "mp2o_tablenum", "mp2o_dab_extension"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.mp2o_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.mp2o_encode_frame.restype = C.c_int
        L.mp2o_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.mp2o_finish.restype = C.c_int
        L.mp2o_get_taps.argtypes = [C.c_void_p]
        L.mp2o_get_taps.restype = C.POINTER(Taps)
        L.mp2o_gen_pcm.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.mp2o_fht1024.argtypes = [C.c_void_p]
        L.mp2o_filterbank_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mp2o_ingest.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.mp2o_silence_ms.argtypes = [C.c_uint, C.c_void_p, C.c_int, C.c_long]
        L.mp2o_silence_ms.restype = C.c_uint
        L.mp2o_bench_stream.argtypes = [C.c_long, C.c_char, C.c_int, C.c_int, C.c_uint32, C.c_int]
        L.mp2o_bench_stream.restype = C.c_long
        _lib = L
    return _lib


TAP_SHAPES = {
    "sb_sample": (2, 3, 12, 32), "j_sample": (3, 12, 32), "scalar_pre": (2, 3, 32), "scalar": (2, 3, 32),
    "j_scale": (3, 32), "max_sc": (2, 32), "smr": (2, 32), "scfsi": (2, 32), "bit_alloc": (2, 32),
    "subband": (2, 3, 12, 32),
}


def taps_to_dict(t):
    d = {k: np.ctypeslib.as_array(getattr(t, k)).reshape(s).copy() for k, s in TAP_SHAPES.items()}
    for k in ("adb_left", "mode", "mode_ext", "jsbound", "crc16"):
        d[k] = int(getattr(t, k))
    d["scfcrc"] = bytes(t.scfcrc)
    return d


class OracleEncoder:
    """One stream of the oracle with toolame_encode_frame() semantics."""

    def __init__(self, samplerate=48000, mode="s", kbps=128, psy=1, pad_len=0):
        self.L = lib()
        self.h = self.L.mp2o_create(samplerate, mode.encode()[0:1], kbps, psy, pad_len)
        if not self.h:
            raise ValueError("illegal configuration")
        self.frame_bytes = self.L.mp2o_frame_bytes(self.h)
        self.nch = self.L.mp2o_nch(self.h)
        self.sblimit = self.L.mp2o_sblimit(self.h)
        self.pad_len = pad_len

    def encode(self, pcm, xpad=None, xpad_len=0):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        assert pcm.shape == (2, 1152)
        out = (C.c_ubyte * 4096)()
        xp = None
        if xpad is not None:
            xp = (C.c_ubyte * len(xpad)).from_buffer_copy(bytes(xpad))
        n = self.L.mp2o_encode_frame(self.h, pcm.ctypes.data, xp, xpad_len, out, 4096)
        return bytes(out[:n])

    def finish(self):
        out = (C.c_ubyte * 4096)()
        n = self.L.mp2o_finish(self.h, out, 4096)
        return bytes(out[:n])

    def taps(self):
        return taps_to_dict(self.L.mp2o_get_taps(self.h).contents)

    def close(self):
        if self.h:
            self.L.mp2o_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def oracle_stream(pcm_frames, **cfg):
    """Encode [n,2,1152] int16 -> (all bytes incl. finish, per-call return lengths)."""
    e = OracleEncoder(**cfg)
    chunks, lens = [], []
    for f in pcm_frames:
        b = e.encode(f)
        chunks.append(b)
        lens.append(len(b))
    b = e.finish()
    chunks.append(b)
    lens.append(len(b))
    e.close()
    return b"".join(chunks), lens


# ---------------------------------------------------------------------------------------------
# the real reference (process-global singleton: one configuration per process -> run in a child)
REF_SO = ORACLE_DIR / "_ref" / "libtoolame_ref.so"

_REF_CHILD = r"""
import ctypes as C, sys, pickle, numpy as np
so, cfg_pkl = sys.argv[1], sys.argv[2]
cfg = pickle.load(open(cfg_pkl, 'rb'))
L = C.CDLL(so)
L.toolame_set_samplerate.argtypes = [C.c_long]
L.toolame_set_channel_mode.argtypes = [C.c_char]
L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.toolame_finish.argtypes = [C.c_void_p, C.c_size_t]
rc = [L.toolame_init(), L.toolame_set_samplerate(cfg['samplerate']), L.toolame_set_psy_model(min(cfg['psy'], 3)),
      L.toolame_set_channel_mode(cfg['mode'].encode()[0:1]), L.toolame_set_bitrate(cfg['kbps']),
      L.toolame_set_pad(cfg['pad_len'])]
if cfg['psy'] > 3:   # the setter refuses model 4 (toolame.c:204-207); the file-scope `model` is exposed by oracle/Makefile
    C.c_int.in_dll(L, 'tlref_model').value = cfg['psy']
def tap(name, ctype, shape, deref=False):
    n = int(np.prod(shape))
    if deref:
        p = C.c_void_p.in_dll(L, 'tlref_' + name).value
        arr = (ctype * n).from_address(p)
    else:
        arr = (ctype * n).in_dll(L, 'tlref_' + name)
    return np.ctypeslib.as_array(arr).reshape(shape).copy()
pcm = cfg['pcm']; xpads = cfg.get('xpads'); taps_for = set(cfg['tap_frames'])
out = (C.c_ubyte * 4096)()
chunks, lens, taps = [], [], {}
for i in range(pcm.shape[0]):
    buf = np.ascontiguousarray(pcm[i])
    if xpads is not None:
        xp = (C.c_ubyte * len(xpads[i][0])).from_buffer_copy(xpads[i][0]); xl = xpads[i][1]
    else:
        xp, xl = None, 0
    n = L.toolame_encode_frame(buf.ctypes.data, xp, xl, out, 4096)
    chunks.append(bytes(out[:n])); lens.append(n)
    if i in taps_for:
        hdr = tap('header', C.c_int, (14,))
        frm_jsbound = None
        taps[i] = dict(
            sb_sample=tap('sb_sample', C.c_double, (2,3,12,32), True),
            j_sample=tap('j_sample', C.c_double, (3,12,32), True),
            subband=tap('subband', C.c_uint, (2,3,12,32), True),
            scalar=tap('scalar', C.c_uint, (2,3,32)), j_scale=tap('j_scale', C.c_uint, (3,32)),
            smr=tap('smr', C.c_double, (2,32)), max_sc=tap('max_sc', C.c_double, (2,32)),
            scfsi=tap('scfsi', C.c_uint, (2,32)), bit_alloc=tap('bit_alloc', C.c_uint, (2,32)),
            mode=int(hdr[9]), mode_ext=int(hdr[10]))
n = L.toolame_finish(out, 4096)
chunks.append(bytes(out[:n])); lens.append(n)
pickle.dump(dict(rc=rc, data=b''.join(chunks), lens=lens, taps=taps), open(cfg_pkl + '.out', 'wb'))
"""


def reference_stream(pcm_frames, samplerate=48000, mode="s", kbps=128, psy=1, pad_len=0, xpads=None,
                     tap_frames=()):
    """Run the REAL reference (oracle/_ref) in a child process. Returns dict(data, lens, taps, rc)."""
    import pickle
    import tempfile
    if not REF_SO.exists():
        raise FileNotFoundError(REF_SO)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "cfg.pkl")
        pickle.dump(dict(samplerate=samplerate, mode=mode, kbps=kbps, psy=psy, pad_len=pad_len,
                         pcm=np.ascontiguousarray(pcm_frames, dtype=np.int16), xpads=xpads,
                         tap_frames=list(tap_frames)), open(p, "wb"))
        r = subprocess.run([sys.executable, "-c", _REF_CHILD, str(REF_SO), p], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference child failed: " + r.stderr[-2000:])
        return pickle.load(open(p + ".out", "rb"))
