"""ctypes binding for the TEST-ONLY host emulation of the HIP kernels (tests/emu/libmp2emu.so)."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
EMU_DIR = ROOT / "tests" / "emu"

# mirror of TlTaps (odr-audioenc_amd/csrc/mp2_types.h)
TAPS_DTYPE = np.dtype([
    ("sb_sample", np.float64, (2, 3, 12, 32)), ("smr", np.float64, (2, 32)), ("max_sc", np.float64, (2, 32)),
    ("subband", np.uint32, (2, 3, 12, 32)), ("scalar_pre", np.uint8, (2, 3, 32)), ("scalar", np.uint8, (2, 3, 32)),
    ("j_scale", np.uint8, (3, 32)), ("scfsi", np.uint8, (2, 32)), ("bit_alloc", np.uint8, (2, 32)),
    ("adb_left", np.int32), ("mode", np.int32), ("mode_ext", np.int32), ("jsbound", np.int32), ("crc16", np.int32),
    ("scfcrc", np.uint8, (4,)), ("pad_", np.int32, (2,)),
])

TL_MAX_XPAD = 256
_lib = None


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get("TL_EMU_LIB")          # e.g. a sanitizer build (tools/emu_sanitize.sh)
        if not alt:
            subprocess.run(["make", "-s", "-C", str(EMU_DIR)], check=True)
        L = C.CDLL(alt or str(EMU_DIR / "libmp2emu.so"))
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_destroy.argtypes = [C.c_void_p]
        L.emu_pair_units.argtypes = [C.c_void_p]; L.emu_pair_units.restype = C.c_long
        L.emu_frame_bytes.argtypes = [C.c_void_p, C.c_int]
        L.emu_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.emu_encode_len.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_max_frame_bytes.argtypes = [C.c_void_p, C.c_int]
        L.emu_pending.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_log10.restype = C.c_double
        L.emu_log10.argtypes = [C.c_double]
        L.emu_log10_pn.restype = C.c_double
        L.emu_log10_pn.argtypes = [C.c_double]
        L.emu_pow10.restype = C.c_double
        L.emu_pow10.argtypes = [C.c_double]
        assert L.emu_sizeof_taps() == TAPS_DTYPE.itemsize, (L.emu_sizeof_taps(), TAPS_DTYPE.itemsize)
        _lib = L
    return _lib


def pack_xpad(xpad_full, xpad_len, pad_len):
    """reference layout (padlen+1 bytes, X-PAD at [padlen-xpad_len, padlen-2), F-PAD last two;
    toolame.c:515-551) -> the xpad_len bytes in transmission order, zero padded to TL_MAX_XPAD."""
    out = np.zeros(TL_MAX_XPAD, dtype=np.uint8)
    if xpad_len:
        out[:xpad_len] = np.frombuffer(bytes(xpad_full), dtype=np.uint8)[pad_len - xpad_len: pad_len]
    return out


class EmuBatch:
    """N streams on the emulated device path.  encode() returns whole frames with one frame of
    latency (the ScF-CRC of frame n travels in frame n-1, toolame.c:527-542); flush() returns the last."""

    def __init__(self, cfgs):
        self.L = lib()
        n = len(cfgs)
        fs = (C.c_long * n)(*[c.get("samplerate", 48000) for c in cfgs])
        mode = bytes(ord(c.get("mode", "s")) for c in cfgs)
        kb = (C.c_int * n)(*[c.get("kbps", 128) for c in cfgs])
        psy = (C.c_int * n)(*[c.get("psy", 1) for c in cfgs])
        pad = (C.c_int * n)(*[c.get("pad_len", 0) for c in cfgs])
        err = C.c_int(0)
        self.h = self.L.emu_create(n, fs, mode, kb, psy, pad, C.byref(err))
        if not self.h:
            raise ValueError(f"illegal configuration (code {err.value})")
        self.n = n
        self.frame_bytes = [self.L.emu_frame_bytes(self.h, s) for s in range(n)]
        self.stride = (max(self.L.emu_max_frame_bytes(self.h, s) for s in range(n)) + 3) & ~3
        self.frames_in = 0

    def encode(self, pcm, xpad=None, xpad_len=None, want_taps=False):
        """pcm [nframes, nstreams, 2, 1152] int16 -> (per stream: bytes of the frames that became final,
        taps array [nframes, nstreams] or None)"""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        nf = pcm.shape[0]
        assert pcm.shape == (nf, self.n, 2, 1152)
        out = np.zeros((nf, self.n, self.stride), dtype=np.uint8)
        taps = np.zeros((nf, self.n), dtype=TAPS_DTYPE) if want_taps else None
        xp = xl = None
        if xpad is not None:
            xp = np.ascontiguousarray(xpad, dtype=np.uint8)
            xl = np.ascontiguousarray(xpad_len, dtype=np.int32)
            assert xp.shape == (nf, self.n, TL_MAX_XPAD) and xl.shape == (nf, self.n)
        lens = np.zeros((nf, self.n), dtype=np.int32)
        self.L.emu_encode_len(self.h, pcm.ctypes.data, nf, xp.ctypes.data if xp is not None else None,
                              xl.ctypes.data if xl is not None else None, out.ctypes.data, self.stride,
                              taps.ctypes.data if taps is not None else None, lens.ctypes.data)
        res = []
        for s in range(self.n):                          # a slot's length: 0 for slot 0 of the very first call (no frame yet); at 44.1 /
            res.append(b"".join(out[f, s, : lens[f, s]].tobytes() for f in range(nf)))      # 22.05 kHz frame_bytes or one more
        self.frames_in += nf
        return res, taps

    def flush(self):
        res = []
        for s in range(self.n):
            buf = (C.c_uint8 * 2048)()
            n = self.L.emu_pending(self.h, s, buf)
            res.append(bytes(buf[:n]))
        return res

    def pair_units(self):
        """(pair of mono streams, frame) units encoded by one wave so far (tl_encode_pair)"""
        return int(self.L.emu_pair_units(self.h))

    def close(self):
        if self.h:
            self.L.emu_destroy(self.h)
            self.h = None
