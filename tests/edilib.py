"""Test helpers for the EDI AF-packet step: state record, emulated device path, reference driver."""
import ctypes as C
import hashlib
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
# mirror of TlEdiState (odr-audioenc_amd/csrc/edi_af.h) == tlb_edi_state (include/toolame_batch.h)
EDI_STATE = np.dtype([("edi_time", np.int64), ("send_version_at_time", np.int64), ("timestamp", np.uint32),
                      ("num_seconds_sent", np.uint32), ("tai_utc_offset", np.int32), ("seq", np.uint16), ("dlfc", np.uint16),
                      ("tist", np.uint8), ("pad_", np.uint8, (7,))])
VERSION = b"v3.5.0-graft"


def payload(nframes, nstreams, stride, frame_bytes, seed):
    """deterministic pseudo-frames (integer LCG), zero beyond each stream's frame size"""
    x = np.uint32(seed * 2654435761 % (1 << 32) | 1)
    out = np.zeros((nframes, nstreams, stride), dtype=np.uint8)
    v = np.arange(nframes * nstreams * stride, dtype=np.uint64)
    v = (v * np.uint64(1664525) + np.uint64(int(x))) % np.uint64(1 << 32)
    v = ((v >> np.uint64(13)) ^ (v >> np.uint64(21))) & np.uint64(0xff)
    out[...] = v.astype(np.uint8).reshape(out.shape)
    for s in range(nstreams):
        out[:, s, frame_bytes[s]:] = 0
    return out


def init_state(n, now_s, delay_ms, tist, tai):
    """first-call branch of EDI::write_frame, src/Outputs.cpp:200-212"""
    st = np.zeros(n, dtype=EDI_STATE)
    st["edi_time"] = now_s + delay_ms // 1000
    st["send_version_at_time"] = st["edi_time"]
    ts = 0
    sub = delay_ms % 1000
    while sub > 0:
        ts += 24 << 14
        sub -= 24
    st["timestamp"] = ts
    st["tist"] = 1 if tist else 0
    st["tai_utc_offset"] = tai
    return st


def pkt_stride(out_stride, vlen):
    return (10 + 16 + 18 + 11 + out_stride + 12 + 12 + vlen + 2 + 3) & ~3


def emu_af(frames, levels, frame_bytes, state, version=VERSION):
    import emulib as E
    L = E.lib()
    assert L.emu_sizeof_edi_state() == EDI_STATE.itemsize
    nf, ns, stride = frames.shape
    ps = pkt_stride(stride, len(version))
    pkts = np.zeros((nf, ns, ps), dtype=np.uint8)
    plen = np.zeros((nf, ns), dtype=np.int32)
    fb = np.ascontiguousarray(frame_bytes, dtype=np.int32)
    st = state.copy()
    lv = np.ascontiguousarray(levels, dtype=np.int16) if levels is not None else None
    L.emu_edi_af.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int,
                             C.c_void_p, C.c_int, C.c_void_p]
    assert L.emu_edi_af(frames.ctypes.data, lv.ctypes.data if lv is not None else None, nf, ns, stride, fb.ctypes.data,
                        st.ctypes.data, version, len(version), pkts.ctypes.data, ps, plen.ctypes.data) == 0
    return pkts, plen, st


def ref_lib():
    p = ROOT / "oracle" / "_ref" / "libedi_ref.so"
    if not p.exists() and Path("/root/reference").exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "_ref/libedi_ref.so"], check=True)
    if not p.exists():
        return None
    L = C.CDLL(str(p))
    L.ediref_stream.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int,
                                C.c_void_p]
    return L


def ref_af(frames, levels, frame_bytes, state, version=VERSION):
    """the reference's own TagItems/TagPacket/AFPacket classes, one stream at a time"""
    L = ref_lib()
    nf, ns, stride = frames.shape
    ps = pkt_stride(stride, len(version))
    pkts = np.zeros((nf, ns, ps), dtype=np.uint8)
    plen = np.zeros((nf, ns), dtype=np.int32)
    st = state.copy()
    for s in range(ns):
        f = np.ascontiguousarray(frames[:, s, :])
        lv = np.ascontiguousarray(levels[:, s, :], dtype=np.int16) if levels is not None else None
        one = st[s:s + 1].copy()
        pk = np.zeros((nf, ps), dtype=np.uint8)
        pl = np.zeros(nf, dtype=np.int32)
        assert L.ediref_stream(f.ctypes.data, nf, int(frame_bytes[s]), stride, lv.ctypes.data if lv is not None else None,
                               one.ctypes.data, version, len(version), pk.ctypes.data, ps, pl.ctypes.data) == 0
        pkts[:, s, :], plen[:, s], st[s] = pk, pl, one[0]
    return pkts, plen, st


def digest(pkts, plen):
    """sha256 over every packet's used bytes, frame-major"""
    h = hashlib.sha256()
    for f in range(pkts.shape[0]):
        for s in range(pkts.shape[1]):
            h.update(pkts[f, s, : plen[f, s]].tobytes())
    return h.hexdigest()


# (name, nframes, frame sizes per stream, init (now, delay_ms, tist, tai), start seq/dlfc, levels?)
CASES = [
    ("tist_wrap", 120, [384, 288], (1700000000, 250, 1, 37), (65530, 4990), True),
    ("plain", 60, [384], (1600000123, 0, 0, 37), (0, 0), False),
    ("long_version_cadence", 900, [576, 96, 384], (1751234567, 1015, 1, 37), (12, 2500), True),
]


def case_inputs(name):
    for c in CASES:
        if c[0] == name:
            _, nf, fbs, (now, delay, tist, tai), (seq, dlfc), with_levels = c
            stride = max(fbs)
            frames = payload(nf, len(fbs), stride, fbs, seed=len(name) + nf)
            st = init_state(len(fbs), now, delay, tist, tai)
            st["seq"], st["dlfc"] = seq, dlfc
            levels = None
            if with_levels:
                k = np.arange(nf * len(fbs) * 2, dtype=np.int64)
                levels = ((k * 7919 + 13) % 32768).astype(np.int16).reshape(nf, len(fbs), 2)
            return frames, levels, np.array(fbs, dtype=np.int32), st
    raise KeyError(name)
