"""ctypes binding of libtoolame_dab_hip.so (include/toolame_batch.h).

Mirrors the reference's operator interface for this path: the six setters of
libtoolame-dab/toolame.h:13-48 become the fields of StreamConfig (same names, same argument meaning,
same validity rules), `Batch.encode()` is toolame_encode_frame() over N streams and F frames, and
`Batch.flush()` is toolame_finish().  Errors surface as ToolameError with the library's code, where
the reference returns non-zero or exit()s.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("TLB_LIB_PATH") or PKG_DIR / "libtoolame_dab_hip.so")     # TLB_LIB_PATH: kernel experiments (tools/) only
FAULT_LIB_PATH = PKG_DIR / "libtoolame_dab_hip_fi.so"    # TEST build with fault injection (csrc/tlb_debug.h, `make fault`); never the product
MAX_XPAD = 256
SAMPLES = 1152

_ERR = {1: "illegal sample rate (48000/44100/32000/24000/22050/16000 Hz; the egress calls: no 32/44.1/22.05 kHz)", 2: "bad channel mode", 3: "invalid PSY model",
        4: "illegal bitrate for this MPEG version", 5: "invalid XPAD length", 16: "no usable HIP device",
        17: "HIP runtime error", 18: "bad argument"}


class ToolameError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__(f"libtoolame-dab-hip: {what or 'error'}: {_ERR.get(code, 'unknown')} (code {code})")


@dataclass(frozen=True)
class StreamConfig:
    """toolame_set_samplerate / _channel_mode / _bitrate / _psy_model / _pad (toolame.c:174-262)."""
    samplerate: int = 48000
    mode: str = "j"          # odr-audioenc's default for two channels (src/odr-audioenc.cpp:697-709)
    bitrate: int = 128
    psy_model: int = 1
    pad_len: int = 0


class _CConfig(C.Structure):
    _fields_ = [("samplerate", C.c_long), ("mode", C.c_char), ("bitrate", C.c_int), ("psy_model", C.c_int),
                ("pad_len", C.c_int)]


_lib = None


def build(verbose=False):
    """Compile csrc/ for gfx950 into libtoolame_dab_hip.so (hipcc cross-compiles without a GPU), and the fault-injection TEST build
    of the same kernels beside it (libtoolame_dab_hip_fi.so: host files only, seconds)."""
    for target in ([], ["fault"]):
        r = subprocess.run(["make", "-j4", "-C", str(PKG_DIR / "csrc")] + target, capture_output=not verbose, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc build failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


_fault_lib = None


def load_fault_library():
    """The TEST build with fault injection (csrc/tlb_debug.h) as a second, independent library object: pass it as `lib=` to Batch / Tick /
    Node.  Only tests use it -- the product library has no such entry points."""
    global _fault_lib
    if _fault_lib is None:
        if not FAULT_LIB_PATH.exists():
            raise ToolameError(16, f"{FAULT_LIB_PATH} is missing (make -C odr-audioenc_amd/csrc fault)")
        L = _bind(C.CDLL(str(FAULT_LIB_PATH)))
        L.tlb_debug_fail_next.argtypes = [C.c_void_p, C.c_int]
        L.tlb_debug_tick_fail_next.argtypes = [C.c_void_p, C.c_int]
        L.tlb_debug_node_fail_next.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _fault_lib = L
    return _fault_lib


def load_library():
    """Load the HIP library or raise -- there is no CPU fallback for the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ToolameError(16, f"{LIB_PATH} is missing (run __graft_entry__.build())")
    _lib = _bind(C.CDLL(str(LIB_PATH)))
    return _lib


def _bind(L):
    """argument / result types of every entry point of include/toolame_batch.h"""
    L.tlb_create.restype = C.c_void_p
    L.tlb_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    L.tlb_destroy.argtypes = [C.c_void_p]
    L.tlb_reset.argtypes = [C.c_void_p]
    for f in ("tlb_nstreams", "tlb_out_stride"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.tlb_frame_bytes.argtypes = [C.c_void_p, C.c_int]
    L.tlb_frames_encoded.argtypes = [C.c_void_p]
    L.tlb_frames_encoded.restype = C.c_long
    L.tlb_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_encode_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_host_alloc.restype = C.c_void_p
    L.tlb_host_alloc.argtypes = [C.c_size_t]
    L.tlb_host_free.argtypes = [C.c_void_p]
    L.tlb_host_free.restype = None
    L.tlb_flush_host.argtypes = [C.c_void_p, C.c_void_p]
    L.tlb_flush_host_len.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_encode_host_len.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_encode_device_len.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_flush_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_last_kernel_ms.argtypes = [C.c_void_p]
    L.tlb_last_kernel_ms.restype = C.c_float
    L.tlb_last_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.tlb_version.restype = C.c_char_p
    L.tlb_set_gain_db.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.tlb_ingest_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_ingest_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    for f in ("tlb_egress_unit_bytes", "tlb_egress_units_per_frame"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_int]
    L.tlb_egress_max_units_per_frame.argtypes = [C.c_void_p]
    L.tlb_zmq_msg_stride.argtypes = [C.c_void_p]
    L.tlb_zmq_frame_host.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
    L.tlb_edi_state_init.argtypes = [C.c_void_p, C.c_longlong, C.c_uint, C.c_int, C.c_int]
    L.tlb_edi_state_init.restype = None
    L.tlb_edi_af_stride.argtypes = [C.c_void_p, C.c_int]
    L.tlb_edi_af_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_edi_af_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    L.tlb_edi_pft_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tlb_edi_pft_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int] * 2
    L.tlb_edi_pft_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3 + [C.c_int] * 2 + [C.c_void_p]
    L.tlb_stream_reset.argtypes = [C.c_void_p, C.c_int]
    L.tlb_stream_finish.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.tlb_stream_reconfigure.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.tlb_tick_stream_reset.argtypes = [C.c_void_p, C.c_int]
    L.tlb_tick_stream_finish.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.tlb_tick_stream_reconfigure.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.tlb_tick_create.restype = C.c_void_p
    L.tlb_tick_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.tlb_tick_destroy.argtypes = [C.c_void_p]
    L.tlb_tick_destroy.restype = None
    for f in ("tlb_tick_pcm", "tlb_tick_xpad", "tlb_tick_xpad_len", "tlb_tick_peaks"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_void_p]
    for f in ("tlb_tick_run", "tlb_tick_finish", "tlb_tick_submit", "tlb_tick_wait"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.tlb_tick_count.argtypes = [C.c_void_p]
    L.tlb_tick_count.restype = C.c_long
    L.tlb_tick_set_gain_db.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.tlb_tick_units.argtypes = [C.c_void_p, C.c_int]
    L.tlb_tick_frame.restype = C.c_void_p
    L.tlb_tick_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.tlb_tick_packet.restype = C.c_void_p
    L.tlb_tick_packet.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.tlb_tick_message.restype = C.c_void_p
    L.tlb_tick_message.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.tlb_tick_silence_ms.restype = C.c_void_p
    L.tlb_tick_silence_ms.argtypes = [C.c_void_p]
    L.tlb_tick_fragments.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.tlb_tick_fragment.restype = C.c_void_p
    L.tlb_tick_fragment.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.tlb_tick_last_ms.argtypes = [C.c_void_p]
    L.tlb_tick_last_ms.restype = C.c_float
    # node level (include/toolame_batch.h part 3)
    L.tlb_node_partition.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tlb_node_partition.restype = None
    L.tlb_node_plan_shard.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 3 + [C.c_void_p, C.POINTER(C.c_int)]
    L.tlb_node_create.restype = C.c_void_p
    L.tlb_node_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.tlb_node_destroy.argtypes = [C.c_void_p]
    L.tlb_node_destroy.restype = None
    for f in ("tlb_node_nshards", "tlb_node_nstreams", "tlb_node_submit", "tlb_node_wait", "tlb_node_run", "tlb_node_finish", "tlb_node_sync"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.tlb_node_shard_of.argtypes = [C.c_void_p, C.c_int]
    L.tlb_node_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_node_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for f in ("tlb_node_pcm", "tlb_node_xpad", "tlb_node_xpad_len", "tlb_node_peaks"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_void_p, C.c_int]
    L.tlb_node_units.argtypes = [C.c_void_p, C.c_int]
    L.tlb_node_silence_ms.argtypes = [C.c_void_p, C.c_int]
    L.tlb_node_silence_ms.restype = C.c_uint32
    L.tlb_node_frame.restype = C.c_void_p
    L.tlb_node_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    for f in ("tlb_node_packet", "tlb_node_message"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.tlb_node_fragments.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.tlb_node_fragment.restype = C.c_void_p
    L.tlb_node_fragment.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.tlb_node_set_gain_db.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.tlb_node_stream_reset.argtypes = [C.c_void_p, C.c_int]
    L.tlb_node_stream_finish.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.tlb_node_stream_reconfigure.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.tlb_node_batch.restype = C.c_void_p
    L.tlb_node_batch.argtypes = [C.c_void_p, C.c_int]
    L.tlb_node_device_alloc.restype = C.c_void_p
    L.tlb_node_device_alloc.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.tlb_node_device_free.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.tlb_node_device_free.restype = None
    L.tlb_node_copy_in.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.tlb_node_copy_out.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    L.tlb_node_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tlb_node_flush_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.toolame_set_samplerate.argtypes = [C.c_long]
    L.toolame_set_channel_mode.argtypes = [C.c_char]
    L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.toolame_finish.argtypes = [C.c_void_p, C.c_size_t]
    if hasattr(L, "tlb_node_shard_status"):       # (an older build loaded through TLB_LIB_PATH for a kernel A/B has none of the round-6 entry points)
        L.tlb_tick_status.argtypes = [C.c_void_p]
        L.tlb_node_describe.argtypes = [C.c_void_p]
        L.tlb_node_describe.restype = C.c_char_p
        L.tlb_node_shard_status.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.tlb_node_shard_restart.argtypes = [C.c_void_p, C.c_int, C.c_longlong]
    return L


def lds_bytes_per_stream():
    return load_library().tlb_lds_bytes_per_stream()


def legacy_api():
    """The nine reference symbols (toolame_init ... toolame_encode_frame) as a ctypes library."""
    return load_library()


# mirror of tlb_edi_state (include/toolame_batch.h): the per-stream EDI sender state
EDI_STATE_DTYPE = np.dtype([("edi_time", np.int64), ("send_version_at_time", np.int64), ("timestamp", np.uint32),
                            ("num_seconds_sent", np.uint32), ("tai_utc_offset", np.int32), ("seq", np.uint16), ("dlfc", np.uint16),
                            ("tist", np.uint8), ("pad_", np.uint8, (7,))])


def edi_state_init(nstreams, now_s, delay_ms=0, tist=False, tai_utc_offset=37):
    """per-stream EDI sender state as EDI::write_frame sets it up on its first call (src/Outputs.cpp:200-212)"""
    st = np.zeros(nstreams, dtype=EDI_STATE_DTYPE)
    L = load_library()
    for s in range(nstreams):
        L.tlb_edi_state_init(st[s:s + 1].ctypes.data, int(now_s), int(delay_ms), 1 if tist else 0, int(tai_utc_offset))
    return st


# mirror of TlTaps (csrc/mp2_types.h) for stage-level parity tests
TAPS_DTYPE = np.dtype([
    ("sb_sample", np.float64, (2, 3, 12, 32)), ("smr", np.float64, (2, 32)), ("max_sc", np.float64, (2, 32)),
    ("subband", np.uint32, (2, 3, 12, 32)), ("scalar_pre", np.uint8, (2, 3, 32)), ("scalar", np.uint8, (2, 3, 32)),
    ("j_scale", np.uint8, (3, 32)), ("scfsi", np.uint8, (2, 32)), ("bit_alloc", np.uint8, (2, 32)),
    ("adb_left", np.int32), ("mode", np.int32), ("mode_ext", np.int32), ("jsbound", np.int32), ("crc16", np.int32),
    ("scfcrc", np.uint8, (4,)), ("pad_", np.int32, (2,)),
])


def _config_array(configs):
    arr = (_CConfig * len(configs))()
    for i, c in enumerate(configs):
        arr[i].samplerate = c.samplerate
        arr[i].mode = c.mode.encode()[:1]
        arr[i].bitrate = c.bitrate
        arr[i].psy_model = c.psy_model
        arr[i].pad_len = c.pad_len
    return arr


class _CTickConfig(C.Structure):
    _fields_ = [("egress", C.c_int), ("ngroups", C.c_int), ("with_xpad", C.c_int), ("version", C.c_char_p), ("version_len", C.c_int),
                ("now_s", C.c_longlong), ("delay_ms", C.c_uint), ("tist", C.c_int), ("tai_utc_offset", C.c_int),
                ("fec", C.c_int), ("chunk_len", C.c_int), ("transport", C.c_int), ("addr_source", C.c_int), ("dest_port", C.c_int)]


class Tick:
    """tlb_tick_*: the caller's loop body -- ingest, encode, EDI egress -- as one call per tick for every stream of a GPU
    (src/odr-audioenc.cpp:1030-1051,1139-1163,1208-1225, src/Outputs.cpp:194-261).  `pcm` (and `xpad`, `xpad_len`) are numpy
    views of the object's pinned host buffers: fill them, run(), read packets()."""
    EGRESS = {"frames": 0, "af": 1, "pft": 2, "zmq": 3}

    def __init__(self, configs, egress="af", ngroups=0, with_xpad=False, version=b"", now_s=1700000000, delay_ms=0, tist=False,
                 tai_utc_offset=37, fec=0, chunk_len=207, transport=False, addr_source=0, dest_port=0, device=0, lib=None):
        self.L = lib or load_library()
        configs = list(configs)
        self.nstreams = len(configs)
        self._version = bytes(version)
        tc = _CTickConfig(self.EGRESS[egress], ngroups, 1 if with_xpad else 0, self._version, len(self._version), int(now_s), int(delay_ms),
                          1 if tist else 0, int(tai_utc_offset), fec, chunk_len, 1 if transport else 0, addr_source, dest_port)
        err = C.c_int(0)
        self.h = self.L.tlb_tick_create(device, self.nstreams, _config_array(configs), C.byref(tc), C.byref(err))
        if not self.h:
            raise ToolameError(err.value, "tlb_tick_create")
        self.egress = egress
        n = self.nstreams
        self.with_xpad = bool(with_xpad)
        self.units = [self.L.tlb_tick_units(self.h, s) for s in range(n)]

    # The object owns TWO sets of pinned host buffers (tlb_tick_submit / tlb_tick_wait): `pcm`, `xpad`, `xpad_len` are views of the
    # input set to fill NEXT, `peaks`, `silence_ms` of the results of the tick waited for last -- fetched on every access.
    def _view(self, fn, ctype, shape):
        p = getattr(self.L, fn)(self.h)
        cnt = int(np.prod(shape))
        return np.ctypeslib.as_array((ctype * cnt).from_address(p)).reshape(shape) if p else None

    @property
    def pcm(self):
        return self._view("tlb_tick_pcm", C.c_int16, (self.nstreams, 2 * SAMPLES))

    @property
    def xpad(self):
        return self._view("tlb_tick_xpad", C.c_uint8, (self.nstreams, MAX_XPAD)) if self.with_xpad else None

    @property
    def xpad_len(self):
        return self._view("tlb_tick_xpad_len", C.c_int32, (self.nstreams,)) if self.with_xpad else None

    @property
    def peaks(self):
        return self._view("tlb_tick_peaks", C.c_int16, (self.nstreams, 2))

    @property
    def silence_ms(self):
        return self._view("tlb_tick_silence_ms", C.c_uint32, (self.nstreams,))

    def submit(self):
        """queue one tick on the input set just filled and return at once; `pcm` then shows the other input set"""
        rc = self.L.tlb_tick_submit(self.h)
        if rc:
            raise ToolameError(rc, "tlb_tick_submit")

    def wait(self):
        """wait for the oldest submitted tick; frame() / packets() / peaks ... then show its results"""
        rc = self.L.tlb_tick_wait(self.h)
        if rc:
            raise ToolameError(rc, "tlb_tick_wait")

    def set_gain_db(self, gain_db, stream=-1):
        rc = self.L.tlb_tick_set_gain_db(self.h, stream, float(gain_db))
        if rc:
            raise ToolameError(rc, "tlb_tick_set_gain_db")

    # -- life cycle of one stream between two runs (toolame_init / toolame_finish / the setters, per stream) --
    def stream_reset(self, s):
        rc = self.L.tlb_tick_stream_reset(self.h, s)
        if rc:
            raise ToolameError(rc, "tlb_tick_stream_reset")

    def stream_finish(self, s):
        buf = (C.c_uint8 * 2048)()
        n = self.L.tlb_tick_stream_finish(self.h, s, buf, 2048)
        if n < 0:
            raise ToolameError(-n, "tlb_tick_stream_finish")
        return bytes(buf[:n])

    def stream_reconfigure(self, s, config):
        rc = self.L.tlb_tick_stream_reconfigure(self.h, s, _config_array([config]))
        if rc:
            raise ToolameError(rc, "tlb_tick_stream_reconfigure")
        self.units[s] = self.L.tlb_tick_units(self.h, s)

    def status(self):
        """0, or 17 (TLB_ERR_HIP) once a device failure has left the object out of step with itself (sticky: destroy it)"""
        return self.L.tlb_tick_status(self.h)

    def fail_next(self, nth=1):
        """fault-injection TEST build only (lib=load_fault_library()): the nth submit from now fails in its last stream group"""
        return self.L.tlb_debug_tick_fail_next(self.h, nth)

    def run(self):
        rc = self.L.tlb_tick_run(self.h)
        if rc:
            raise ToolameError(rc, "tlb_tick_run")

    def finish(self):
        rc = self.L.tlb_tick_finish(self.h)
        if rc:
            raise ToolameError(rc, "tlb_tick_finish")

    def last_ms(self):
        return float(self.L.tlb_tick_last_ms(self.h))

    def frame(self, s):
        n = C.c_int(0)
        p = self.L.tlb_tick_frame(self.h, s, C.byref(n))
        return C.string_at(p, n.value) if p and n.value else b""

    def packets(self, s):
        """the AF packets of stream s from the last run, one per unit (empty list: none this tick)"""
        out = []
        for u in range(self.units[s]):
            n = C.c_int(0)
            p = self.L.tlb_tick_packet(self.h, s, u, C.byref(n))
            if p and n.value:
                out.append(C.string_at(p, n.value))
        return out

    def messages(self, s):
        """the ZeroMQ messages (header + unit) of stream s from the last run"""
        out = []
        for u in range(self.units[s]):
            n = C.c_int(0)
            p = self.L.tlb_tick_message(self.h, s, u, C.byref(n))
            if p and n.value:
                out.append(C.string_at(p, n.value))
        return out

    def fragments(self, s):
        """per unit: the list of PFT fragments of stream s from the last run"""
        out = []
        for u in range(self.units[s]):
            fr = []
            for k in range(self.L.tlb_tick_fragments(self.h, s, u)):
                n = C.c_int(0)
                p = self.L.tlb_tick_fragment(self.h, s, u, k, C.byref(n))
                fr.append(C.string_at(p, n.value))
            if fr:
                out.append(fr)
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.tlb_tick_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """N independent DAB MP2 encoders on one MI355X (one wavefront per stream)."""

    def __init__(self, configs, device=0, lib=None):
        self.L = lib or load_library()
        configs = list(configs)
        if not configs:
            raise ToolameError(18, "empty batch")
        arr = _config_array(configs)
        err = C.c_int(0)
        self.h = self.L.tlb_create(device, len(configs), arr, C.byref(err))
        if not self.h:
            raise ToolameError(err.value, "tlb_create")
        self.device = device
        self.nstreams = len(configs)
        self.configs = configs
        self.frame_bytes = [self.L.tlb_frame_bytes(self.h, s) for s in range(self.nstreams)]
        self.out_stride = self.L.tlb_out_stride(self.h)
        self.unit_bytes = [self.L.tlb_egress_unit_bytes(self.h, s) for s in range(self.nstreams)]
        self.units_per_frame = [self.L.tlb_egress_units_per_frame(self.h, s) for s in range(self.nstreams)]
        self.max_upf = self.L.tlb_egress_max_units_per_frame(self.h)
        self._first = True

    def fail_next(self, nth=1):
        """fault-injection TEST build only (lib=load_fault_library()): the nth launch from now fails half way (csrc/tlb_debug.h)"""
        return self.L.tlb_debug_fail_next(self.h, nth)

    # -- host-buffer path (tests, legacy-style callers) ------------------------------------
    def encode(self, pcm, xpad=None, xpad_len=None, want_taps=False):
        """pcm int16 [nframes, nstreams, 2, 1152] -> (per stream: bytes of the frames that became
        final during this call, taps [nframes, nstreams] or None).  One frame of latency, see
        include/toolame_batch.h."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        nf = pcm.shape[0]
        if pcm.shape != (nf, self.nstreams, 2, SAMPLES):
            raise ToolameError(18, f"pcm shape {pcm.shape}")
        out = np.zeros((nf, self.nstreams, self.out_stride), dtype=np.uint8)
        taps = np.zeros((nf, self.nstreams), dtype=TAPS_DTYPE) if want_taps else None
        xp = xl = None
        if xpad is not None:
            xp = np.ascontiguousarray(xpad, dtype=np.uint8)
            xl = np.ascontiguousarray(xpad_len, dtype=np.int32)
            if xp.shape != (nf, self.nstreams, MAX_XPAD) or xl.shape != (nf, self.nstreams):
                raise ToolameError(18, "xpad shape")
        lens = np.zeros((nf, self.nstreams), dtype=np.int32)
        rc = self.L.tlb_encode_host_len(self.h, pcm.ctypes.data, nf, xp.ctypes.data if xp is not None else None,
                                        xl.ctypes.data if xl is not None else None, out.ctypes.data, lens.ctypes.data,
                                        taps.ctypes.data if taps is not None else None)
        if rc:
            raise ToolameError(rc, "tlb_encode_host_len")
        self._first = False
        # a slot's length: 0 for slot 0 of the very first call (no frame yet); frame_bytes, or one more at 44.1 / 22.05 kHz
        res = [b"".join(out[f, s, : lens[f, s]].tobytes() for f in range(nf)) for s in range(self.nstreams)]
        return res, taps

    def flush(self):
        """toolame_finish(): the last (pending) frame of every stream, carrying its own ScF-CRC."""
        out = np.zeros((self.nstreams, self.out_stride), dtype=np.uint8)
        lens = np.zeros(self.nstreams, dtype=np.int32)
        rc = self.L.tlb_flush_host_len(self.h, out.ctypes.data, lens.ctypes.data)
        if rc:
            raise ToolameError(rc, "tlb_flush_host_len")
        return [out[s, : lens[s]].tobytes() for s in range(self.nstreams)]

    def reset(self):
        rc = self.L.tlb_reset(self.h)
        if rc:
            raise ToolameError(rc, "tlb_reset")

    # -- life cycle of one stream inside the live batch (toolame.c:120-166 per stream) --------
    def stream_reset(self, s):
        """toolame_init() for stream s alone: its next frame is its frame 0; no other stream notices"""
        rc = self.L.tlb_stream_reset(self.h, s)
        if rc:
            raise ToolameError(rc, "tlb_stream_reset")

    def stream_finish(self, s):
        """toolame_finish() for stream s alone: its pending frame (bytes), then as after stream_reset()"""
        buf = (C.c_uint8 * 2048)()
        n = self.L.tlb_stream_finish(self.h, s, buf, 2048)
        if n < 0:
            raise ToolameError(-n, "tlb_stream_finish")
        return bytes(buf[:n])

    def stream_reconfigure(self, s, config):
        """the six setters + toolame_init() for stream s alone"""
        rc = self.L.tlb_stream_reconfigure(self.h, s, _config_array([config]))
        if rc:
            raise ToolameError(rc, "tlb_stream_reconfigure")
        self.configs[s] = config
        self.frame_bytes[s] = self.L.tlb_frame_bytes(self.h, s)
        self.unit_bytes[s] = self.L.tlb_egress_unit_bytes(self.h, s)
        self.units_per_frame[s] = self.L.tlb_egress_units_per_frame(self.h, s)

    # -- device-resident path (bench, production) -------------------------------------------
    def encode_device(self, d_pcm_ptr, nframes, d_out_ptr, d_xpad_ptr=None, d_xpad_len_ptr=None, stream=None):
        rc = self.L.tlb_encode_device(self.h, d_pcm_ptr, nframes, d_xpad_ptr, d_xpad_len_ptr, d_out_ptr, stream)
        if rc:
            raise ToolameError(rc, "tlb_encode_device")
        self._first = False

    def flush_device(self, d_out_ptr, stream=None):
        rc = self.L.tlb_flush_device(self.h, d_out_ptr, stream)
        if rc:
            raise ToolameError(rc, "tlb_flush_device")

    # -- caller-side glue (gain, peak, de-interleave; src/odr-audioenc.cpp:1030-1051,1139-1152) ----
    def set_gain_db(self, gain_db, stream=-1):
        rc = self.L.tlb_set_gain_db(self.h, stream, float(gain_db))
        if rc:
            raise ToolameError(rc, "tlb_set_gain_db")

    def ingest(self, interleaved):
        """int16 [nframes, nstreams, 2304] interleaved s16le -> (planar [nframes, nstreams, 2, 1152], peaks [.., 2])"""
        a = np.ascontiguousarray(interleaved, dtype=np.int16)
        nf = a.shape[0]
        if a.shape != (nf, self.nstreams, 2 * SAMPLES):
            raise ToolameError(18, f"interleaved shape {a.shape}")
        pcm = np.zeros((nf, self.nstreams, 2, SAMPLES), dtype=np.int16)
        peaks = np.zeros((nf, self.nstreams, 2), dtype=np.int16)
        rc = self.L.tlb_ingest_host(self.h, a.ctypes.data, nf, pcm.ctypes.data, peaks.ctypes.data)
        if rc:
            raise ToolameError(rc, "tlb_ingest_host")
        return pcm, peaks

    def ingest_device(self, d_in_ptr, nframes, d_pcm_ptr, d_peaks_ptr, stream=None):
        rc = self.L.tlb_ingest_device(self.h, d_in_ptr, nframes, d_pcm_ptr, d_peaks_ptr, stream)
        if rc:
            raise ToolameError(rc, "tlb_ingest_device")

    def edi_af(self, frames, levels, state, version=b""):
        """frames uint8 [nframes, nstreams, out_stride] (encode() layout), levels int16 [nframes, nstreams, 2] or None,
        state = edi_state_init(...) (advanced in place) -> (packets uint8 [nframes * max_upf, nstreams, stride], lengths int32):
        one packet per 24-ms unit in slot order frame * max_upf + unit (include/toolame_batch.h, UNITS)"""
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        nf = f.shape[0]
        if f.shape != (nf, self.nstreams, self.out_stride) or state.dtype != EDI_STATE_DTYPE or state.shape != (self.nstreams,):
            raise ToolameError(18, f"frames {f.shape} / state {state.shape}")
        lv = np.ascontiguousarray(levels, dtype=np.int16) if levels is not None else None
        ps = self.L.tlb_edi_af_stride(self.h, len(version))
        if ps <= 0:
            raise ToolameError(18, "version string too long")
        if self.max_upf <= 0:
            raise ToolameError(1, "a stream's frames are no whole number of 24-ms units (32 kHz)")
        pkts = np.zeros((nf * self.max_upf, self.nstreams, ps), dtype=np.uint8)
        plen = np.zeros((nf * self.max_upf, self.nstreams), dtype=np.int32)
        rc = self.L.tlb_edi_af_host(self.h, f.ctypes.data, lv.ctypes.data if lv is not None else None, nf, state.ctypes.data,
                                    bytes(version), len(version), pkts.ctypes.data, plen.ctypes.data)
        if rc:
            raise ToolameError(rc, "tlb_edi_af_host")
        return pkts, plen

    def zmq_frames(self, frames, peaks=None):
        """frames uint8 [nframes, nstreams, out_stride] -> ZeroMQ messages uint8 [nframes * max_upf, nstreams, 12 + out_stride],
        one per 24-ms unit in slot order (struct zmq_frame_header_t + unit; datasize 0 = absent slot)"""
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        nf = f.shape[0]
        if f.shape != (nf, self.nstreams, self.out_stride) or self.max_upf <= 0:
            raise ToolameError(18 if self.max_upf > 0 else 1, f"frames {f.shape}")
        pk = np.ascontiguousarray(peaks, dtype=np.int16) if peaks is not None else None
        ms = self.L.tlb_zmq_msg_stride(self.h)
        msgs = np.zeros((nf * self.max_upf, self.nstreams, ms), dtype=np.uint8)
        rc = self.L.tlb_zmq_frame_host(self.h, f.ctypes.data, pk.ctypes.data if pk is not None else None, nf, msgs.ctypes.data)
        if rc:
            raise ToolameError(rc, "tlb_zmq_frame_host")
        return msgs

    def edi_pft(self, af, af_len, pseq, fec=0, chunk_len=207, transport=False, addr_source=0, dest_port=0):
        """AF packets (edi_af() output) -> PFT fragments.  pseq uint16 [nstreams] is advanced in place.
        Returns (fragments uint8 [nframes, nstreams, max_frags, frag_stride], lengths int32 [.., max_frags], counts int32 [nframes, nstreams])"""
        a = np.ascontiguousarray(af, dtype=np.uint8)
        nf, ns, stride = a.shape
        al = np.ascontiguousarray(af_len, dtype=np.int32)
        if ns != self.nstreams or al.shape != (nf, ns) or pseq.dtype != np.uint16 or pseq.shape != (ns,):
            raise ToolameError(18, "edi_pft argument shapes")
        mf, fs = C.c_int(0), C.c_int(0)
        rc = self.L.tlb_edi_pft_shape(self.h, stride, fec, chunk_len, 1 if transport else 0, C.byref(mf), C.byref(fs))
        if rc:
            raise ToolameError(rc, "tlb_edi_pft_shape")
        frags = np.zeros((nf, ns, mf.value, fs.value), dtype=np.uint8)
        flen = np.zeros((nf, ns, mf.value), dtype=np.int32)
        nfrag = np.zeros((nf, ns), dtype=np.int32)
        rc = self.L.tlb_edi_pft_host(self.h, a.ctypes.data, al.ctypes.data, nf, stride, pseq.ctypes.data, fec, chunk_len, 1 if transport else 0,
                                     addr_source, dest_port, frags.ctypes.data, flen.ctypes.data, nfrag.ctypes.data, mf.value, fs.value)
        if rc:
            raise ToolameError(rc, "tlb_edi_pft_host")
        return frags, flen, nfrag

    def last_kernel_ms(self):
        return float(self.L.tlb_last_kernel_ms(self.h))

    def last_stage_ms(self):
        """(psy-2 kernel ms, encode kernel ms) of the last launch of an all-model-2/4 batch, else None (models 1/3: one kernel; model 0: no psy kernel)"""
        a, b = C.c_float(0), C.c_float(0)
        if self.L.tlb_last_stage_ms(self.h, C.byref(a), C.byref(b)):
            return None
        return float(a.value), float(b.value)

    def reset(self):
        rc = self.L.tlb_reset(self.h)
        if rc:
            raise ToolameError(rc, "tlb_reset")
        self._first = True

    def close(self):
        if getattr(self, "h", None):
            self.L.tlb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# node level: every GPU of one host behind one handle (include/toolame_batch.h part 3, csrc/tlb_node.cpp)
class _CNodeConfig(C.Structure):
    _fields_ = [("plane", C.c_int), ("tick", _CTickConfig)]


class _CNodeShardInfo(C.Structure):
    _fields_ = [("shard", C.c_int), ("device", C.c_int), ("first", C.c_int), ("nstreams", C.c_int), ("state", C.c_int), ("last_err", C.c_int),
                ("failures", C.c_long), ("restarts", C.c_long), ("lost_steps", C.c_long), ("what", C.c_char * 192),
                ("device_name", C.c_char * 64), ("pci", C.c_char * 24), ("uuid", C.c_char * 36), ("num_cu", C.c_int), ("num_xcd", C.c_int),
                ("hbm_gb", C.c_double)]


class _CNodeCounter(C.Structure):
    _fields_ = [("shard", C.c_int), ("device", C.c_int), ("first", C.c_int), ("nstreams", C.c_int), ("steps", C.c_long), ("frames", C.c_long),
                ("busy_ns", C.c_double), ("device_ms", C.c_double), ("wall_ns", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def node_partition(nstreams, nshards):
    """[(first, n)] per shard: streams [g*N/G, (g+1)*N/G); pure arithmetic, no GPU"""
    L = load_library()
    out = []
    for g in range(nshards):
        f, n = C.c_int(0), C.c_int(0)
        L.tlb_node_partition(nstreams, nshards, g, C.byref(f), C.byref(n))
        out.append((f.value, n.value))
    return out


def node_plan(configs, nshards):
    """per shard: dict(first, n, nconfigs, lists = streams per kernel list (psy 0, 1, 2+4, 3), mono_pairs) -- what tlb_create() will make
    of each block; raises ToolameError on the first illegal configuration.  No GPU needed."""
    L = load_library()
    configs = list(configs)
    arr = _config_array(configs)
    out = []
    for g in range(nshards):
        f, n, nc, mp = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        ls = (C.c_int * 4)()
        rc = L.tlb_node_plan_shard(len(configs), arr, nshards, g, C.byref(f), C.byref(n), C.byref(nc), ls, C.byref(mp))
        if rc:
            raise ToolameError(rc, "tlb_node_plan_shard")
        out.append(dict(first=f.value, n=n.value, nconfigs=nc.value, lists=list(ls), mono_pairs=mp.value))
    return out


class Node:
    """tlb_node_*: N streams over the shards of one host (one tlb_tick or tlb_batch + one host thread per shard; `devices[g]` is the HIP
    device of shard g and may repeat).  plane "tick": fill pcm(s), run() / submit() + wait(), read frame(s) / packets(s) ...;
    plane "batch": device-resident buffers per shard, encode(pcm) with pcm int16 [nframes][nstreams][2][1152] split by the wrapper."""

    def __init__(self, configs, devices=(0,), plane="tick", egress="frames", ngroups=0, with_xpad=False, version=b"", now_s=1700000000,
                 delay_ms=0, tist=False, tai_utc_offset=37, fec=0, chunk_len=207, transport=False, addr_source=0, dest_port=0, lib=None):
        self.L = lib or load_library()
        configs = list(configs)
        self.configs = configs
        self.nstreams = len(configs)
        self.plane = plane
        self.with_xpad = bool(with_xpad)
        self._version = bytes(version)
        nc = _CNodeConfig()
        nc.plane = 0 if plane == "tick" else 1
        nc.tick = _CTickConfig(Tick.EGRESS[egress], ngroups, 1 if with_xpad else 0, self._version, len(self._version), int(now_s), int(delay_ms),
                               1 if tist else 0, int(tai_utc_offset), fec, chunk_len, 1 if transport else 0, addr_source, dest_port)
        devs = (C.c_int * len(devices))(*devices)
        err = C.c_int(0)
        self.h = self.L.tlb_node_create(len(devices), devs, self.nstreams, _config_array(configs), C.byref(nc), C.byref(err))
        if not self.h:
            raise ToolameError(err.value, "tlb_node_create")
        self.nshards = self.L.tlb_node_nshards(self.h)
        self.blocks = node_partition(self.nstreams, self.nshards)
        self._dev = {}          # BATCH plane: (shard, tag) -> (device pointer, bytes)
        if plane == "tick":
            self.units = [self.L.tlb_node_units(self.h, s) for s in range(self.nstreams)]

    def close(self):
        if self.h:
            for (g, _), (p, _) in list(self._dev.items()):
                self.L.tlb_node_device_free(self.h, g, p)
            self._dev.clear()
            self.L.tlb_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rc(self, rc, what):
        if rc:
            raise ToolameError(rc, what)

    # ---- health: fault isolation per shard (include/toolame_batch.h, FAULT ISOLATION) ----
    def describe(self):
        """one line per shard: device name, CUs, XCDs, memory, PCI address, UUID, stream block"""
        return self.L.tlb_node_describe(self.h).decode()

    def shard_status(self, g):
        info = _CNodeShardInfo()
        st = self.L.tlb_node_shard_status(self.h, g, C.byref(info))
        if st < 0:
            raise ToolameError(-st, "tlb_node_shard_status")
        d = {k: getattr(info, k) for k, _ in info._fields_}
        for k in ("what", "device_name", "pci", "uuid"):
            d[k] = d[k].decode(errors="replace")
        d["ok"] = st == 0
        return d

    def shard_ok(self, g):
        return self.L.tlb_node_shard_status(self.h, g, None) == 0

    def shard_restart(self, g, now_s=-1):
        self._rc(self.L.tlb_node_shard_restart(self.h, g, int(now_s)), "tlb_node_shard_restart")
        if self.plane == "tick":
            f, n = self.blocks[g]
            for s in range(f, f + n):
                self.units[s] = self.L.tlb_node_units(self.h, s)
        # (the wrapper's device buffers belong to the device, not to the shard's object: they stay)

    def fail_next(self, g, nth=1):
        """fault-injection TEST build only (lib=load_fault_library()): the nth launch / tick of shard g from now fails"""
        return self.L.tlb_debug_node_fail_next(self.h, g, nth)

    def counters(self):
        per = (_CNodeCounter * self.nshards)()
        tot = _CNodeCounter()
        self._rc(self.L.tlb_node_counters(self.h, per, C.byref(tot)), "tlb_node_counters")
        return [p.asdict() for p in per], tot.asdict()

    # ---- TICK plane ----
    def pcm(self, s):
        p = self.L.tlb_node_pcm(self.h, s)
        return np.ctypeslib.as_array((C.c_int16 * (2 * SAMPLES)).from_address(p)) if p else None

    def set_pcm(self, inter):
        """inter int16 [nstreams][2304]: every stream's interleaved frame into the current input set (block copies per shard)"""
        for g, (f, n) in enumerate(self.blocks):
            p = self.L.tlb_node_pcm(self.h, f)
            if not p:
                if not self.shard_ok(g):
                    continue                # a broken shard takes no input; its streams are off air until it is restarted
                raise ToolameError(18, "tlb_node_pcm: no input set is free (two ticks in flight)")
            np.ctypeslib.as_array((C.c_int16 * (n * 2 * SAMPLES)).from_address(p)).reshape(n, 2 * SAMPLES)[:] = inter[f:f + n]

    def set_xpad(self, s, rec, length):
        p, q = self.L.tlb_node_xpad(self.h, s), self.L.tlb_node_xpad_len(self.h, s)
        if not p or not q:
            return
        np.ctypeslib.as_array((C.c_uint8 * MAX_XPAD).from_address(p))[:] = rec
        C.c_int32.from_address(q).value = int(length)

    def submit(self):
        self._rc(self.L.tlb_node_submit(self.h), "tlb_node_submit")

    def wait(self):
        self._rc(self.L.tlb_node_wait(self.h), "tlb_node_wait")

    def run(self):
        self._rc(self.L.tlb_node_run(self.h), "tlb_node_run")

    def finish(self):
        self._rc(self.L.tlb_node_finish(self.h), "tlb_node_finish")

    def peaks(self, s):
        p = self.L.tlb_node_peaks(self.h, s)
        return tuple(np.ctypeslib.as_array((C.c_int16 * 2).from_address(p))) if p else None

    def silence_ms(self, s):
        return int(self.L.tlb_node_silence_ms(self.h, s))

    def frame(self, s):
        n = C.c_int(0)
        p = self.L.tlb_node_frame(self.h, s, C.byref(n))
        return C.string_at(p, n.value) if p and n.value else b""

    def _units(self, fn, s):
        out = []
        for u in range(self.units[s]):
            n = C.c_int(0)
            p = fn(self.h, s, u, C.byref(n))
            if p and n.value:
                out.append(C.string_at(p, n.value))
        return out

    def packets(self, s):
        return self._units(self.L.tlb_node_packet, s)

    def messages(self, s):
        return self._units(self.L.tlb_node_message, s)

    def fragments(self, s):
        out = []
        for u in range(self.units[s]):
            fr = []
            for k in range(self.L.tlb_node_fragments(self.h, s, u)):
                n = C.c_int(0)
                p = self.L.tlb_node_fragment(self.h, s, u, k, C.byref(n))
                fr.append(C.string_at(p, n.value))
            if fr:
                out.append(fr)
        return out

    # ---- both planes ----
    def set_gain_db(self, gain_db, stream=-1):
        self._rc(self.L.tlb_node_set_gain_db(self.h, stream, float(gain_db)), "tlb_node_set_gain_db")

    def stream_reset(self, s):
        self._rc(self.L.tlb_node_stream_reset(self.h, s), "tlb_node_stream_reset")

    def stream_finish(self, s):
        buf = (C.c_uint8 * 2048)()
        n = self.L.tlb_node_stream_finish(self.h, s, buf, 2048)
        if n < 0:
            raise ToolameError(-n, "tlb_node_stream_finish")
        return bytes(buf[:n])

    def stream_reconfigure(self, s, config):
        self._rc(self.L.tlb_node_stream_reconfigure(self.h, s, _config_array([config])), "tlb_node_stream_reconfigure")
        if self.plane == "tick":
            self.units[s] = self.L.tlb_node_units(self.h, s)

    # ---- BATCH plane ----
    def out_stride(self, g):
        return self.L.tlb_out_stride(self.L.tlb_node_batch(self.h, g))

    def _buf(self, g, tag, nbytes):
        cur = self._dev.get((g, tag))
        if cur and cur[1] >= nbytes:
            return cur[0]
        if cur:
            self.L.tlb_node_device_free(self.h, g, cur[0])
        p = self.L.tlb_node_device_alloc(self.h, g, nbytes)
        if not p:
            raise ToolameError(17, "tlb_node_device_alloc")
        self._dev[(g, tag)] = (p, nbytes)
        return p

    def upload(self, pcm, slot=0):
        """pcm int16 [nframes][nstreams][2][1152] -> one device buffer per shard (the shard's streams, [nframes][n_g][2][1152]);
        `slot` names the buffer set (a caller alternating two resident inputs uploads slot 0 and slot 1 once)"""
        nf = pcm.shape[0]
        ptrs = (C.c_void_p * self.nshards)()
        for g, (f, n) in enumerate(self.blocks):
            blk = np.ascontiguousarray(pcm[:, f:f + n], dtype=np.int16)
            ptrs[g] = self._buf(g, ("pcm", slot), blk.nbytes)
            self._rc(self.L.tlb_node_copy_in(self.h, g, ptrs[g], blk.ctypes.data, blk.nbytes), "tlb_node_copy_in")
        self._pcm_ptrs = getattr(self, "_pcm_ptrs", {})
        self._pcm_ptrs[slot] = (ptrs, nf)
        return ptrs

    def encode_resident(self, slot=0, want_len=True):
        """queue one encode call on every shard over the PCM uploaded as `slot`; returns at once (sync() waits)"""
        ptrs, nf = self._pcm_ptrs[slot]
        outs, lens = (C.c_void_p * self.nshards)(), (C.c_void_p * self.nshards)()
        for g, (f, n) in enumerate(self.blocks):
            if not self.shard_ok(g):
                continue                    # (the node skips a broken shard: its array elements are not read)
            outs[g] = self._buf(g, "out", nf * n * self.out_stride(g))
            lens[g] = self._buf(g, "len", nf * n * 4)
        self._out_ptrs, self._len_ptrs, self._nf, self._have_len = outs, lens, nf, bool(want_len)
        self._rc(self.L.tlb_node_encode_device(self.h, ptrs, nf, None, None, outs, lens if want_len else None), "tlb_node_encode_device")

    def sync(self):
        self._rc(self.L.tlb_node_sync(self.h), "tlb_node_sync")

    def download(self, nframes=None):
        """-> per stream: the bytes of output slots 0..nframes-1 concatenated (slot lengths from the _len variant)"""
        if not getattr(self, "_have_len", True):
            raise ToolameError(18, "Node.download(): the last encode_resident() was queued with want_len=False -- no slot lengths to cut the frames by")
        nf = nframes or self._nf
        out = [b""] * self.nstreams
        for g, (f, n) in enumerate(self.blocks):
            if not self.shard_ok(g):
                continue                    # nothing of a broken shard is read (its buffers hold a half-finished step)
            st = self.out_stride(g)
            fr = np.empty((nf, n, st), dtype=np.uint8)
            ln = np.empty((nf, n), dtype=np.int32)
            self._rc(self.L.tlb_node_copy_out(self.h, g, fr.ctypes.data, self._out_ptrs[g], fr.nbytes), "tlb_node_copy_out")
            self._rc(self.L.tlb_node_copy_out(self.h, g, ln.ctypes.data, self._len_ptrs[g], ln.nbytes), "tlb_node_copy_out")
            for k in range(n):
                out[f + k] = b"".join(fr[i, k, :ln[i, k]].tobytes() for i in range(nf))
        return out

    def encode(self, pcm):
        """upload + encode + sync + download (the convenience path of the parity tests)"""
        self.upload(pcm)
        self.encode_resident()
        self.sync()
        return self.download()

    def flush(self):
        """the pending (last) frame of every stream"""
        outs, lens = (C.c_void_p * self.nshards)(), (C.c_void_p * self.nshards)()
        for g, (f, n) in enumerate(self.blocks):
            if not self.shard_ok(g):
                continue
            outs[g] = self._buf(g, "fout", n * self.out_stride(g))
            lens[g] = self._buf(g, "flen", n * 4)
        self._rc(self.L.tlb_node_flush_device(self.h, outs, lens), "tlb_node_flush_device")
        self.sync()
        out = [b""] * self.nstreams
        for g, (f, n) in enumerate(self.blocks):
            if not self.shard_ok(g):
                continue
            st = self.out_stride(g)
            fr = np.empty((n, st), dtype=np.uint8)
            ln = np.empty((n,), dtype=np.int32)
            self._rc(self.L.tlb_node_copy_out(self.h, g, fr.ctypes.data, outs[g], fr.nbytes), "tlb_node_copy_out")
            self._rc(self.L.tlb_node_copy_out(self.h, g, ln.ctypes.data, lens[g], ln.nbytes), "tlb_node_copy_out")
            for k in range(n):
                out[f + k] = fr[k, :ln[k]].tobytes()
        return out
