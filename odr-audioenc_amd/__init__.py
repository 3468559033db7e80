"""odr-audioenc_amd -- MI355X-native batched DAB MP2 (MPEG-1/2 Layer II) encode path.

A drop-in for ONE hot path of Opendigitalradio/ODR-AudioEnc: libtoolame-dab's per-frame encoder
(toolame_encode_frame(), libtoolame-dab/toolame.c:267-554), re-designed as hand-written HIP kernels
for gfx950 behind the reference's own C-ABI plus a batched, handle-based API
(include/toolame_batch.h).  Python is only the thin binding used by tests and bench.py; the product
is csrc/ (kernels + C-ABI, built into libtoolame_dab_hip.so).

There is NO CPU fallback: importing works anywhere, but creating an encoder raises if the HIP
library is missing or no GPU is visible.
"""
from .toolame import (  # noqa: F401
    Batch, StreamConfig, ToolameError, LIB_PATH, build, load_library, lds_bytes_per_stream, legacy_api,
    EDI_STATE_DTYPE, edi_state_init, Tick, Node, node_partition, node_plan, load_fault_library, FAULT_LIB_PATH,
)
