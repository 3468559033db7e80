"""The C-ABI library loads without a GPU and exports every function include/toolame_batch.h declares (the nine reference
symbols of libtoolame-dab.sym among them); creating an encoder without a GPU fails loudly -- there is no CPU fallback."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF_SYMS = ["toolame_init", "toolame_finish", "toolame_enable_byteswap", "toolame_set_channel_mode", "toolame_set_psy_model",
            "toolame_set_bitrate", "toolame_set_samplerate", "toolame_set_pad", "toolame_encode_frame"]   # libtoolame-dab.sym


def _declared():
    src = (ROOT / "include" / "toolame_batch.h").read_text()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:toolame|tlb)_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_reference_abi():
    names = _declared()
    for s in REF_SYMS:
        assert s in names


def test_library_exports_every_declared_symbol():
    import odr_audioenc_amd as M
    if not M.LIB_PATH.exists():
        M.build()
    lib = C.CDLL(str(M.LIB_PATH))
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    import torch
    import odr_audioenc_amd as M
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(M.ToolameError) as e:
        M.Batch([M.StreamConfig()])
    assert e.value.code == 16


def test_product_does_not_reference_oracle():
    """nothing under the product package includes, links or loads oracle/ or tests/emu"""
    for p in (ROOT / "odr-audioenc_amd").rglob("*"):
        if p.suffix in (".h", ".hip", ".cpp", ".py", ".inc") or p.name == "Makefile":
            t = p.read_text(errors="ignore")
            assert "oracle/" not in t.replace("oracle/mp2_oracle.c:psy3_run", "") and "mp2_emu" not in t and "libmp2oracle" not in t, p
