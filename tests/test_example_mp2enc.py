"""examples/mp2enc.cpp -- a file-to-file encoder in plain C++ over the batched C-ABI (ingest -> encode -> whole frames): it must
build against include/toolame_batch.h with a host compiler alone, and on the GPU its output must be the oracle's bytes."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oraclelib as O
from pcmgen import gen_pcm


def build(tmp_path):
    import odr_audioenc_amd as M
    M.build()
    exe = tmp_path / "mp2enc"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", str(ROOT / "examples" / "mp2enc.cpp"), "-I" + str(ROOT / "include"),
                    "-L" + str(ROOT / "odr-audioenc_amd"), "-ltoolame_dab_hip", "-Wl,-rpath," + str(ROOT / "odr-audioenc_amd"), "-o", str(exe)], check=True)
    return exe


def test_example_builds_with_a_host_compiler(tmp_path):
    assert build(tmp_path).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("fs,channels,mode,kbps,psy,nstreams", [(48000, 2, "j", 128, 1, 1), (24000, 1, "m", 64, 3, 5), (48000, 2, "s", 192, 0, 3)])
def test_example_output_equals_oracle(tmp_path, fs, channels, mode, kbps, psy, nstreams):
    exe = build(tmp_path)
    nframes = 300                                            # more than one 256-frame call
    pcm = gen_pcm(77, 0, 0, nframes)                         # [nframes][2][1152]
    inter = pcm[:, :channels].transpose(0, 2, 1).reshape(-1).astype("<i2")      # L R L R ... (mono: L only)
    (tmp_path / "in.pcm").write_bytes(inter.tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "in.pcm"), str(tmp_path / "out.mp2"), "-r", str(fs), "-c", str(channels), "-b", str(kbps),
                        "-m", mode, "-p", str(psy), "-n", str(nstreams)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref, _ = O.oracle_stream(pcm if channels == 2 else pcm[:, :1].repeat(2, axis=1), samplerate=fs, mode=mode, kbps=kbps, psy=psy)
    assert (tmp_path / "out.mp2").read_bytes() == ref
