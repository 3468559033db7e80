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


def build(tmp_path, name="mp2enc"):
    import odr_audioenc_amd as M
    M.build()
    exe = tmp_path / name
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", str(ROOT / "examples" / (name + ".cpp")), "-I" + str(ROOT / "include"),
                    "-L" + str(ROOT / "odr-audioenc_amd"), "-ltoolame_dab_hip", "-Wl,-rpath," + str(ROOT / "odr-audioenc_amd"), "-o", str(exe)], check=True)
    return exe


def test_example_builds_with_a_host_compiler(tmp_path):
    assert build(tmp_path).exists()
    assert build(tmp_path, "editick").exists()               # the tick API (tlb_tick_*) from plain C++
    assert build(tmp_path, "nodetick").exists()              # the node level (tlb_node_*): a fleet of services over the host's GPUs


@pytest.mark.gpu
@pytest.mark.parametrize("fs,channels,mode,kbps,psy,nstreams", [(48000, 2, "j", 128, 1, 3), (24000, 1, "m", 64, 1, 2)])
def test_editick_packets_equal_the_binding(tmp_path, fs, channels, mode, kbps, psy, nstreams):
    """examples/editick.cpp (tlb_tick_* from C++: PCIe in, ingest, encode, EDI AF, PCIe out per tick) writes the same AF packets
    for stream 0 as the Python binding's Tick object driven with the same PCM."""
    import struct
    import odr_audioenc_amd as M
    exe = build(tmp_path, "editick")
    nframes = 40
    pcm = gen_pcm(91, 0, 0, nframes)
    inter = pcm[:, :channels].transpose(0, 2, 1).reshape(nframes, -1).astype("<i2")
    (tmp_path / "in.pcm").write_bytes(inter.tobytes())
    r = subprocess.run([str(exe), str(tmp_path / "in.pcm"), str(tmp_path / "out.af"), "-r", str(fs), "-c", str(channels), "-b", str(kbps),
                        "-m", mode, "-p", str(psy), "-n", str(nstreams), "-t", "1712345678"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    blob, got, o = (tmp_path / "out.af").read_bytes(), [], 0
    while o < len(blob):
        n = struct.unpack_from("<I", blob, o)[0]
        got.append(blob[o + 4:o + 4 + n])
        o += 4 + n
    t = M.Tick([M.StreamConfig(samplerate=fs, mode=mode, bitrate=kbps, psy_model=psy)] * nstreams, egress="af", version=b"editick example",
               now_s=1712345678, delay_ms=0, tist=True, tai_utc_offset=37)
    want = []
    for f in range(nframes):
        t.pcm[:, :inter.shape[1]] = inter[f]
        t.run()
        want += t.packets(0)
    t.finish()
    want += t.packets(0)
    t.close()
    upf = 2 if fs == 24000 else 1
    assert got == want and len(got) == nframes * upf and all(p[:2] == b"AF" for p in got)


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


@pytest.mark.gpu
def test_nodetick_fleet_two_shards_equal_the_binding(tmp_path):
    """examples/nodetick.cpp: 37 services over two shards on this box's GPU ("-d 0,0"), ticks overlapped, fill and ship on the shards'
    own threads.  The AF packets of the LAST service (it lives in shard 1) equal those of a single Tick object fed the same PCM, the
    node's counters add up, and a run with ONE shard ships exactly as many packets and bytes."""
    import json
    import struct
    import odr_audioenc_amd as M
    exe = build(tmp_path, "nodetick")
    nin, ns, ticks = 60, 37, 25
    pcm = gen_pcm(123, 0, 0, nin)
    inter = pcm.transpose(0, 2, 1).reshape(nin, 2304).astype("<i2")
    (tmp_path / "in.pcm").write_bytes(inter.tobytes())
    outs = {}
    for d in ("0,0", "0"):
        r = subprocess.run([str(exe), str(tmp_path / "in.pcm"), "-n", str(ns), "-d", d, "-k", str(ticks), "-p", "3", "-o", str(tmp_path / f"out_{len(d)}.af")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[d] = json.loads(r.stdout.strip().splitlines()[-1])
    two, one = outs["0,0"], outs["0"]
    assert two["shards"] == 2 and one["shards"] == 1 and two["frames"] == one["frames"] == ns * ticks
    assert two["packets"] == one["packets"] == ns * ticks and two["bytes"] == one["bytes"]
    blob, got, o = (tmp_path / "out_3.af").read_bytes(), [], 0
    while o < len(blob):
        n = struct.unpack_from("<I", blob, o)[0]
        got.append(blob[o + 4:o + 4 + n])
        o += 4 + n
    assert (tmp_path / "out_1.af").read_bytes() == blob
    t = M.Tick([M.StreamConfig(mode="j", bitrate=128, psy_model=3)], egress="af", version=b"nodetick example", now_s=1712345678, delay_ms=0, tist=True, tai_utc_offset=37)
    want = []
    for f in range(ticks):
        t.pcm[0] = inter[(ns - 1 + f) % nin]
        t.run()
        want += t.packets(0)
    t.finish()
    want += t.packets(0)
    t.close()
    assert got == want and len(got) == ticks
