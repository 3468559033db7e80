"""A slice of tools/fuzz_oracle_vs_ref.py in the CPU suite: the oracle against the LIVE reference (oracle/_ref, the reference's own
sources compiled here) on random legal configurations x all psy models x all signal kinds, plus a slice of the impulse / square-wave
sweep.  Skipped where the reference build cannot exist (the GPU box: /root/reference does not travel)."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import oraclelib as O


@pytest.fixture(scope="module")
def Fz():
    if not O.REF_SO.exists():
        pytest.skip("oracle/_ref/libtoolame_ref.so not built here")
    import fuzz_oracle_vs_ref as F
    O.build_oracle()
    return F


def test_fresh_library_instance_per_stream_equals_fresh_process(Fz):
    """the fuzz loads and unloads the reference once per stream instead of starting a process per stream: same bytes, same burst lengths"""
    from pcmgen import gen_pcm
    for job in [(48000, "j", 192, 2, 0, 7, 14, 0), (22050, "m", 32, 3, 4, 8, 14, 0), (48000, "s", 128, 4, 2, 9, 14, 0), (32000, "d", 256, 1, 7, 10, 14, 0),
                (48000, "s", 192, 1, 0, 11, 14, 196), (24000, "m", 32, 2, 5, 12, 14, 40)]:
        fs, mode, kbps, psy, kind, seed, F, pad_len = job
        a = Fz.one_reference(job)
        b = O.reference_stream(gen_pcm(seed, kind, 0, F), samplerate=fs, mode=mode, kbps=kbps, psy=psy, pad_len=pad_len, xpads=Fz.xpads_for(seed, pad_len, F))
        assert a[0] == b["data"] and list(a[1]) == list(b["lens"]), job


def test_oracle_equals_live_reference_on_random_streams(Fz):
    jobs = Fz.random_jobs(200, 10, seed=20261003) + Fz.sweep_jobs(10)[::9]
    assert len(jobs) == 200 + 128
    bad = Fz.run(jobs, 4)
    assert not bad, bad[:5]


def test_oracle_equals_live_reference_with_random_xpad(Fz):
    """the X-PAD / `adb` path (toolame.c:301,515-551) against the LIVE reference, not only the six goldens -- every
    stream a random legal PAD length (2..255), every frame a random X-PAD length in {0, 2..pad_len} with random bytes, all models."""
    jobs = Fz.random_jobs(64, 10, seed=20261004, xpad_share=1.0)
    assert len(jobs) == 64 and all(j[7] >= 2 for j in jobs) and len({j[7] for j in jobs}) > 8
    assert any(j[7] == 255 for j in jobs) or any(j[7] >= 196 for j in jobs)
    bad = Fz.run(jobs, 4)
    assert not bad, bad[:5]
