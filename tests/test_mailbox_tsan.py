"""ThreadSanitizer on the node level's threading primitive (csrc/tlb_mailbox.h), CPU only: the mailbox a shard of csrc/tlb_node.cpp IS,
driven with fake jobs in the node's own call patterns (tests/emu/mailbox_tsan.cpp).  Sanitizers belong on the CPU build -- the GPU box is
never asked for them."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_shard_mailbox_is_clean_under_threadsanitizer(tmp_path):
    exe = tmp_path / "mailbox_tsan"
    src = ROOT / "tests" / "emu" / "mailbox_tsan.cpp"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "3000"], capture_output=True, text=True, timeout=300, env={"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"})
    assert r.returncode == 0 and "mailbox ok" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout, r.stderr[-2000:])


def test_node_source_uses_the_tested_mailbox():
    """the Shard of csrc/tlb_node.cpp derives from the header the sanitizer run drives -- not from a private copy"""
    text = (ROOT / "odr-audioenc_amd" / "csrc" / "tlb_node.cpp").read_text()
    assert '#include "tlb_mailbox.h"' in text and "struct Shard : TlbMailbox" in text
    assert "std::condition_variable cv;" not in text
