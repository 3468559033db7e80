"""The tests' own CRC-16 parser against the reference's frames (CPU)."""
import numpy as np

from conftest import golden_cases
from framecheck import crc16_frame_ok as _crc16_frame_ok


def test_crc16_parser_against_golden_taps():
    """the test's own CRC parser, checked against the reference's frames (all golden 48 kHz table-0 stereo cases verify)"""
    n = 0
    for p in golden_cases():
        g = np.load(p)
        fs, mode, kbps, psy, kind, seed, pad_len, nframes = (int(v) for v in g["cfg"])
        if fs != 48000 or chr(mode) not in "sj" or kbps not in (128, 192):
            continue
        data, fb = g["data"].tobytes(), 3 * kbps
        for f in range(nframes):
            assert _crc16_frame_ok(data[fb * f: fb * (f + 1)]), (p.stem, f)
            n += 1
    assert n > 500
