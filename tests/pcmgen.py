"""Integer-only synthetic PCM shared by the oracle (oracle/mp2_oracle.c:gen_sample), the tests and
bench.py.  No libm, so numpy / C / torch produce identical int16 samples.

kind: 0 tones+noise, 1 silence, 2 full-scale square, 3 impulse, 4 full-scale noise,
      5 channel-identical tones, 6 low-level (+-1 LSB) noise, 7 kind-0 with a stepped envelope.
"""
import numpy as np

U32 = np.uint32


def _mix32(x):
    x = x.astype(U32, copy=True)
    x ^= x >> U32(16)
    x *= U32(0x7FEB352D)
    x ^= x >> U32(15)
    x *= U32(0x846CA68B)
    x ^= x >> U32(16)
    return x


def _tri16(phase):
    p = (phase & U32(0xFFFF)).astype(np.int64)
    t = np.where(p < 32768, p, 65535 - p)
    return t - 16384


def _asr(v, s):
    return np.floor_divide(v, np.int64(1) << np.int64(s))


def gen_pcm(seed, kind, frame, nframes=1):
    """-> int16 array [nframes, 2, 1152] (planar per frame), frames frame..frame+nframes-1."""
    with np.errstate(over="ignore"):
        n = (np.arange(nframes * 1152, dtype=np.uint64) + np.uint64(frame) * np.uint64(1152)).astype(U32)
        out = np.zeros((2, nframes * 1152), dtype=np.int64)
        seed = U32(seed & 0xFFFFFFFF)
        for ch in range(2):
            c = U32(0 if kind == 5 else ch)
            h = _mix32((n * U32(0x9E3779B1)) ^ (seed * U32(0x85EBCA6B) + c * U32(0xC2B2AE35) + U32(0x165667B1)))
            if kind == 1:
                v = np.zeros_like(n, dtype=np.int64)
            elif kind == 2:
                P = U32(2 + int(seed) % 63)
                v = np.where(((n // P) & U32(1)) != 0, 32767, -32768).astype(np.int64)
            elif kind == 3:
                v = np.where((n % U32(5000)) == 100, 32767, 0).astype(np.int64)
            elif kind == 4:
                v = (h & U32(0xFFFF)).astype(np.int64) - 32768
            elif kind == 6:
                v = (h % U32(3)).astype(np.int64) - 1
            else:
                one = np.ones(1, dtype=U32)
                s0 = U32(150) + _mix32(one * (seed * U32(3) + U32(1)))[0] % U32(3000)
                s1 = U32(3000) + _mix32(one * (seed * U32(3) + U32(2)))[0] % U32(9000) + c * U32(37)
                s2 = U32(12000) + _mix32(one * (seed * U32(3) + U32(3)))[0] % U32(10000)
                v = (_asr(_tri16(n * s0 + c * U32(9000)), 1) + _asr(_tri16(n * s1), 2)
                     + _asr(_tri16(n * s2 + c * U32(20000)), 3))
                v = v + (h & U32(0x1FFF)).astype(np.int64) - 4096
                if kind == 7:
                    sh = ((n >> U32(9)) % U32(8)).astype(np.int64)
                    v = np.floor_divide(v, np.int64(1) << sh)
            out[ch] = np.clip(v, -32768, 32767)
    return np.ascontiguousarray(out.reshape(2, nframes, 1152).transpose(1, 0, 2)).astype(np.int16)
