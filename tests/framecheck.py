"""Bit-level checks of MPEG-1 Layer II frames that take nothing from the encoder but the bytes."""
import numpy as np

# ---- CRC-16 of a frame, recomputed from its bytes (crc.c:12-56): header bits 16..31, then the bit_alloc and scfsi fields ----
_NBAL_TAB0 = [4, 4, 4] + [4] * 8 + [3] * 12 + [2] * 4                 # alloc table 0 (48 kHz, >= 56 kbps/ch), sblimit 27


def crc16_frame_ok(frame, nbal=_NBAL_TAB0, nch=2):
    """True when the CRC-16 stored in bytes 4..5 equals the one computed over the protected fields of this frame (stereo or
    joint stereo, table 0).  A bit-level parser: nothing is taken from the encoder but the bytes."""
    bits = np.unpackbits(np.frombuffer(frame, dtype=np.uint8))
    mode, mode_ext = (frame[3] >> 6) & 3, (frame[3] >> 4) & 3
    jsbound = 4 * (mode_ext + 1) if mode == 1 else len(nbal)
    pos, alloc = 48, []
    for sb, nb in enumerate(nbal):
        for ch in range(nch if sb < jsbound else 1):
            v = int("".join(map(str, bits[pos:pos + nb])), 2)
            pos += nb
            alloc.append((sb, ch, v))
            if sb >= jsbound:
                alloc.append((sb, 1, v))
    alloc.sort()
    nsel = sum(1 for _, _, v in alloc if v)
    prot = np.concatenate([bits[16:32], bits[48:pos + 2 * nsel]])
    crc = 0xffff
    for bit in prot:
        carry = (crc >> 15) & 1
        crc = (crc << 1) & 0xffff
        if carry ^ int(bit):
            crc ^= 0x8005
    return crc == ((frame[4] << 8) | frame[5])


