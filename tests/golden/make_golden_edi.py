#!/usr/bin/env python3
"""Golden vectors for the EDI AF-packet step, produced by the reference's own classes (oracle/_ref/libedi_ref.so =
contrib/edioutput/{TagItems,TagPacket,AFPacket}.cpp + contrib/crc.c + oracle/edi_ref_driver.cpp).
Run here (needs /root/reference); writes tests/golden/edi_cases.npz: per case the first 16 packets in full, all packet
lengths, a sha256 over every packet and the final sender state."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import edilib as E

out = {}
for name, *_ in E.CASES:
    frames, levels, fb, st = E.case_inputs(name)
    pkts, plen, st2 = E.ref_af(frames, levels, fb, st, unit_bytes=E.case_unit_bytes(name))
    out[name + "_head"] = pkts[:16]
    out[name + "_len"] = plen
    out[name + "_sha"] = np.frombuffer(bytes.fromhex(E.digest(pkts, plen)), dtype=np.uint8)
    out[name + "_state"] = st2
    print(name, pkts.shape, plen.min(), plen.max(), E.digest(pkts, plen)[:16])
np.savez_compressed(Path(__file__).resolve().parent / "edi_cases.npz", **out)

# ---- PFT layer: reference Reed-Solomon/CRC code under the restated PFT.cpp logic (oracle/pft_ref_driver.cpp) ----
pft = {}
for name, *_ in E.PFT_CASES:
    af, af_len, pseq, kw = E.pft_case_inputs(name)
    frags, flen, nfrag, ps = E.ref_pft(af, af_len, pseq, **kw)
    pft[name + "_head"] = frags[:2]
    pft[name + "_len"] = flen
    pft[name + "_n"] = nfrag
    pft[name + "_pseq"] = ps
    pft[name + "_sha"] = np.frombuffer(bytes.fromhex(E.pft_digest(frags, flen, nfrag)), dtype=np.uint8)
    print(name, frags.shape, int(nfrag.min()), int(nfrag.max()), E.pft_digest(frags, flen, nfrag)[:16])
np.savez_compressed(Path(__file__).resolve().parent / "edi_pft_cases.npz", **pft)
