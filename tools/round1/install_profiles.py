#!/usr/bin/env python3
"""Copy the summaries a profiling round left in gpurun_out/ (tools/profile_round.sh V, tools/pmc_sq.sh V, tools/edi_bench.py)
into profiles/ under r01_<V>_* names, replacing the previous version's files, and derive the PMC json files bench.py reads.
usage: tools/install_profiles.py V [OLD_V]"""
import csv
import glob
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / "gpurun_out", ROOT / "profiles"
V = sys.argv[1]
OLD = sys.argv[2] if len(sys.argv) > 2 else None
ROUND = "r01"

if OLD:
    for f in P.glob(f"*{OLD}*"):
        subprocess.run(["git", "rm", "-q", "--cached", str(f)], cwd=ROOT, capture_output=True)
        f.unlink(missing_ok=True)


def cp(src, dst):
    if Path(src).exists():
        shutil.copy(src, P / dst)
    else:
        print("missing", src)


cp(G / f"{ROUND}_{V}_bench_kernel_stats.csv", f"{ROUND}_{V}_bench_kernel_stats.csv")
for f in G.glob(f"stage_{ROUND}_{V}_psy*.txt"):
    cp(f, f.name)
for f in G.glob(f"bench_{ROUND}_{V}_*.json"):
    cp(f, f.name)
bench = json.loads(open(G / f"bench_{V}_default.json").read().strip().splitlines()[-1])
json.dump(bench, open(P / f"{ROUND}_{V}_bench_default.json", "w"))
cp(G / f"edi_bench_{V}.txt", f"{ROUND}_{V}_edi_kernels.txt")

pmc = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(str(G / f"pmc_{V}_{name}" / "**" / "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        shutil.copy(f, P / f"{ROUND}_{V}_pmc_{name}_counter_collection.csv")
        for row in csv.DictReader(open(f)):
            if "tl_encode_kernel" in row.get("Kernel_Name", "") and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    pmc[name] = sum(vals) / len(vals) if vals else None
if pmc["FETCH_SIZE"] and pmc["WRITE_SIZE"]:
    f, w = pmc["FETCH_SIZE"], pmc["WRITE_SIZE"]
    d = {"kernel": "tl_encode_kernel<1>", "version": f"{ROUND} {V} ({bench['value'] / 1e6:.2f} M frames/s)",
         "workload": {"streams": 4096, "frames_per_step": 8, "psy": 1, "mode": "s"},
         "FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
         "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
         "hbm_bytes_per_launch": int(round((2 * f + w) * 1024, -3)), "algorithmic_bytes_per_launch": 163577856,
         "note": "fetch (corrected) = PCM 151 MB + pending-frame reads 12.6 MB + state/tables; the psy-stage and history re-reads of "
                 "the same PCM hit L2. writes = frames 12.6 MB + pending frames 12.6 MB + PCM history/state.",
         "source": f"tools/pmc_traffic.sh via tools/profile_round.sh {V} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes); "
                   f"raw CSVs: profiles/{ROUND}_{V}_pmc_*"}
    json.dump(d, open(P / "pmc_traffic_latest.json", "w"), indent=1)
    json.dump(d, open(P / f"{ROUND}_pmc_traffic_{V}.json", "w"), indent=1)
    print("hbm bytes/launch", d["hbm_bytes_per_launch"], "= %.2f x algorithmic" % (d["hbm_bytes_per_launch"] / 163577856))

acc = {}
for f in glob.glob(str(G / f"sq_{V}_*" / "**" / "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "tl_encode_kernel" in row.get("Kernel_Name", ""):
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
if acc:
    sq = {k: int(sum(v) / len(v)) for k, v in sorted(acc.items())}
    frames = 4096 * 8
    wc = sq["SQ_WAVE_CYCLES"]
    sq["source"] = f"tools/pmc_sq.sh {V} (rocprofv3 --pmc, two passes), bench.py --steps 3: per launch of 4096 streams x 8 frames, psy 1"
    sq["derived"] = {"per_frame": {"valu": round(sq["SQ_INSTS_VALU"] / frames), "salu": round(sq["SQ_INSTS_SALU"] / frames),
                                   "lds": round(sq["SQ_INSTS_LDS"] / frames), "vmem": round(sq["SQ_INSTS_VMEM"] / frames)},
                     "issuing_share_of_wave_cycles": round(sq["SQ_ACTIVE_INST_ANY"] / wc, 3),
                     "waiting_share": round(sq["SQ_WAIT_ANY"] / wc, 3),
                     "valu_busy_per_wave": round(sq["SQ_ACTIVE_INST_VALU"] / wc, 3),
                     "valu_busy_per_simd_at_2_waves": round(2 * sq["SQ_ACTIVE_INST_VALU"] / wc, 3),
                     "lds_conflict_share_of_lds_active": round(sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"], 3)}
    json.dump(sq, open(P / f"{ROUND}_{V}_sq_counters.json", "w"), indent=1)
    json.dump(sq, open(P / "sq_counters_latest.json", "w"), indent=1)
    print(json.dumps(sq["derived"]))
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step")}), bench["roofline"]["kernel_ms"], bench.get("cpu_baseline", {}))
