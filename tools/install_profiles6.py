#!/usr/bin/env python3
"""Copy the summaries tools/profile_round6.sh TAG left in gpurun_out/ into profiles/ (tracked) and derive the JSON files bench.py
quotes by workload (profiles/LATEST names the tag): HBM traffic per launch and kernel (FETCH x 2 + WRITE) and the SQ counters with
their derived shares, for configs[1], configs[2] / one GPU of configs[3], psy 2, configs[4] with psy 4 and psy 2, mono pairs.
Round 5 adds: the instruction classes per frame and the lane occupancy (fourth counter pass of tools/pmc_split.sh), which bench.py's
`roofline.valu_issue_utilisation` quotes, and the psy-2 tick shapes (16384 streams x 1 frame per launch).
usage: tools/install_profiles6.py TAG"""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / "gpurun_out", ROOT / "profiles"
V = sys.argv[1]
KERNELS = ("tl_frame_kernel", "tl_psy2_kernel", "tl_main_kernel", "tl_finish_kernel")


def cp(src, dst=None):
    src = Path(src)
    if src.exists():
        shutil.copy(src, P / (dst or src.name))
    else:
        print("missing", src)


def algo(wl):
    """SURVEY 8(d) bytes per launch of a workload"""
    frames = wl["streams"] * wl["frames_per_step"]
    if wl.get("mixed"):
        return frames // 2 * ((1152 * 2 + 144000 * 64 // 32000) + (2 * 1152 * 2 + 144000 * 192 // 48000))
    nch = 1 if wl["mode"] == "m" else 2
    return frames * (nch * 1152 * 2 + 144000 * 128 // 48000)


WORKLOADS = {
    "hl": {"streams": 4096, "frames_per_step": 32, "psy": 1, "mode": "s"},
    "psy3": {"streams": 16384, "frames_per_step": 8, "psy": 3, "mode": "s"},
    "psy2": {"streams": 4096, "frames_per_step": 32, "psy": 2, "mode": "s"},
    "cfg4psy4": {"streams": 16384, "frames_per_step": 8, "psy": 4, "mode": "s", "mixed": True},
    "cfg4psy2": {"streams": 16384, "frames_per_step": 8, "psy": 2, "mode": "s", "mixed": True},
    "mono": {"streams": 4096, "frames_per_step": 32, "psy": 1, "mode": "m"},
    "tick2": {"streams": 16384, "frames_per_step": 1, "psy": 2, "mode": "s"},
    "tick3": {"streams": 16384, "frames_per_step": 1, "psy": 3, "mode": "s"},
    "cfg4tick": {"streams": 16384, "frames_per_step": 1, "psy": 4, "mode": "s", "mixed": True},
}

cp(G / f"{V}_bench_kernel_stats.csv")
for f in (list(G.glob(f"stage_{V}_psy*.txt")) + list(G.glob(f"ab_{V}*.txt")) + list(G.glob(f"pmc_quick_{V}*.txt")) + list(G.glob(f"bench_{V}_*.json")) + list(G.glob(f"sq_{V}_*.txt")) + list(G.glob(f"class_budget_{V}*.txt")) + list(G.glob(f"gputests_{V}_final.log")) +
          list(G.glob(f"classes_{V}_*.json")) + list(G.glob(f"legacy_latency_{V}.txt"))):
    cp(f)
bench = json.loads(open(G / f"bench_{V}_default.json").read().strip().splitlines()[-1])
json.dump(bench, open(P / f"{V}_bench_default.json", "w"))


def traffic(name):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = {}
        for f in glob.glob(str(G / f"pmc_{V}_{name}_{c}" / "**" / "*counter_collection.csv"), recursive=True):
            shutil.copy(f, P / f"{V}_pmc_{name}_{c}_counter_collection.csv")
            for row in csv.DictReader(open(f)):
                for k in KERNELS:
                    if k in row.get("Kernel_Name", "") and row["Counter_Name"] == c:
                        acc.setdefault(k, []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {})[c + "_KB_per_launch"] = round(sum(v) / len(v), 1)
    tot = 0
    for k, d in out.items():
        d["hbm_bytes_per_launch"] = int(round((2 * d.get("FETCH_SIZE_KB_per_launch", 0) + d.get("WRITE_SIZE_KB_per_launch", 0)) * 1024, -3))
        tot += d["hbm_bytes_per_launch"]
    return out, tot


for name, wl in WORKLOADS.items():
    k, t = traffic(name)
    if not k:
        print("no traffic data for", name)
        continue
    a = algo(wl)
    d = {"version": V, "workload": wl, "kernels": k, "hbm_bytes_per_launch": t, "algorithmic_bytes_per_launch": a, "ratio_to_algorithmic": round(t / a, 3),
         "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
         "source": f"tools/profile_round6.sh {V} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, kernel trace only); raw CSVs: profiles/{V}_pmc_{name}_*"}
    json.dump(d, open(P / (f"{V}_pmc_traffic.json" if name == "hl" else f"{V}_pmc_traffic_{name}.json"), "w"), indent=1)
    print("traffic", name, t, "ratio", d["ratio_to_algorithmic"], {kk: v["hbm_bytes_per_launch"] for kk, v in k.items()})


def sq(tag, nframes):
    p = G / f"sq_{tag}.json"
    if not p.exists():
        print("missing", p)
        return None
    raw = json.load(open(p))
    out = {"bench_args": raw.get("bench_args"), "note": "per launch; SQ cycle counters in units of 4 clocks; rocprofv3 --pmc, four passes (tools/pmc_split.sh)", "kernels": {}}
    for k, c in raw.items():
        if not isinstance(c, dict) or "SQ_WAVE_CYCLES" not in c:
            continue
        wc, gui = c["SQ_WAVE_CYCLES"], c.get("GRBM_GUI_ACTIVE", 0) / 8
        simd = gui * 1024 / 4 if gui else None            # SIMD quad-cycles of the launch (8 XCDs x 32 CUs x 4 SIMDs)
        c["derived"] = {
            "per_frame": {"valu": round(c["SQ_INSTS_VALU"] / nframes), "salu": round(c["SQ_INSTS_SALU"] / nframes),
                          "lds": round(c["SQ_INSTS_LDS"] / nframes), "vmem": round(c["SQ_INSTS_VMEM"] / nframes)},
            "issuing_share_of_wave_cycles": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3), "waiting_share": round(c["SQ_WAIT_ANY"] / wc, 3),
            "issue_stall_share": round(c["SQ_WAIT_INST_ANY"] / wc, 3), "valu_busy_per_wave": round(c["SQ_ACTIVE_INST_VALU"] / wc, 3),
            "waves_per_simd": round(wc / simd, 2) if simd else None, "valu_busy_per_simd": round(c["SQ_ACTIVE_INST_VALU"] / simd, 3) if simd else None,
            "lds_busy_per_simd": round(c["SQ_ACTIVE_INST_LDS"] / simd, 3) if simd else None,
            "lds_pipe_busy_per_cu": round(c["SQ_LDS_IDX_ACTIVE"] / (gui * 256), 3) if gui else None,
            "lds_conflict_share_of_lds_active": round(c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 3),
            "clock_ghz": round(gui / c["kernel_ms_under_pmc"] / 1e6, 2) if gui else None}
        if "SQ_INSTS_VALU_ADD_F64" in c:
            f = {"add_f64": c["SQ_INSTS_VALU_ADD_F64"], "mul_f64": c["SQ_INSTS_VALU_MUL_F64"], "fma_f64": c["SQ_INSTS_VALU_FMA_F64"], "trans_f64": c["SQ_INSTS_VALU_TRANS_F64"],
                 "cvt": c["SQ_INSTS_VALU_CVT"], "int32": c["SQ_INSTS_VALU_INT32"], "int64": c["SQ_INSTS_VALU_INT64"]}
            f["other"] = c["SQ_INSTS_VALU"] - sum(f.values())
            c["derived"]["classes_per_frame"] = {k: round(v / nframes) for k, v in f.items()}
        if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            c["derived"]["lane_occupancy"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64, 3)
        out["kernels"][k] = c
    return out


for name, wl in WORKLOADS.items():
    s1 = sq(f"{V}_{name}", wl["streams"] * wl["frames_per_step"])
    if not s1:
        continue
    s1["workload"] = wl
    json.dump(s1, open(P / (f"{V}_sq_counters.json" if name == "hl" else f"{V}_sq_counters_{name}.json"), "w"), indent=1)
    for k, c in s1["kernels"].items():
        print(name, k, c["derived"])
(P / "LATEST").write_text(V + "\n")        # bench.py quotes the counter files of this tag (committed_counters)
