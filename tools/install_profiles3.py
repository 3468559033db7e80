#!/usr/bin/env python3
"""Copy the summaries tools/profile_round3.sh TAG left in gpurun_out/ into profiles/ (tracked) and derive the two JSON files
bench.py quotes (profiles/pmc_traffic_latest.json, profiles/sq_counters_latest.json).   usage: tools/install_profiles2.py TAG"""
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


cp(G / f"{V}_bench_kernel_stats.csv")
for f in list(G.glob(f"stage_{V}_psy*.txt")) + list(G.glob(f"bench_{V}_*.json")) + list(G.glob(f"sq_{V}_*.txt")):
    cp(f)
cp(G / f"edi_bench_{V}.txt", f"{V}_edi_kernels.txt")
bench = json.loads(open(G / f"bench_{V}_default.json").read().strip().splitlines()[-1])
wl = {"streams": bench["config"]["streams_per_gpu"], "frames_per_step": bench["config"]["frames_per_step"], "psy": 1, "mode": "s"}
frames = wl["streams"] * wl["frames_per_step"]


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


k1, t1 = traffic("hl")
if k1:
    d = {"version": f"{V} ({bench['value'] / 1e6:.2f} M frames/s)", "workload": wl, "kernels": k1, "hbm_bytes_per_launch": t1,
         "algorithmic_bytes_per_launch": 4992 * frames, "ratio_to_algorithmic": round(t1 / (4992 * frames), 3),
         "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
         "note": "models 1/3: tl_frame_kernel (psy model, then encoder, per (stream, frame) unit; the model's record stays on chip) + "
                 "tl_finish_kernel; units come off per-XCD lists in stream order, so the PCM is fetched once and a frame's 480 samples of "
                 "history come out of the XCD's L2 (DESIGN.md section 4)",
         "source": f"tools/profile_round3.sh {V} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, kernel trace only); raw CSVs: profiles/{V}_pmc_hl_*"}
    json.dump(d, open(P / f"{V}_pmc_traffic.json", "w"), indent=1)
    json.dump(d, open(P / "pmc_traffic_latest.json", "w"), indent=1)
    print("traffic", t1, "ratio", d["ratio_to_algorithmic"], {k: v["hbm_bytes_per_launch"] for k, v in k1.items()})
k3, t3 = traffic("psy3")
if k3:
    f3 = 16384 * 8
    d3 = {"version": V, "workload": {"streams": 16384, "frames_per_step": 8, "psy": 3, "mode": "s"}, "kernels": k3, "hbm_bytes_per_launch": t3,
          "algorithmic_bytes_per_launch": 4992 * f3, "ratio_to_algorithmic": round(t3 / (4992 * f3), 3),
          "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
          "note": "BASELINE configs[2] = one GPU's share of configs[3]: tl_frame_kernel<3> + tl_finish_kernel", "source": f"tools/profile_round3.sh {V}; raw CSVs: profiles/{V}_pmc_psy3_*"}
    json.dump(d3, open(P / f"{V}_pmc_traffic_psy3.json", "w"), indent=1)
    print("psy3 traffic", t3, "ratio", d3["ratio_to_algorithmic"])
k2, t2 = traffic("psy2")
if k2:
    d2 = {"version": V, "workload": dict(wl, psy=2), "kernels": k2, "hbm_bytes_per_launch": t2, "algorithmic_bytes_per_launch": 4992 * frames,
          "ratio_to_algorithmic": round(t2 / (4992 * frames), 3),
          "note": "tl_psy2_kernel (models 2 and 4): the r/phi prediction state of psycho_2.c:300-306 (32.8 KB per stream) is read and "
                  "written per 576-sample pass; then tl_main_kernel<2> + tl_finish_kernel as for the other models", "source": f"tools/profile_round3.sh {V}; raw CSVs: profiles/{V}_pmc_psy2_*"}
    json.dump(d2, open(P / f"{V}_pmc_traffic_psy2.json", "w"), indent=1)
    print("psy2 traffic", t2, "ratio", d2["ratio_to_algorithmic"])


def sq(tag, nframes):
    p = G / f"sq_{tag}.json"
    if not p.exists():
        print("missing", p)
        return None
    raw = json.load(open(p))
    out = {"bench_args": raw.get("bench_args"), "note": "per launch; SQ cycle counters in units of 4 clocks; rocprofv3 --pmc, three passes (tools/pmc_split.sh)", "kernels": {}}
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
            "lds_conflict_share_of_lds_active": round(c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 3),
            "clock_ghz": round(gui / c["kernel_ms_under_pmc"] / 1e6, 2) if gui else None}
        out["kernels"][k] = c
    return out


s1 = sq(f"{V}_hl", frames)
if s1:
    s1["workload"] = wl
    json.dump(s1, open(P / f"{V}_sq_counters.json", "w"), indent=1)
    json.dump(s1, open(P / "sq_counters_latest.json", "w"), indent=1)
    for k, c in s1["kernels"].items():
        print(k, c["derived"])
for name, nf, wl2 in (("psy3", 16384 * 8, {"streams": 16384, "frames_per_step": 8, "psy": 3, "mode": "s"}), ("psy2", frames, dict(wl, psy=2))):
    s2 = sq(f"{V}_{name}", nf)
    if s2:
        s2["workload"] = wl2
        json.dump(s2, open(P / f"{V}_sq_counters_{name}.json", "w"), indent=1)
        for k, c in s2["kernels"].items():
            print(name, k, c["derived"])
json.dump(bench, open(P / f"{V}_bench_default.json", "w"))
cp(G / f"legacy_latency_{V}.txt")
(P / "LATEST").write_text(V + "\n")        # bench.py quotes the counter files of this tag (committed_counters)
