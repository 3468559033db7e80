#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched DAB MP2 encode path on MI355X.

Metric (BASELINE.json): real-time stereo DAB MP2 streams sustained, as frames/s, @128 kbps / 48 kHz.

Workload
  N = 1 : BASELINE.json configs[1] -- 4096 streams x 48 kHz stereo x 128 kbps, psy model 1, mode 's', full encode
          (filterbank + psy + allocation + quantise + pack + CRC/ScF-CRC), PCM resident in HBM.  After the timed region
          (never inside it) the same process also times mode 'j', configs[2] (16384 streams, psy 3) and the
          PCIe-inclusive rate (pinned host buffers through tlb_encode_host) and reports them under "also".
  N > 1 : BASELINE.json configs[3] -- 16384 streams PER GPU, psy model 3 (131072 streams on 8 GPUs), weak scaling.
  --config 4 (any N): BASELINE.json configs[4] -- per GPU 16384 streams, even 32 kHz mono 64 kbps / odd 48 kHz stereo 192 kbps in ONE
          batch, psy model 4 as `value`, psy model 2 (its sibling the reference's setter accepts) as `also.configs4_psy2`; both sharded
          over the N ranks with the same barriers, per-GPU and aggregate rates, an oracle check on every rank.

A "step" = one launch of the hot path: every stream encodes --frames-per-step consecutive frames (default: 131072 (stream,
frame) units per launch, i.e. 32 frames at 4096 streams, 8 at 16384; the kernels' ramp-up and tail are paid once per launch).  One process per GPU; streams shard with no data-path collective; RCCL carries only the
barriers, the max-over-ranks of the elapsed time and one all_gather of (frames, seconds) per rank.

Launching: `python bench.py --gpus N` starts the N rank processes ITSELF (fresh children, created before this
process imports torch or touches a GPU); under `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE set)
it is one of the ranks.  --gpus must equal the world size.  `--dry-run` replaces the GPU by the TEST-ONLY lane-loop
emulation (tests/emu) over gloo so that the multi-rank plumbing can be exercised on a CPU box; its numbers are not
measurements and the line says so.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

FS, KBPS = 48000, 128
ALGO_BYTES_PER_FRAME = 2 * 1152 * 2 + 144000 * KBPS // FS      # PCM in + bitstream out = 4992
HBM_PEAK_GBS = 8000.0                                           # MI355X_MICROARCH.md: 8 TB/s HBM3E
# BASELINE.json configs[k] -> (streams per GPU, psy model)
CONFIGS = {1: (4096, 1), 2: (16384, 3), 3: (16384, 3), 4: (16384, 4)}
FP64_PEAK_TFLOPS = 78.6          # MI355X vector fp64 (MI355X_MICROARCH.md); MFMA does not apply: no dense contraction, exact summation order
# SURVEY 8(d) "algorithmic flops per frame", per CHANNEL-frame and stage, fp64: filterbank 72 x (512 + 512 MAC) / 2 ch = 73.7 k; encoder rest
# (scalefactors, quantiser) 5 k; psy 1 / 3: one 1024-point FHT 25 k + model 50-75 k (taken as 62.5 k) + 544 log10 -> 0.35 MFLOP per STEREO
# frame with the filterbank, the figure SURVEY quotes; psy 0: no model; psy 2 / 4 (psycho_2.c:52-254, two 576-sample passes per frame, each a
# 1024-point FHT 25 k, polar form + prediction + unpredictability 513 x ~40 = 20 k, partition sums + 64 x 64 spreading x 2 = 18 k, thresholds 7 k)
# = 140 k per channel-frame.  These are ALGORITHMIC counts (what the reference's arithmetic needs), not instructions issued.
FLOPS_PER_CHANNEL_FRAME = {0: 78.7e3, 1: 175e3, 3: 175e3, 2: 78.7e3 + 140e3, 4: 78.7e3 + 140e3}
# Issue cost of one wave64 vector instruction by class, SIMD cycles.  Measured per instruction in single-class loops at 3 waves per SIMD
# (profiles/instr_rates_r04.txt: add / mul_f64 4.41, fma_f64 5.20, rcp / sqrt_f64 16.3, cvt 4.3, plain 32-bit integer ops 2.84 and shifts / max /
# mul 4.3 -> 3.75 for the kernel's mix, 64-bit integer 4.4, the rest -- moves 2.9, DPP / selects 4.4, compares and lane reads 5.4-5.5 -> 4.4) and
# scaled by 4.00 / 4.41: the loops' own overhead shows in their fp64 add, which the hardware issues in exactly 4 cycles (16 lanes per cycle).
_K = 4.00 / 4.41
CLASS_COST = {"add_f64": 4.41 * _K, "mul_f64": 4.41 * _K, "fma_f64": 5.20 * _K, "trans_f64": 16.3 * _K, "cvt": 4.3 * _K, "int32": 3.75 * _K, "int64": 4.4 * _K,
              "other": 4.4 * _K}
FLAT_COST = 3.9                  # no class split committed for the workload: every instruction at the mean cost of the configs[1] mix


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (default: the BASELINE config of this N)")
    ap.add_argument("--frames-per-step", type=int, default=None, help="frames per stream per launch (default: 131072 (stream, frame) units per launch: 32 at 4096 streams, 8 at 16384)")
    ap.add_argument("--config", type=int, default=None, choices=(1, 2, 3, 4), help="BASELINE.json configs[k] (default: 1 at --gpus 1, 3 above)")
    ap.add_argument("--psy", type=int, default=None)
    ap.add_argument("--mode", default="s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements after the headline")
    ap.add_argument("--dry-run", action="store_true", help="CPU emulation over gloo: plumbing check only, not a measurement")
    ap.add_argument("--force-group", action="store_true",
                    help="build the process group even at world size 1, so that init_process_group / barrier / all_reduce / all_gather run through the "
                         "chosen backend (RCCL with --backend nccl) on a one-GPU box")
    ap.add_argument("--in-process", type=int, default=0, metavar="G",
                    help="drive the product's node level (tlb_node_*, csrc/tlb_node.cpp) instead of one rank per GPU: G shards in THIS process, shard g on "
                         "device g mod device count, one host thread each; reported under `node` beside the usual line of shard 0's workload")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend of the ranks; gloo (with ranks sharing GPUs: device = LOCAL_RANK mod device count) is for smoke-testing "
                         "the multi-rank path on a box with fewer GPUs than ranks -- its value is not a scaling measurement")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent has not imported torch and
    never touches a GPU), relay rank 0's JSON line, fail if any rank fails."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        return 1
    return 0


def stream_configs(M, S, psy, mode, mixed):
    """the S stream configurations of one GPU: BASELINE configs[4]'s interleaved mix, or S times the metric's configuration"""
    if mixed:
        return [M.StreamConfig(samplerate=32000, mode="m", bitrate=64, psy_model=psy) if s % 2 == 0 else
                M.StreamConfig(samplerate=48000, mode="s", bitrate=192, psy_model=psy) for s in range(S)]
    return [M.StreamConfig(samplerate=FS, mode=mode, bitrate=KBPS, psy_model=psy)] * S


def algo_bytes(c):
    """PCM in + bitstream out per frame of configuration c"""
    return (1 if c.mode == "m" else 2) * 1152 * 2 + 144000 * c.bitrate // c.samplerate


def workload_label(streams, psy, mode, frames_per_step, world, mixed=False):
    if mixed:
        total = f", {streams * world} streams in total" if world > 1 else ""
        return (f"{streams} streams/GPU, even 32 kHz mono 64 kbps / odd 48 kHz stereo 192 kbps interleaved in one batch, psy {psy}, full encode "
                f"(BASELINE configs[4]{total}), {frames_per_step} frames/stream/step"), 4
    k = None
    if mode == "s":                                  # the BASELINE configurations are plain two-channel stereo: mono pairs, joint and dual channel are variants
        if (streams, psy) == CONFIGS[1] and world == 1:
            k = 1
        elif (streams, psy) == CONFIGS[2]:
            k = 2 if world == 1 else 3
    near = {"m": "a mono variant of it", "j": "its joint-stereo variant", "d": "its dual-channel variant"}.get(mode)
    is_shape = (streams, psy) in (CONFIGS[1], CONFIGS[2])
    tag = f"BASELINE configs[{k}]" if k is not None else ("not a BASELINE config" + (f": {near}" if near and is_shape else ""))
    total = f", {streams * world} streams in total" if world > 1 else ""
    chans = "mono, two streams per wave" if mode == "m" else f"stereo (mode '{mode}')"
    return (f"{streams} streams/GPU x 48 kHz {chans} x 128 kbps, psy {psy}, full encode ({tag}{total}), "
            f"{frames_per_step} frames/stream/step"), k


def cpu_baseline(psy, mode, seconds_target=12.0):
    """The REAL reference (oracle/_ref/libtoolame_ref.so, built from the reference's own sources) when
    it travelled with the repo, else the oracle port; one core, bounded sample of the same workload."""
    ref_so = ROOT / "oracle" / "_ref" / "libtoolame_ref.so"
    ora_so = ROOT / "oracle" / "libmp2oracle.so"
    if not ora_so.exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "libmp2oracle.so"], check=False)
    nframes = int(seconds_target * 4500)
    child = r"""
import ctypes as C, sys, time, numpy as np
ref, ora, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
O = C.CDLL(ora); O.mp2o_gen_pcm.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_void_p]
pcm = np.zeros((64, 2, 1152), dtype=np.int16)
for f in range(64): O.mp2o_gen_pcm(0, 0, f, pcm[f].ctypes.data)
out = (C.c_ubyte * 4096)()
if ref:
    L = C.CDLL(ref)
    L.toolame_set_samplerate.argtypes = [C.c_long]; L.toolame_set_channel_mode.argtypes = [C.c_char]
    L.toolame_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.toolame_init(); L.toolame_set_samplerate(%d); L.toolame_set_psy_model(%d); L.toolame_set_channel_mode(b'%s')
    L.toolame_set_bitrate(%d); L.toolame_set_pad(0)
    ptrs = [pcm[f].ctypes.data for f in range(64)]
    t = time.perf_counter()
    for i in range(n): L.toolame_encode_frame(ptrs[i & 63], None, 0, out, 4096)
    dt = time.perf_counter() - t
else:
    O.mp2o_create.restype = C.c_void_p; O.mp2o_create.argtypes = [C.c_long, C.c_char, C.c_int, C.c_int, C.c_int]
    O.mp2o_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    h = O.mp2o_create(%d, b'%s', %d, %d, 0)
    ptrs = [pcm[f].ctypes.data for f in range(64)]
    t = time.perf_counter()
    for i in range(n): O.mp2o_encode_frame(h, ptrs[i & 63], None, 0, out, 4096)
    dt = time.perf_counter() - t
print(n / dt)
""" % (FS, psy, mode, KBPS, FS, mode, KBPS, psy)
    kind = "reference" if ref_so.exists() else "port"
    cmd = [sys.executable, "-c", child, str(ref_so) if ref_so.exists() else "", str(ora_so)]
    try:
        r = subprocess.run(cmd + [str(nframes)], capture_output=True, text=True, timeout=120)
        fps = float(r.stdout.strip().splitlines()[-1])
    except Exception as ex:  # noqa: BLE001
        return {"value": None, "unit": "frames/s", "cores": 1, "kind": kind, "sample": f"failed: {ex}"}
    # also the whole host -- one independent encoder process per core (the reference is a process-global
    # singleton), half the sample each, rates summed
    allc = None
    try:
        ncpu = min(len(os.sched_getaffinity(0)), 32)
        procs = [subprocess.Popen(cmd + [str(nframes // 2)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for _ in range(ncpu)]
        rates = [float(p_.communicate(timeout=180)[0].strip().splitlines()[-1]) for p_ in procs]
        model = ""
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
        allc = {"value": round(sum(rates), 1), "cores": ncpu, "cpu": model}
    except Exception as ex:  # noqa: BLE001
        allc = {"value": None, "error": str(ex)}
    return {"value": round(fps, 1), "unit": "frames/s", "cores": 1, "kind": kind, "all_cores": allc,
            "sample": f"{nframes} frames of one stream (seed 0, tones+noise), {FS} Hz mode '{mode}' {KBPS} kbps psy {psy}, "
                      f"{'libtoolame-dab compiled from the reference sources' if kind == 'reference' else 'oracle/mp2_oracle.c'}, gcc -O2, 1 thread"}


def committed_counters(S, F, psy, mode, mixed=False):
    """HBM bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes) and the SQ issue counters of THIS
    workload, from the files the last profiling round committed (profiles/LATEST names the tag; tools/profile_round3.sh): quoted
    only when (streams, frames per step, psy model, mode) match; never measured inside a bench run."""
    traffic, source, valu = None, None, None
    try:
        tag = (ROOT / "profiles" / "LATEST").read_text().split()[0]
    except Exception:  # noqa: BLE001
        return None, None, None
    for f in sorted((ROOT / "profiles").glob(f"{tag}_pmc_traffic*.json")):
        try:
            pm = json.load(open(f))
            wl = pm["workload"]
            if (wl["streams"], wl["frames_per_step"], wl["psy"], wl["mode"], bool(wl.get("mixed"))) == (S, F, psy, mode, mixed):
                traffic = pm["hbm_bytes_per_launch"]
                source = (f"profiles/{f.name} (committed rocprofv3 --pmc measurement of this workload, summed over the launch's kernels: "
                          + ", ".join(f"{k} {v['hbm_bytes_per_launch']}" for k, v in pm["kernels"].items()) + "; not measured in this run)")
                break
        except Exception:  # noqa: BLE001
            continue
    for f in sorted((ROOT / "profiles").glob(f"{tag}_sq_counters*.json")):
        try:
            sq = json.load(open(f))
            wl = sq["workload"]
            if (wl["streams"], wl["frames_per_step"], wl["psy"], wl["mode"], bool(wl.get("mixed"))) == (S, F, psy, mode, mixed):
                valu = {"kernels": {k: {"valu_busy_per_simd": c["derived"]["valu_busy_per_simd"], "waves_per_simd": c["derived"]["waves_per_simd"],
                                        "wave_cycles_waiting": c["derived"]["waiting_share"], "valu_instructions_per_frame": c["derived"]["per_frame"]["valu"]}
                                    for k, c in sq["kernels"].items()}}
                valu["clock_ghz"] = next((c["derived"].get("clock_ghz") for c in sq["kernels"].values() if c["derived"].get("clock_ghz")), None)
                cls = [c["derived"]["classes_per_frame"] for c in sq["kernels"].values() if c["derived"].get("classes_per_frame")]
                valu["classes_per_frame"] = {k: sum(c[k] for c in cls) for k in cls[0]} if cls and len(cls) == len(sq["kernels"]) else None
                valu["lane_occupancy"] = {k: c["derived"].get("lane_occupancy") for k, c in sq["kernels"].items()}
                valu["source"] = f"profiles/{f.name} (committed rocprofv3 --pmc SQ counters of this workload, not this run)"
                break
        except Exception:  # noqa: BLE001
            continue
    return traffic, source, valu


def device_identity(torch, index):
    """what this rank's GPU IS: UUID, PCI address, name -- gathered over the collective into `devices_observed`"""
    d = {"index": int(index)}
    try:
        pr = torch.cuda.get_device_properties(index)
        d["name"] = str(pr.name)
        u = getattr(pr, "uuid", None)
        if u is not None:
            d["uuid"] = str(u)
        dom, bus, dev = (getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is not None:
            d["pci_bus_id"] = "%04x:%02x:%02x.0" % (int(dom or 0), int(bus), int(dev or 0))
        d["compute_units"] = int(pr.multi_processor_count)
        d["hbm_gb"] = round(pr.total_memory / 1e9, 1)
    except Exception as ex:  # noqa: BLE001
        d["error"] = str(ex)
    if "pci_bus_id" not in d:          # a torch build without the PCI fields: ask the HIP runtime itself
        try:
            hip = C.CDLL("libamdhip64.so")
            buf = C.create_string_buffer(32)
            if hip.hipDeviceGetPCIBusId(buf, 32, int(index)) == 0:
                d["pci_bus_id"] = buf.value.decode().lower()
        except Exception:  # noqa: BLE001
            pass
    return d


def observed_devices(shard, dist, ident, cdev, setup_s):
    """-> the per-rank identity records, gathered with ONE all_gather over the run's backend, and how many DISTINCT GPUs they name"""
    me = json.dumps(dict(ident, setup_s=round(setup_s, 1), host=socket.gethostname(), pid=os.getpid()), separators=(",", ":")).encode()
    recs = []
    for r, b in enumerate(shard.gather_bytes(dist, me, width=320, device=cdev)):
        try:
            recs.append(dict(json.loads(b.decode()), rank=r))
        except Exception:  # noqa: BLE001
            recs.append({"rank": r, "error": "unreadable identity record"})
    keys = {(x.get("host"), x.get("uuid") or x.get("pci_bus_id") or ("index", x.get("index"))) for x in recs}
    return recs, len(keys)


class SclkSampler:
    """The GPU's shader clock WHILE the timed region runs: a host thread reads the driver's DPM tables (sysfs pp_dpm_sclk: the line marked
    '*' is the current level, what `rocm-smi --showclocks` prints) every 20 ms.  A host may expose several cards in sysfs while the process
    sees one GPU, and the numbering need not agree: every card is sampled and the one that ran fastest over the region -- the one under
    load; idle cards sit near 150 MHz -- is reported.  result(): {"median_mhz", "min_mhz", "max_mhz", "samples", "source"} or None where
    no table can be read or no card left its idle clocks (a region too short for the table to follow).  (The clock the committed SQ
    counter profile derives -- GRBM_GUI_ACTIVE / kernel time under the profiler -- comes out near 2.17 GHz; rocm-smi and this sampler see
    the kernels run at 2.39-2.41 GHz of the 2.4 GHz peak, 1.1 kW of 1.4 kW: tools/clock_probe.sh, profiles/clock_probe_r05.txt.)
    Round 6: the sampler no longer runs INSIDE the timed region (its thread competed for the GIL with the launching thread and
    queried the SMU while the clock was being measured): GpuRun.timed() repeats a few launches after the region, untimed, and samples there;
    and it reads the card whose PCI address is this rank's HIP device, so that another job's card on a shared host is never reported."""

    def __init__(self, pci_bus_id=None):
        """pci_bus_id ("0000:05:00.0", any case): only THAT card's table is read; None / no match: every card, the fastest one reported"""
        import glob
        self.paths = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.matched = False
        if pci_bus_id:
            want = str(pci_bus_id).lower()
            hit = [p_ for p_ in self.paths if os.path.basename(os.path.realpath(os.path.dirname(p_))).lower() == want]
            if hit:
                self.paths, self.matched = hit[:1], True
        self.vals = {p_: [] for p_ in self.paths}
        self.stop, self.th = False, None

    @staticmethod
    def _read(path):
        try:
            for ln in open(path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:  # noqa: BLE001
            return None
        return None

    def __enter__(self):
        import threading
        if self.paths:
            def loop():
                while not self.stop:
                    for p_ in self.paths:
                        v = self._read(p_)
                        if v:
                            self.vals[p_].append(v)
                    time.sleep(0.005)
            self.th = threading.Thread(target=loop, daemon=True)
            self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.th:
            self.th.join(timeout=1.0)

    def result(self):
        best = None
        for p_, v in self.vals.items():
            if v:
                v = sorted(v)
                r = {"median_mhz": v[len(v) // 2], "min_mhz": v[0], "max_mhz": v[-1], "samples": len(v), "source": p_,
                     "card_chosen_by": "the PCI address of this rank's HIP device" if self.matched else "the fastest card over the probe (no PCI match)",
                     "when": "a separate untimed probe of the same launches right after the timed region (ADVICE r5: nothing but the launches runs inside it)"}
                if best is None or r["median_mhz"] > best["median_mhz"]:
                    best = r
        return best if best and best["median_mhz"] >= 1000.0 else None


class GpuRun:
    """One workload resident on this rank's GPU: two alternating PCM buffers (frames [0,F) and [F,2F) of every stream), the
    batch, the output buffer.  step(i) = one launch."""

    def __init__(self, M, torch, np, gen_pcm, stream_ids, F, mode, psy, local_rank, distinct=None, mixed=False):
        S = len(stream_ids)
        host = np.empty((2 * F, S, 2, 1152), dtype=np.int16)
        nd = S if distinct is None else min(S, distinct)
        for k, sid in enumerate(stream_ids[:nd]):     # stream i uses seed i
            host[:, k] = gen_pcm(sid, 0, 0, 2 * F)
        for k in range(nd, S, nd):                    # secondary workloads only: the signals repeat every `distinct` streams
            host[:, k:k + nd] = host[:, :min(nd, S - k)]
        self.pcm = [torch.from_numpy(host[:F].copy()).cuda(), torch.from_numpy(host[F:].copy()).cuda()]
        self.check_streams = sorted({0, S - 1})
        self.check_pcm = {k: host[:, k].copy() for k in self.check_streams}      # the 2F frames stream k loops over
        self.launches = 0
        self.cfgs = stream_configs(M, S, psy, mode, mixed)
        self.algo_bytes_per_launch = sum(algo_bytes(c) for c in self.cfgs) * F
        self.audio_s_per_launch = sum(1152.0 / c.samplerate for c in self.cfgs) * F
        self.batch = M.Batch(self.cfgs, device=local_rank)
        self.out = torch.zeros((F, S, self.batch.out_stride), dtype=torch.uint8, device="cuda")
        self.stream = torch.cuda.current_stream()
        self.F, self.S, self.torch, self.np = F, S, torch, np
        self.cdev = "cuda"
        self.dev, self.sclk_mhz = local_rank, None
        self.pci = device_identity(torch, local_rank).get("pci_bus_id")

    def step(self, i):
        assert (i & 1) == (self.launches & 1)          # the two PCM buffers alternate without a gap: each stream sees one looped signal
        self.batch.encode_device(self.pcm[i & 1].data_ptr(), self.F, self.out.data_ptr(), stream=self.stream.cuda_stream)
        self.launches += 1

    def timed(self, dist, shard, warmup, steps):
        """-> (max-over-ranks seconds, own seconds, mean kernel ms from HIP events on the launch stream)"""
        torch = self.torch
        for i in range(warmup):
            self.step(i)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

        def run():
            for i in range(steps):
                evs[i][0].record(self.stream)
                self.step(warmup + i)
                evs[i][1].record(self.stream)

        elapsed, own = shard.timed_region_detail(dist, run, device_sync=torch.cuda.synchronize, device=self.cdev)
        kernel_ms = float(self.np.mean([a.elapsed_time(b) for a, b in evs]))
        self.stage_ms = self.batch.last_stage_ms()      # (psy-2 kernel, encode + finish kernels) of the last launch: batches of models 2/4 only
        # the shader clock under this load: an untimed probe of the same launches, AFTER the region (an even number: the buffers keep alternating)
        probe = max(2, min(2 * (steps // 8), int(0.06 / max(own / max(steps, 1), 1e-6)) // 2 * 2))
        with SclkSampler(self.pci) as sclk:
            for i in range(probe):
                self.step(warmup + steps + i)
            torch.cuda.synchronize()
        self.sclk_mhz = sclk.result()
        return elapsed, own, kernel_ms

    def check_flag(self):
        """check() that never raises: a failing rank must still reach the collectives the other ranks wait in (the flag travels with
        the gathered floats, the line reports it per rank and the process exits non-zero afterwards)"""
        try:
            return self.check()
        except AssertionError as ex:
            return {"checked": True, "failed": str(ex)}

    def check(self, max_oracle_frames=12000):
        """After the timed region: what the LAST timed launch wrote is what the reference writes.  The oracle (TEST-ONLY CPU
        restatement, pinned against the compiled reference) encodes the same looped PCM of the first and the last stream from the
        first warm-up launch on and its bytes for the last launch's frames are compared with the output buffer (slot f of a launch
        holds the frame that became final during input frame f: global frame index (launches - 1) * F + f - 1)."""
        import oraclelib as O
        chk = self.out[min(1, self.F - 1), :4].cpu().numpy()
        assert all(bytes(chk[s, :2]) == b"\xff\xfc" for s in range(min(4, self.S))), "output is not an MPEG audio frame"
        total = self.launches * self.F
        if total > max_oracle_frames:
            return {"checked": False, "why": f"{total} frames per stream since the first launch: beyond the oracle budget of this check"}
        out = self.out.cpu().numpy()
        for k in self.check_streams:
            c, fb = self.cfgs[k], self.batch.frame_bytes[k]
            loop = self.check_pcm[k]
            pcm = self.np.concatenate([loop] * (self.launches // 2 + 1))[:total]
            ref = O.oracle_stream(pcm, samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=c.psy_model)[0]
            first = (self.launches - 1) * self.F - 1                    # global index of the frame in slot 0
            for f in range(self.F):
                g = first + f
                if g < 0:
                    continue
                assert bytes(out[f, k, :fb]) == ref[g * fb:(g + 1) * fb], f"bench output differs from the oracle: stream {k}, slot {f}"
        return {"checked": True, "streams": self.check_streams, "frames_compared_per_stream": self.F, "frames_encoded_by_the_oracle_per_stream": total,
                "what": "the last launch's output buffer (the timed launches and the short clock probe behind them are one uninterrupted sequence of the same "
                        "launch), byte for byte against oracle/mp2_oracle.c on the same looped PCM"}

    def close(self):
        self.batch.close()
        del self.pcm, self.out


def pcie_inclusive(M, np, gen_pcm, F, psy, mode, local_rank, streams, reps=3):
    """tlb_encode_host on PINNED host buffers: PCM over PCIe in, frames over PCIe out, per call.  A labelled secondary
    figure; never `value`."""
    L = M.load_library()
    b = M.Batch([M.StreamConfig(samplerate=FS, mode=mode, bitrate=KBPS, psy_model=psy)] * streams, device=local_rank)
    n_in, n_out = F * streams * 2304 * 2, F * streams * b.out_stride
    p_in, p_out = L.tlb_host_alloc(n_in), L.tlb_host_alloc(n_out)
    if not p_in or not p_out:
        return {"value": None, "error": "tlb_host_alloc failed"}
    pcm = np.ctypeslib.as_array(C.cast(p_in, C.POINTER(C.c_int16)), shape=(F, streams, 2, 1152))
    base = np.stack([gen_pcm(s, 0, 0, F) for s in range(64)], axis=1)
    pcm[...] = np.tile(base, (1, streams // 64 + 1, 1, 1))[:, :streams]
    L.tlb_encode_host(b.h, p_in, F, None, None, p_out, None)        # first call allocates the device staging buffers
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = L.tlb_encode_host(b.h, p_in, F, None, None, p_out, None)
        assert rc == 0
    dt = time.perf_counter() - t0
    L.tlb_host_free(p_in)
    L.tlb_host_free(p_out)
    b.close()
    return {"value": round(reps * F * streams / dt, 1), "unit": "frames/s", "what": "tlb_encode_host, pinned host buffers: "
            f"{n_in / 1e6:.0f} MB PCM in + {n_out / 1e6:.0f} MB frames out over PCIe per call (copy-in, kernels and copy-out of four chunks of frames "
            "overlapped on three streams inside the library; the call returns when everything is back on the host)", "streams": streams,
            "frames_per_call": F, "gbytes_per_s_over_pcie": round(reps * (n_in + n_out) / dt / 1e9, 2)}


def tick_pipeline(M, np, gen_pcm, nstreams, psy, mode, local_rank, ticks=1000, egress="af", ngroups=0):
    """The composed real-time loop body (tlb_tick_*: pinned host PCM -> PCIe -> ingest -> encode -> EDI AF packets -> PCIe -> pinned
    host), one tick per 24 ms for `nstreams` streams, PCIe-inclusive.  Two ways of driving it: tlb_tick_run (a tick start to end per call:
    wall-clock LATENCY per tick) and tlb_tick_submit / tlb_tick_wait with one tick in flight behind the one being waited for (the next
    tick's copy-in runs under this tick's kernels and copy-out: INTERVAL between finished ticks, and each tick's latency submit -> results).
    The input buffers are filled once (in a deployment the capture side writes them); every tick moves and encodes all of them."""
    t = M.Tick([M.StreamConfig(samplerate=FS, mode=mode, bitrate=KBPS, psy_model=psy)] * nstreams, egress=egress, ngroups=ngroups,
               version=b"odr-audioenc_amd bench", device=local_rank)
    nd = min(nstreams, 1024)
    base = np.stack([gen_pcm(s, 0, 0, 1)[0].T.reshape(-1) for s in range(nd)])          # interleaved L R L R
    for _ in range(2):                                                                  # both input sets
        pcm = t.pcm
        for k in range(0, nstreams, nd):
            pcm[k:k + nd] = base[:min(nd, nstreams - k)]
        t.run()
    for _ in range(4):
        t.run()
    lat, dev = np.empty(ticks), np.empty(ticks)
    t0 = time.perf_counter()
    for i in range(ticks):
        a = time.perf_counter()
        t.run()
        lat[i] = time.perf_counter() - a
        dev[i] = t.last_ms()
    wall = time.perf_counter() - t0
    pk = t.packets(0)
    assert len(pk) == 1 and pk[0][:2] == b"AF", "no AF packet came out of the tick"
    n_in, n_out = nstreams * 2304 * 2, nstreams * (len(pk[0]) + 4 + 4)
    # overlapped: submit tick i + 1, then wait for tick i
    done, sub = np.empty(ticks), np.empty(ticks + 1)
    sub[0] = time.perf_counter()
    t.submit()
    for i in range(ticks):
        sub[i + 1] = time.perf_counter()
        t.submit()
        t.wait()
        done[i] = time.perf_counter()
    t.wait()
    pk2 = t.packets(0)
    assert len(pk2) == 1 and pk2[0][:2] == b"AF"
    t.close()
    lat *= 1e3
    iv = np.diff(done) * 1e3
    olat = (done - sub[:ticks]) * 1e3
    p50, p99 = float(np.percentile(lat, 50)), float(np.percentile(lat, 99))
    return {"workload": f"{nstreams} streams x 1 frame per tick (48 kHz stereo {KBPS} kbps, psy {psy}, mode '{mode}'), tlb_tick_run: interleaved PCM in pinned host "
                        f"memory -> PCIe -> gain/peak/de-interleave -> encode -> EDI AF packet per stream -> PCIe -> pinned host memory; {ticks} ticks back to back",
            "ticks": ticks, "p50_ms": round(p50, 3), "p99_ms": round(p99, 3), "max_ms": round(float(lat.max()), 3), "mean_ms": round(float(lat.mean()), 3),
            "device_ms_mean": round(float(dev.mean()), 3), "budget_ms": 24.0, "p99_share_of_budget": round(p99 / 24.0, 4),
            "frames_per_s": round(nstreams * ticks / wall, 1), "mbytes_in_per_tick": round(n_in / 1e6, 1), "mbytes_out_per_tick": round(n_out / 1e6, 1),
            "pcie_gbytes_per_s_in": round(n_in / (float(lat.mean()) * 1e-3) / 1e9, 2),
            "overlapped": {"what": "tlb_tick_submit / tlb_tick_wait, the next tick submitted before this one is waited for: `interval` = time between finished ticks "
                                   "(sustained rate), `latency` = submit -> results of the same tick (includes waiting behind the tick before it on the link)",
                           "interval_p50_ms": round(float(np.percentile(iv, 50)), 3), "interval_p99_ms": round(float(np.percentile(iv, 99)), 3),
                           "latency_p50_ms": round(float(np.percentile(olat, 50)), 3), "latency_p99_ms": round(float(np.percentile(olat, 99)), 3),
                           "pcie_gbytes_per_s_in": round(n_in / (float(iv.mean()) * 1e-3) / 1e9, 2),
                           "link_floor_ms": round(n_in / 57.6e9 * 1e3, 3), "link_floor_basis": "one 604 MB hipMemcpyAsync runs at 57.6 GB/s on this box (profiles/h2d_probe_r04.txt)"},
            "limit": "the host-to-device link: " + f"{n_in / 1e6:.0f} MB of PCM per tick" if p50 > 0.4 * 24 else "none near the budget"}


def node_in_process(M, np, gen_pcm, G, S, F, psy, mode, ndev, warmup, steps, mixed=False):
    """The product's own multi-GPU object (include/toolame_batch.h part 3, csrc/tlb_node.cpp) driven from ONE process: G shards of S streams
    each (weak scaling, like the ranks), shard g on device g mod `ndev`, one host thread per shard, BATCH plane (PCM resident in each shard's
    HBM, two alternating buffers).  A step = tlb_node_encode_device() on every shard; the region ends with tlb_node_sync().  The oracle
    then encodes the first stream of the first shard and the last stream of the last shard."""
    import oraclelib as O
    cfgs = []
    for g in range(G):
        cfgs += stream_configs(M, S, psy, mode, mixed)
    nd = M.Node(cfgs, devices=[g % ndev for g in range(G)], plane="batch")
    describe = nd.describe()
    distinct = min(S, 1024)
    base = np.stack([gen_pcm(k, 0, 0, 2 * F) for k in range(distinct)], axis=1)            # [2F][distinct][2][1152]
    host = np.tile(base, (1, (G * S + distinct - 1) // distinct, 1, 1))[:, :G * S]
    if mixed:
        host = host.copy()
    nd.upload(host[:F], slot=0)
    nd.upload(host[F:], slot=1)
    launches = 0
    for i in range(warmup):
        nd.encode_resident(slot=launches & 1)
        launches += 1
    nd.sync()
    c0 = nd.counters()[1]
    t0 = time.perf_counter()
    for i in range(steps):
        nd.encode_resident(slot=launches & 1)
        launches += 1
    nd.sync()
    dt = time.perf_counter() - t0
    per, tot = nd.counters()
    got = nd.download()
    total = launches * F
    checked = {"checked": False, "why": f"{total} frames per stream: beyond the oracle budget of this check"}
    if total <= 12000:
        for k in (0, G * S - 1):
            c = cfgs[k]
            fb = 144000 * c.bitrate // c.samplerate
            pcm = np.concatenate([host[:, k]] * (launches // 2 + 1))[:total]
            ref = O.oracle_stream(pcm, samplerate=c.samplerate, mode=c.mode, kbps=c.bitrate, psy=c.psy_model)[0]
            first = (launches - 1) * F - 1
            want = ref[max(first, 0) * fb:(first + F) * fb]
            assert got[k][-len(want):] == want, f"node output differs from the oracle: stream {k}"
        checked = {"checked": True, "streams": [0, G * S - 1], "frames_compared_per_stream": F,
                   "what": "the last launch's output of the first and the last stream of the node against oracle/mp2_oracle.c"}
    nd.close()
    frames = G * S * F * steps
    return {"what": f"tlb_node_* (csrc/tlb_node.cpp): {G} shards x {S} streams in ONE process, one host thread per shard, devices {[g % ndev for g in range(G)]}, "
                    f"psy {psy}, {F} frames/stream/step, PCM resident per shard; no collective anywhere",
            "shards": G, "devices": [g % ndev for g in range(G)], "describe": describe, "value": round(frames / dt, 1), "unit": "frames/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 4), "frames_counted_by_the_node": tot["frames"] - c0["frames"],
            "per_shard": [{"shard": p_["shard"], "device": p_["device"], "first": p_["first"], "nstreams": p_["nstreams"], "frames": p_["frames"],
                           "busy_ms": round(p_["busy_ns"] / 1e6, 3)} for p_ in per],
            "output_check": checked}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args, argv)               # before torch / any GPU call in this process
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}; they must agree")

    import numpy as np
    import torch

    import odr_audioenc_amd as M
    import odr_audioenc_amd.shard as shard
    from pcmgen import gen_pcm

    world = args.gpus
    cfg_k_arg = args.config if args.config is not None else (1 if world == 1 else 3)
    mixed = cfg_k_arg == 4
    cfg_streams, cfg_psy = CONFIGS[cfg_k_arg]
    S = args.streams if args.streams is not None else cfg_streams
    psy = args.psy if args.psy is not None else cfg_psy
    F = args.frames_per_step if args.frames_per_step is not None else max(1, 131072 // S)
    t0 = time.time()

    if args.dry_run:
        return dry_run(args, shard, np, gen_pcm, world, S if args.streams is not None else 2, psy, min(F, 2))

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback; --dry-run only checks the plumbing)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and int(os.environ.get("LOCAL_RANK", "0")) >= ndev:
        raise SystemExit(f"bench.py: rank needs GPU {os.environ.get('LOCAL_RANK')} but the box has {ndev} (RCCL wants one GPU per rank)")
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)
    torch.cuda.set_device(dev_index)
    if int(os.environ.get("RANK", "0")) != 0:
        # rank 0 alone owns stdout (ONE JSON line): librccl announces itself there with printf ("Librccl path : ..."), buffered until exit
        sys.stdout.flush()
        os.dup2(2, 1)
    rank, local_rank, world_env, dist = shard.init_from_env(args.backend, device_index=dev_index, force=args.force_group)      # "nccl" is RCCL on ROCm
    C.CDLL(None).fflush(None)                    # ... and rank 0's copy of that line goes out now, not after the JSON line
    local_rank = dev_index
    cdev = "cuda" if args.backend == "nccl" else "cpu"          # where the collectives' few scalars live
    assert world_env == world
    observed_world = dist.get_world_size() if dist is not None else 1
    # which GPU every rank REALLY has, gathered by the run's own collective (one all_gather): N ranks on N distinct devices is in the line, not inferred
    devices_seen, n_distinct = observed_devices(shard, dist, device_identity(torch, dev_index), cdev, time.time() - t0)
    num_simds = 4 * torch.cuda.get_device_properties(dev_index).multi_processor_count

    def sharded_run(psy_k):
        """one workload on every rank: timed region between barriers, the oracle check on EVERY rank (its verdict travels as a flag
        with the gathered floats: a rank whose check fails still takes part in the collectives), per-rank (frames, seconds)"""
        r = GpuRun(M, torch, np, gen_pcm, shard.weak_stream_ids(rank, S), F, args.mode, psy_k, local_rank, mixed=mixed)
        r.cdev = cdev
        el, own_s, k_ms = r.timed(dist, shard, args.warmup, args.steps)
        l_ms = r.batch.last_kernel_ms()
        chk = r.check_flag()
        pr = shard.gather_floats(dist, [S * F * args.steps, own_s, 0.0 if chk.get("failed") else 1.0], device=cdev)
        out = dict(elapsed=el, kernel_ms=k_ms, last_ms=l_ms, checked=chk, per_rank=pr, stage_ms=r.stage_ms,
                   algo=r.algo_bytes_per_launch, audio_s=r.audio_s_per_launch, sclk=r.sclk_mhz)
        r.close()
        return out

    head = sharded_run(psy)
    sib = sharded_run(2) if mixed and psy == 4 else None         # configs[4]: psy 2 next to psy 4, on every rank too
    elapsed, kernel_ms, last_ms, checked, per_rank, run_stage_ms = (head[k] for k in ("elapsed", "kernel_ms", "last_ms", "checked", "per_rank", "stage_ms"))
    if dist is not None:          # nothing below is collective: the ranks part here, rank 0 goes on to its host-side legs alone
        dist.destroy_process_group()
        dist = None
    had_group = world > 1 or args.force_group
    all_ok = all(p[2] == 1.0 for p in per_rank) and (sib is None or all(p[2] == 1.0 for p in sib["per_rank"]))

    def roofline(run_d, S_, F_, psy_k, mode_k):
        """The SURVEY 8(d) fractions of the launch's dominant kernel from THIS run's event times: `frac` = algorithmic fp64 flops per second
        over the vector-fp64 peak (the resource that binds this path), `hbm_frac` = algorithmic bytes per second over 8 TB/s (what
        BASELINE.json asks for).  Beside them, clearly apart, how busy the kernel keeps the vector issue ports with its OWN instruction
        stream: the counter value of the committed profile of this workload and the same figure recomputed for this run from the
        committed instruction classes."""
        k_ms = run_d["kernel_ms"]
        traffic, traffic_source, valu = committed_counters(S_, F_, psy_k, mode_k, mixed)
        hbm_ach = run_d["algo"] / (k_ms * 1e-3) / 1e9
        nch_ = 1 if mode_k == "m" else 2
        fps_k = S_ * F_ / (k_ms * 1e-3)
        flops_frame = FLOPS_PER_CHANNEL_FRAME[psy_k] * (1.5 if mixed else nch_)          # configs[4]: mono and stereo streams alternate
        ach = flops_frame * fps_k / 1e12
        rf = {"bound": "valu_fp64", "achieved": round(ach, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP64_PEAK_TFLOPS, 4),
              "flops_per_frame": flops_frame,
              "frac_basis": f"{flops_frame:.0f} algorithmic fp64 flops per frame (SURVEY 8d; FLOPS_PER_CHANNEL_FRAME in bench.py) x {fps_k:.0f} frames/s (this run, "
                            f"HIP-event kernel time) / {FP64_PEAK_TFLOPS} TFLOP/s vector fp64.  The reference's arithmetic allows no FMA contraction (-ffp-contract=off), "
                            "so a kernel of pure fp64 adds and multiplies tops out at half that peak",
              "hbm": {"achieved": round(hbm_ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_ach / HBM_PEAK_GBS, 6),
                      "algorithmic_bytes_per_launch": run_d["algo"], "traffic": traffic, "traffic_source": traffic_source},
              "hbm_frac": round(hbm_ach / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source,
              "kernel_ms": round(k_ms, 4), "last_kernel_ms_hip_events": round(run_d["last_ms"], 4)}
        util = {"what": "share of the SIMDs' issue cycles the launch's vector instructions occupy -- a utilisation of the kernel's own instruction stream, "
                        "NOT a roofline fraction"}
        if valu:
            live = run_d.get("sclk")
            clock = round(live["median_mhz"] / 1e3, 3) if live else (valu.get("clock_ghz") or 2.2)
            clock_note = ("the shader clock sampled from the driver's DPM table while this run's timed region ran (SclkSampler)" if live else
                          "the clock rocprofv3 measured in the committed profile run, not this run's")
            per_frame = sum(v["valu_instructions_per_frame"] for v in valu["kernels"].values())
            classes = valu.get("classes_per_frame")
            if classes:
                cyc = sum(CLASS_COST[c] * n for c, n in classes.items())
                basis = "committed class counts x CLASS_COST (bench.py: profiles/instr_rates_r04.txt at 3 waves per SIMD, scaled so that an fp64 add costs its 4 cycles)"
            else:
                cyc = per_frame * FLAT_COST
                basis = f"no class split committed for this workload: {per_frame} instructions per frame x {FLAT_COST} cycles"
            util.update({"counter_valu_busy_per_simd": {k_: v["valu_busy_per_simd"] for k_, v in valu["kernels"].items()},
                         "counter_lane_occupancy": valu.get("lane_occupancy"),
                         "this_run_from_instruction_classes": round(cyc * fps_k / (num_simds * clock * 1e9), 4),
                         "valu_instructions_per_frame": per_frame, "classes_per_frame": classes, "issue_cycles_per_frame": round(cyc),
                         "basis": f"{basis} x {fps_k:.0f} frames/s / ({num_simds} SIMDs x {clock} GHz -- {clock_note})",
                         "shader_clock": live, "source": valu.get("source")})
        else:
            util.update({"counter_valu_busy_per_simd": None, "this_run_from_instruction_classes": None, "basis": "no committed SQ counters for this workload (profiles/LATEST)"})
        rf["valu_issue_utilisation"] = util
        rf["binding_resource"] = ("the vector ALU: 5 KB of compulsory HBM traffic per 0.35 MFLOP frame (SURVEY F9) leaves HBM at ~2 % of its peak.  `frac` is the "
                                  "algorithmic fp64 fraction; the gap between it and `valu_issue_utilisation` is instructions that are not algorithmic fp64 arithmetic "
                                  "(selects, compares, moves, integer and address work, division and logarithm expansions: DESIGN.md section 4)")
        return rf

    def leg_roofline(S_, F_, psy_k, mode_k, k_ms, algo_b, mixed_k=False):
        """the headline's two SURVEY 8(d) fractions for a secondary leg: fp64 `frac` and `hbm_frac` from THIS run's kernel time, the
        committed HBM traffic of the same workload where a counter file exists (every leg, not only the headline)"""
        tr, trs, _ = committed_counters(S_, F_, psy_k, mode_k, mixed_k)
        fps_k = S_ * F_ / (k_ms * 1e-3)
        ff = FLOPS_PER_CHANNEL_FRAME[psy_k] * (1.5 if mixed_k else (1 if mode_k == "m" else 2))
        return {"bound": "valu_fp64", "frac": round(ff * fps_k / 1e12 / FP64_PEAK_TFLOPS, 4), "achieved": round(ff * fps_k / 1e12, 3), "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s", "flops_per_frame": ff, "hbm_frac": round(algo_b / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "algorithmic_bytes_per_launch": algo_b, "traffic": tr, "traffic_source": trs,
                "traffic_over_algorithmic": round(tr / algo_b, 2) if tr else None, "kernel_ms": round(k_ms, 4)}

    res = None
    if rank == 0:
        frames = sum(int(p[0]) for p in per_rank)
        value = frames / elapsed
        pk = "tl_psy2_kernel" if psy in (2, 4) else None
        kname = (f"{pk} + tl_main_kernel<2> + tl_finish_kernel (the three kernels of one launch of the path)" if pk
                 else "tl_main_kernel<0> + tl_finish_kernel (model 0 has no psy kernel)" if psy == 0
                 else f"tl_frame_kernel<{psy}> (psy model, then encoder, per (stream, frame) unit) + tl_finish_kernel")
        kernels = None
        if pk and run_stage_ms:
            kernels = {pk: round(run_stage_ms[0], 4), "tl_main_kernel + tl_finish_kernel": round(run_stage_ms[1], 4),
                       "source": "hipEvents on the launch stream inside the library (tlb_last_stage_ms), last timed launch"}
        label, cfg_k = workload_label(S, psy, args.mode, F, world, mixed)
        rf = roofline(head, S, F, psy, args.mode)
        rf.update({"kernel": kname, "kernels_ms": kernels})
        checked["per_rank_ok"] = [p[2] == 1.0 for p in per_rank]
        res = {
            "metric": "real-time stereo DAB MP2 streams sustained (frames/s) @128 kbps/48 kHz" if not mixed else
                      "real-time DAB MP2 streams sustained (frames/s), BASELINE configs[4]: 32 kHz mono 64 kbps + 48 kHz stereo 192 kbps interleaved",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "baseline_config": cfg_k, "streams_per_gpu": S, "frames_per_step": F,
                       "frames_per_stream_timed": F * args.steps, "frames_per_stream_warmup": F * args.warmup,
                       "parallelism": f"streams sharded over {world} GPU(s), no data-path collective"},
            "realtime_streams": round(head["audio_s"] * world * args.steps / elapsed, 1),
            "devices_observed": devices_seen, "distinct_devices_observed": n_distinct,
            "world_size_observed": observed_world, "collective_backend": ("rccl (torch.distributed nccl)" if args.backend == "nccl" else "gloo (smoke test: ranks may share GPUs)") if had_group else None,
            "per_gpu_frames_per_s": [round(p[0] / p[1], 1) for p in per_rank],
            "roofline": rf,
            "lds_bytes_per_stream": M.lds_bytes_per_stream(),
            "output_check": checked,
        }
        if sib is not None:
            sib["checked"]["per_rank_ok"] = [p[2] == 1.0 for p in sib["per_rank"]]
            res["also"] = {"configs4_psy2": {
                "workload": workload_label(S, 2, args.mode, F, world, True)[0], "value": round(sum(int(p[0]) for p in sib["per_rank"]) / sib["elapsed"], 1),
                "unit": "frames/s", "per_gpu_frames_per_s": [round(p[0] / p[1], 1) for p in sib["per_rank"]], "ms_per_step": round(sib["elapsed"] / args.steps * 1e3, 4),
                "realtime_streams": round(sib["audio_s"] * world * args.steps / sib["elapsed"], 1), "roofline": roofline(sib, S, F, 2, args.mode),
                "output_check": sib["checked"]}}
    # ---- secondary measurements, after (outside) the headline's timed region; single GPU only ----
    if world == 1 and not args.no_also and not mixed:
        also = {}
        F2 = max(1, 131072 // CONFIGS[2][0])        # 16384 streams x 8 frames = 604 MB of PCM per buffer
        # encoder_only_psy0: BASELINE configs[1] names "filterbank+quantise kernels only" -- that is the encoder without a psychoacoustic
        # model, which the library offers as model 0 (filterbank, scalefactors, bit allocation, quantiser, packing); `value` is the FULL encode
        for name, (s2, p2, m2, f2) in {"mode_j": (S, psy, "j" if args.mode != "j" else "s", F),
                                       "configs2_psy3_16384": (CONFIGS[2][0], CONFIGS[2][1], "s", F2),
                                       "encoder_only_psy0": (S, 0, args.mode, F),
                                       # the same workload with twice the frames per launch: what the fixed part of a launch (tail of the
                                       # persistent waves, finish pass, launch gaps: ~0.09 ms) costs at F frames
                                       "frames_per_launch_x2": (S, psy, args.mode, 2 * F),
                                       # 128 kbps MONO streams, two per wave (tl_encode_pair)
                                       "mono_pairs": (S, psy, "m", F)}.items():
            try:
                r2 = GpuRun(M, torch, np, gen_pcm, range(s2), f2, m2, p2, local_rank)
                e2, _, k2 = r2.timed(None, shard, max(2, args.warmup // 2), max(5, args.steps // 2))
                chk2 = r2.check()
                ab2 = r2.algo_bytes_per_launch
                r2.close()
                n2 = max(5, args.steps // 2)
                wl2 = workload_label(s2, p2, m2, f2, 1)[0]
                if name == "encoder_only_psy0":
                    wl2 = (f"{s2} streams/GPU x 48 kHz stereo (mode '{m2}') x 128 kbps, psy 0 = the encoder without a psychoacoustic model: filterbank, "
                           f"scalefactors, bit allocation, quantiser, packing (what BASELINE configs[1] calls 'filterbank+quantise kernels only'), {f2} frames/stream/step")
                also[name] = {"workload": wl2, "value": round(s2 * f2 * n2 / e2, 1), "unit": "frames/s",
                              "steps": n2, "kernel_ms": round(k2, 4), "roofline": leg_roofline(s2, f2, p2, m2, k2, ab2), "output_check": chk2}
            except Exception as ex:  # noqa: BLE001
                also[name] = {"value": None, "error": str(ex)}
        try:     # BASELINE configs[0] on the GPU: ONE stream -- its frames are independent units for the kernels, so one stream fills the chip
            n1 = 16384
            r1 = GpuRun(M, torch, np, gen_pcm, [0], n1 // 2, args.mode, psy, local_rank)
            e1, _, k1 = r1.timed(None, shard, 2, 6)
            chk1 = r1.check()
            ab1 = r1.algo_bytes_per_launch
            r1.close()
            also["one_stream"] = {"roofline": leg_roofline(1, n1 // 2, psy, args.mode, k1, ab1), "workload": f"1 stream x {n1 // 2} frames per launch (48 kHz stereo 128 kbps, psy {psy}, mode '{args.mode}'; BASELINE configs[0] "
                                              "is this stream on the CPU reference)", "value": round((n1 // 2) * 6 / e1, 1), "unit": "frames/s",
                                  "x_realtime": round((n1 // 2) * 6 / e1 / (FS / 1152.0), 1), "kernel_ms": round(k1, 4), "output_check": chk1}
        except Exception as ex:  # noqa: BLE001
            also["one_stream"] = {"value": None, "error": str(ex)}
        try:     # the whole population of BASELINE configs[3] (131072 streams, psy 3) on ONE GPU, one 24-ms frame each per launch
            nt = 131072
            rt = GpuRun(M, torch, np, gen_pcm, list(range(nt)), 1, args.mode, 3, local_rank, distinct=4096)
            et, _, kt = rt.timed(None, shard, 2, 6)
            chkt = rt.check()
            abt = rt.algo_bytes_per_launch
            rt.close()
            also["tick_131072"] = {"roofline": leg_roofline(nt, 1, 3, args.mode, kt, abt), "workload": f"{nt} streams x 1 frame per launch on one GPU (48 kHz stereo 128 kbps, psy 3, mode '{args.mode}'): "
                                               "one real-time tick of everything BASELINE configs[3] spreads over 8 GPUs; 4096 distinct signals, repeated",
                                   "value": round(nt * 6 / et, 1), "unit": "frames/s", "kernel_ms": round(kt, 4),
                                   "share_of_the_24_ms_tick": round(kt / 24.0, 4), "output_check": chkt}
        except Exception as ex:  # noqa: BLE001
            also["tick_131072"] = {"value": None, "error": str(ex)}
        try:
            also["pcie_inclusive"] = pcie_inclusive(M, np, gen_pcm, 8, psy, args.mode, local_rank, S)
        except Exception as ex:  # noqa: BLE001
            also["pcie_inclusive"] = {"value": None, "error": str(ex)}
        for p4 in (4, 2):        # BASELINE configs[4] names psy 4 (an extension of the batched API); psy 2 is its sibling the reference's setter accepts.  (`--config 4` makes it the headline, at any N)
            try:
                s4, f4 = CONFIGS[4][0], max(1, 131072 // CONFIGS[4][0])
                r4 = GpuRun(M, torch, np, gen_pcm, range(s4), f4, "s", p4, local_rank, distinct=256, mixed=True)
                e4, _, k4 = r4.timed(None, shard, 2, 20)
                also[f"configs4_share_psy{p4}"] = {"workload": workload_label(s4, p4, "s", f4, 1, True)[0] + " (one GPU's share)", "value": round(s4 * f4 * 20 / e4, 1), "unit": "frames/s",
                                                   "ms_per_launch": round(e4 / 20 * 1e3, 3), "streams_at_realtime": round(r4.audio_s_per_launch * 20 / e4),
                                                   "kernels_ms": r4.stage_ms, "roofline": leg_roofline(s4, f4, p4, "s", k4, r4.algo_bytes_per_launch, True), "output_check": r4.check()}
                r4.close()
            except Exception as ex:  # noqa: BLE001
                also[f"configs4_share_psy{p4}"] = {"value": None, "error": str(ex)}
        try:     # ONE stream with psy 2: runs of its frames seed themselves (csrc/mp2_wave.h tl_psy2_chain), so one stream fills the chip with this model too
            n1 = 16384
            r1 = GpuRun(M, torch, np, gen_pcm, [0], n1 // 2, args.mode, 2, local_rank)
            e1, _, k1 = r1.timed(None, shard, 2, 6)
            st1 = r1.stage_ms
            chk1 = r1.check()
            ab1 = r1.algo_bytes_per_launch
            r1.close()
            also["one_stream_psy2"] = {"roofline": leg_roofline(1, n1 // 2, 2, args.mode, k1, ab1), "workload": f"1 stream x {n1 // 2} frames per launch (48 kHz stereo 128 kbps, psy 2, mode '{args.mode}')", "value": round((n1 // 2) * 6 / e1, 1),
                                       "unit": "frames/s", "x_realtime": round((n1 // 2) * 6 / e1 / (FS / 1152.0), 1), "kernel_ms": round(k1, 4), "kernels_ms": st1, "output_check": chk1}
        except Exception as ex:  # noqa: BLE001
            also["one_stream_psy2"] = {"value": None, "error": str(ex)}
        try:     # the real-time shape of models 2 / 4: ONE frame per launch, where the 16 KB per channel of r / phi prediction state is read and written per frame
            st_, ft_ = CONFIGS[2][0], 1
            rt2 = GpuRun(M, torch, np, gen_pcm, range(st_), ft_, args.mode, 2, local_rank, distinct=1024)
            et2, _, kt2 = rt2.timed(None, shard, 4, 40)
            tr2, trs2, _ = committed_counters(st_, ft_, 2, args.mode)
            state_b = st_ * 2 * 2 * 2 * 513 * 8 * 2          # streams x channels x (r, phi) x two passes x 513 lines x 8 B, read + written
            also["psy2_tick_shape"] = {"roofline": leg_roofline(st_, ft_, 2, args.mode, kt2, rt2.algo_bytes_per_launch), "workload": f"{st_} streams x 1 frame per launch (48 kHz stereo 128 kbps, psy 2, mode '{args.mode}'): one GPU's share of a real-time tick with the model "
                                                   "that carries prediction state from frame to frame (psycho_2.c:300-306)",
                                       "value": round(st_ * ft_ * 40 / et2, 1), "unit": "frames/s", "kernel_ms": round(kt2, 4), "kernels_ms": rt2.stage_ms,
                                       "share_of_the_24_ms_tick": round(kt2 / 24.0, 4), "algorithmic_bytes_per_launch": rt2.algo_bytes_per_launch,
                                       "prediction_state_bytes_per_launch": state_b, "traffic": tr2, "traffic_source": trs2,
                                       "traffic_over_algorithmic": round(tr2 / rt2.algo_bytes_per_launch, 2) if tr2 else None,
                                       "hbm_time_of_the_state_ms": round(state_b / (HBM_PEAK_GBS * 1e9) * 1e3, 4), "output_check": rt2.check()}
            rt2.close()
        except Exception as ex:  # noqa: BLE001
            also["psy2_tick_shape"] = {"value": None, "error": str(ex)}
        also["tick_pipeline"] = {}
        for nt2 in (16384, 131072):   # one GPU's share of BASELINE configs[3], and all of configs[3] on one GPU
            try:
                also["tick_pipeline"][str(nt2)] = tick_pipeline(M, np, gen_pcm, nt2, 3, args.mode, local_rank)
            except Exception as ex:  # noqa: BLE001
                also["tick_pipeline"][str(nt2)] = {"value": None, "error": str(ex)}
        res["also"] = also
    if rank == 0 and world == 1 and args.in_process > 0:
        try:
            res["node"] = node_in_process(M, np, gen_pcm, args.in_process, S, F, psy, args.mode, ndev, args.warmup, args.steps, mixed)
        except AssertionError:
            raise
        except Exception as ex:  # noqa: BLE001
            res["node"] = {"value": None, "error": str(ex)}
    if rank == 0:
        if had_group:
            res["collectives_executed"] = ["init_process_group(device_id)", "barrier", "all_reduce(MAX)", "all_gather"]
        res["setup_s"] = round(time.time() - t0, 1)
        if not args.no_cpu_baseline:         # rank 0's host cores, after every rank's timed region (N > 1: the configs[3] model, psy 3)
            res["cpu_baseline"] = cpu_baseline(psy, args.mode)
        else:
            res["cpu_baseline"] = None
        C.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if not all_ok:
        print("bench.py: the oracle check failed on at least one rank (output_check.per_rank_ok)", file=sys.stderr)
        return 3
    return 0


def dry_run(args, shard, np, gen_pcm, world, S, psy, F):
    """The same rank plumbing (env, process group, barriers, max-over-ranks, gather, rank-0 line) with the TEST-ONLY
    emulation of the kernel standing in for the GPU, over gloo.  Not a measurement."""
    import emulib as E
    rank, local_rank, world_env, dist = shard.init_from_env("gloo", force=args.force_group)
    assert world_env == world
    t0 = time.time()
    devices_seen, n_distinct = observed_devices(shard, dist, {"index": rank, "name": "CPU emulation of the kernel (no GPU)"}, "cpu", 0.0)
    ids = list(shard.weak_stream_ids(rank, S))
    pcm = np.stack([gen_pcm(i, 0, 0, 2 * F) for i in ids], axis=1)
    b = E.EmuBatch([dict(mode=args.mode, psy=psy)] * S)
    steps = min(args.steps, 2)

    def run():
        for i in range(steps):
            b.encode(pcm[(i & 1) * F:(i & 1) * F + F])

    elapsed, own = shard.timed_region_detail(dist, run)
    per_rank = shard.gather_floats(dist, [S * F * steps, own])
    if rank == 0:
        frames = sum(int(p[0]) for p in per_rank)
        label, cfg_k = workload_label(S, psy, args.mode, F, world)
        print(json.dumps({
            "metric": "DRY RUN (CPU emulation of the kernel over gloo; plumbing check, NOT a measurement)", "dry_run": True,
            "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": 0,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": {"workload": label, "streams_per_gpu": S, "frames_per_step": F},
            "world_size_observed": dist.get_world_size() if dist is not None else 1, "collective_backend": "gloo" if dist is not None else None,
            "devices_observed": devices_seen, "distinct_devices_observed": n_distinct,
            "per_gpu_frames_per_s": [round(p[0] / p[1], 1) for p in per_rank], "roofline": None, "cpu_baseline": None}), flush=True)
    b.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
