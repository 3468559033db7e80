export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/ifl
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/ifl -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/ifl.log 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/ifl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tl_frame_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
d={k:sum(v)/len(v) for k,v in acc.items()}
print(d)
if d.get("SQ_IFETCH"): print("avg fetch latency (cycles, LEVEL/IFETCH):", d["SQ_IFETCH_LEVEL"]/d["SQ_IFETCH"], " fetch-outstanding share of wave cycles:", d["SQ_IFETCH_LEVEL"]/d["SQ_WAVE_CYCLES"])
PY
