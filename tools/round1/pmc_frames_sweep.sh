export TMPDIR=/tmp; R=$PWD
for F in 1 8; do for C in WRITE_SIZE FETCH_SIZE; do
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcx_${F}_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-per-step $F > /dev/null 2>&1
python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("gpurun_out/pmcx_${F}_$C/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "tl_encode" in r["Kernel_Name"] and r["Counter_Name"]=="$C"]
print("F=$F $C per launch KB", sum(v)/len(v), "per frame B", sum(v)/len(v)*1024/(4096*$F))
PY
done; done
