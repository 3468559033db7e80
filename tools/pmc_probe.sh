export TMPDIR=/tmp; R=$PWD
run() { # psy F counters...
  P=$1; F=$2; shift 2
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/probe_${P}_${F} -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-per-step $F --psy $P > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/probe_${P}_${F}/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "tl_encode" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("psy $P F=$F", {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
  rm -rf gpurun_out/probe_${P}_${F}
}
run 1 8 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum
run 1 8 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
run 0 8 WRITE_SIZE
run 0 1 WRITE_SIZE
