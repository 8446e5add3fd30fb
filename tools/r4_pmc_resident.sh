#!/bin/bash
# round 4: HBM counters (FETCH_SIZE / WRITE_SIZE, separate passes) for the resident decoder kernel ALONE - counter collection limited to
# it by --kernel-include-regex, so that the attention kernel beside it is not serialised behind it (if the profiler serialises all
# dispatches regardless, the forward times out into the launch-per-step kernels and no row of the kernel appears)
set -u
: "${GRAFT_REPO_ROOT:?}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4pmcres
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c --kernel-include-regex "decoder_resident_kernel" -d $O/pmc_$c -- python3 $R/tools/run_config.py tf 32 2 > $O/run_$c.log 2>&1; echo "$c rc=$?"
  tail -2 $O/run_$c.log
done
cd $R
python - <<'P'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/r4pmcres/pmc_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(c, k, len(v), [round(x) for x in v][:8])
    ks = glob.glob(f"gpurun_out/r4pmcres/pmc_{c}/**/*kernel_trace.csv", recursive=True)
    for f in ks:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "decoder_resident_kernel" in r["Kernel_Name"]]
        print(c, "decoder_resident_kernel durations (us):", [round(x) for x in d])
P
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
