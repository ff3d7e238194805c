#!/bin/bash
# HBM traffic of the edge kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "k_edge_(fwd|bwd)_mfma" --output-format csv -d gpurun_out/pmc_$c -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_$c.err
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c] = {"n": len(v), "mean_raw": sum(v) / len(v)}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/traffic_raw.json", "w"), indent=1)
PY
