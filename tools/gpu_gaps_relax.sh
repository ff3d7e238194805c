#!/bin/bash
# idle time between dispatches inside a lock-step relaxation (kernel + memory-copy trace of tools/bench_relax.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gaps_relax; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o g -- python3 tools/bench_relax.py > $O/relax.jsonl 2> $O/err.txt
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
rows = []
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
for f in glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy:" + r.get("Direction", "")[:24]))
rows.sort()
steps = [i for i, r in enumerate(rows) if r[2].startswith("vssr::k_bfgs_step")]
print("dispatches", len(rows), "bfgs steps", len(steps))
# one BFGS relaxation = 21 consecutive steps; look at iterations 5 .. 16 of the first one
gap_by = collections.defaultdict(list)
for a, b in zip(steps[4:16], steps[5:17]):
    seq = rows[a:b + 1]
    for x, y in zip(seq, seq[1:]):
        gap_by[x[2]].append((y[0] - x[1]) / 1e3)
    if a == steps[8]:
        t0 = seq[0][0]
        for x, y in zip(seq, seq[1:]):
            print(f"   +{(x[0] - t0) / 1e3:9.1f} us  dur {(x[1] - x[0]) / 1e3:8.1f}  gap-after {(y[0] - x[1]) / 1e3:7.1f}  {x[2]}")
n = 12
for k, v in sorted(gap_by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:42s} n/iter {len(v) / n:4.1f}  gap after it: mean {sum(v) / len(v):7.1f} us  total/iter {sum(v) / n:7.1f} us")
PY
find $O -name "*trace.csv" -delete
