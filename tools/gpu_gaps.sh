#!/bin/bash
# Inter-kernel gaps of one engine's launch sequence: rocprofv3 kernel trace of a short single-stream bench, per evaluation the
# idle time between the end of a dispatch and the start of the next one on the stream, by predecessor kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_gaps; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o g -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --profile-steps 1 > $O/bench.json 2> $O/err.txt
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
rows = []
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
for f in glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy:" + r.get("Direction", "")[:20]))
rows.sort()
# the timed region: find runs of k_wrap .. k_finalize_energy; take evaluations 3..6 of the timed steps
starts = [i for i, r in enumerate(rows) if r[2].startswith("vssr::k_wrap")]
print("dispatches", len(rows), "evaluations", len(starts))
gap_by = collections.defaultdict(list)
busy = idle = 0
for a, b in zip(starts[3:8], starts[4:9]):
    seq = rows[a:b]
    for x, y in zip(seq, seq[1:] + [rows[b]]):
        g = (y[0] - x[1]) / 1e3
        gap_by[x[2]].append(g)
        idle += max(g, 0); busy += (x[1] - x[0]) / 1e3
a, b = starts[5], starts[6]
t0 = rows[a][0]
for x, y in zip(rows[a:b + 1], rows[a + 1:b + 2]):
    print(f"   +{(x[0] - t0) / 1e3:9.1f} us  dur {(x[1] - x[0]) / 1e3:8.1f}  gap-after {(y[0] - x[1]) / 1e3:7.1f}  {x[2]}")
n = 5
print(f"per evaluation: busy {busy / n:.1f} us, idle between dispatches {idle / n:.1f} us")
for k, v in sorted(gap_by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:42s} n/eval {len(v) / n:4.1f}  gap after it: mean {sum(v) / len(v):7.1f} us  total/eval {sum(v) / n:7.1f} us")
PY
find $O -name "*trace.csv" -delete
