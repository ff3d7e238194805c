"""Summarise the rocprofv3 --pmc passes of tools/gpu_pmc.sh: per kernel the per-launch mean of every raw counter and
the derived figures bench.py quotes (`roofline.from_committed_profile`).  Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md:
SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed
over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE are in KB and FETCH_SIZE under-reports
16-B/lane streams 2x on gfx950 (doubled here, as in profiles/r02-r03).  Usage: python tools/pmc_summarize.py <dir> -> text;
writes <dir>/pmc_summary.json."""
import collections, csv, glob, json, os, sys

N_XCD, N_CU, N_SIMD = 8, 256, 1024


def main():
    root = sys.argv[1]
    raw = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                raw[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, d in raw.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        n = max(len(v) for v in d.values())
        e = {"launches_seen": n, "raw_per_launch": m}
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        if cyc > 0:
            e["cycles_per_launch"] = cyc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                e["matrix_pipe_busy_pct_of_simd_cycles"] = 100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc)
            if "SQ_ACTIVE_INST_VALU" in m:
                e["valu_active_pct_of_simd_cycles"] = 100.0 * 4.0 * m["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc)
            if "SQ_INSTS_VALU" in m:
                e["valu_insts_per_simd_cycle"] = m["SQ_INSTS_VALU"] / (N_SIMD * cyc)
            if "SQ_INSTS_MFMA" in m:
                e["mfma_insts_per_simd_cycle"] = m["SQ_INSTS_MFMA"] / (N_SIMD * cyc)
            if "TA_TA_BUSY_sum" in m:
                e["ta_busy_pct"] = 100.0 * m["TA_TA_BUSY_sum"] / (N_CU * cyc)
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in m:
                e["l1_accesses_per_cu_cycle"] = m["TCP_TOTAL_CACHE_ACCESSES_sum"] / (N_CU * cyc)
        if m.get("SQ_WAVE_CYCLES"):
            for c, name in (("SQ_WAIT_INST_ANY", "wait_inst_any_pct_of_wave_cycles"), ("SQ_WAIT_ANY", "wait_any_pct_of_wave_cycles")):
                if c in m:
                    e[name] = 100.0 * m[c] / m["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_traffic_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        out[k] = e
    # which kernels the counters belong to: a digest of the kernel sources the profiled library was built from (bench.csrc_digest);
    # bench.py quotes these figures only while the tree's digest is still this one
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_meta"] = {"csrc_sha256": bench.csrc_digest(), "collected_by": "tools/gpu_pmc.sh (rocprofv3 --pmc passes of "
                    "`bench.py --steps 1 --warmup 1 --streams 1`, one pass per counter set, no tracing flags)"}
    json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    cols = ("cycles_per_launch", "matrix_pipe_busy_pct_of_simd_cycles", "valu_active_pct_of_simd_cycles", "wait_inst_any_pct_of_wave_cycles",
            "wait_any_pct_of_wave_cycles", "ta_busy_pct", "l1_accesses_per_cu_cycle", "hbm_traffic_bytes_per_launch")
    print(f"{'kernel':40s} {'n':>3s} " + " ".join(f"{c[:14]:>14s}" for c in cols))
    rows = [(k, e) for k, e in out.items() if not k.startswith("_")]
    for k, e in sorted(rows, key=lambda kv: -kv[1].get("cycles_per_launch", 0) * kv[1]["launches_seen"]):
        print(f"{k[:40]:40s} {e['launches_seen']:3d} " + " ".join(f"{e.get(c, float('nan')):14.4g}" for c in cols))


if __name__ == "__main__":
    main()
