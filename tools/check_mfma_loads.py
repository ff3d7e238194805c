#!/usr/bin/env python3
"""Build-time check of the MFMA / load ordering rules (see mfma_load_fence in painn_edge_mfma.hip) on the emitted gfx950
ISA.  Rule 1 (files with fences): no load (LDS, global, scratch, flat) may sit between the first MFMA of a step and the
'; mfma_load_fence' marker that follows it.  Rule 2 (every file): no load inside a dense MFMA block, i.e. between two
MFMAs that are at most DENSE_GAP instructions apart; a '; gemm16_group_end' marker (painn_node_mfma.hip, pipelined GEMMs)
closes a block: the fragment loads of the next chunk group legitimately follow it.  Rule 3 (files with fences, i.e. the edge
kernels): inside a dense block an MFMA never accumulates into the destination of one of the two MFMAs in front of it -- the
producer of its accumulator is at least three matrix instructions back (dependent MFMAs issued closer stall inside the pipe
and read their other sources late; profiles/r01/NOTES_mfma_hazards.md).  Rule 4 (same files): a product that is not the first of its
chain never overwrites its own A / B operand.  Rule 5 (every file): no FLAT memory instructions.  Rule 6 (edge kernels): the static count
of matrix instructions per 16-slot step is printed and checked against the K-dense form's 2 per tile.  Usage: check_mfma_loads.py [file.hip ...]  (exit 1 on violation)."""
import os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "surface-sampling_amd", "csrc")
LOAD = re.compile(r"^\s*(ds_read|ds_load|ds_bpermute|ds_permute|ds_swizzle|global_load|buffer_load|scratch_load|flat_load)")
DENSE_GAP = 6
DEFAULT_FILES = ("painn_edge_mfma.hip", "painn_node_mfma.hip", "painn_l0.hip")


def check(hip):
    with tempfile.TemporaryDirectory() as tmp:
        extra = ["-fno-slp-vectorize"] if hip.endswith("painn_edge_mfma.hip") else []   # per-file flag of csrc/Makefile
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", *extra,
                        "-save-temps", "-c", hip, "-o", os.path.join(tmp, "x.o")], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        lines = open(os.path.join(tmp, asm)).read().splitlines()
    bad = 0
    kernel = None
    open_group = None          # line index of the first MFMA since the last fence
    loads = []
    groups = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\w+):", l)
        if m:
            kernel, open_group, loads = m.group(1), None, []
        t = l.strip()
        if t.startswith("v_mfma"):
            if open_group is None:
                open_group, loads = i, []
        elif "mfma_load_fence" in t:
            if open_group is not None:
                groups += 1
                if loads:
                    bad += 1
                    print(f"VIOLATION in {kernel}: {len(loads)} load(s) between MFMA at line {open_group} and fence at {i}:")
                    for j in loads[:6]:
                        print("    ", lines[j].strip())
            open_group = None
        elif open_group is not None and LOAD.match(l):
            loads.append(i)
        elif t.startswith("s_endpgm"):
            if open_group is not None and kernel and "edge" in kernel and "mfma" in kernel:
                print(f"note: {kernel}: MFMA group at line {open_group} without a following fence")
            open_group = None
    # rule 2: dense MFMA blocks must be free of loads
    kernel, last_mfma, pending, dense_blocks = None, None, [], 0
    instr_idx = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\w+):", l)
        if m:
            kernel, last_mfma, pending = m.group(1), None, []
        t = l.strip()
        if "gemm16_group_end" in t:
            last_mfma, pending = None, []
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        instr_idx += 1
        if t.startswith("v_mfma"):
            if last_mfma is not None and instr_idx - last_mfma <= DENSE_GAP:
                dense_blocks += 1
                if pending:
                    bad += 1
                    print(f"VIOLATION in {kernel}: load inside a dense MFMA block near line {i}:")
                    for j in pending[:4]:
                        print("    ", lines[j].strip())
            last_mfma, pending = instr_idx, []
        elif LOAD.match(l) and last_mfma is not None and instr_idx - last_mfma <= DENSE_GAP:
            pending.append(i)
        elif last_mfma is not None and instr_idx - last_mfma > DENSE_GAP:
            pending = []
    # rule 3: accumulator chains of the fenced kernels keep their distance
    if groups:
        kernel, recent, last_idx, idx = None, [], None, 0
        for i, l in enumerate(lines):
            m = re.match(r"^(_ZN\w+):", l)
            if m:
                kernel, recent, last_idx = m.group(1), [], None
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            idx += 1
            if t.startswith("v_mfma"):
                dst = t.split()[1].rstrip(",")
                if last_idx is not None and idx - last_idx > DENSE_GAP:
                    recent = []
                if dst in recent[-2:]:
                    bad += 1
                    print(f"VIOLATION in {kernel}: MFMA at line {i} accumulates into {dst}, written {len(recent) - recent.index(dst)} MFMA(s) earlier")
                recent.append(dst)
                last_idx = idx
    # rule 4: a product that continues an accumulator chain never writes over its own A / B operand.  Known-good (soaked): a chain's
    # first product (accumulator constant 0) overwrites its B operand; later products accumulate in place or into other registers
    # (also registers the instructions in front of them read).  Observed bad (round 3, packed-fp32 experiment, 8-feature forward
    # kernel): `v_mfma D = A, B, C` with D == B and C a register -> run-to-run differences in the forces.
    if groups:
        def regs(tok):
            m = re.match(r"^[va]\[(\d+):(\d+)\]$", tok)
            return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()
        kernel = None
        for i, l in enumerate(lines):
            m = re.match(r"^(_ZN\w+):", l)
            if m:
                kernel = m.group(1)
            t = l.strip()
            if t.startswith("v_mfma"):
                ops = [o.strip() for o in t.split(None, 1)[1].split(",")][:4]
                dst, a, b, c = (regs(o) for o in ops)
                if c and ops[0] != ops[3] and dst & (a | b):
                    bad += 1
                    print(f"VIOLATION in {kernel}: MFMA at line {i} continues a chain and overwrites its own operand: {t}")
    # rule 5: no FLAT memory instructions.  A flat access ticks vmcnt AND lgkmcnt; while one is pending every wait the compiler inserts
    # is a full drain -- pending flat weight loads in front of the forward neighbor loop turned its header wait into vmcnt(0) for
    # every iteration (profiles/r03/NOTES_edge_traffic.md).  Pointers read out of tables in memory must be dereferenced through the
    # global address space (gload4u / gload4f / gload1f / gload_u32x4).
    kernel = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\w+):", l)
        if m:
            kernel = m.group(1)
        if re.match(r"^\s*flat_(load|store|atomic)", l):
            bad += 1
            print(f"VIOLATION in {kernel}: FLAT memory instruction at line {i}: {l.strip()}")
    # rule 6 (edge kernels): matrix instructions per 16-slot step of the hot loops = v_mfma between two '; arrival_fence' markers.
    # The K-dense radial filter (round 6) issues 2 per tile: 6 / 4 / 2 in the forward kernels with 16- / 8- / 4-feature slices,
    # 12 / 8 / 4 in the reverse kernels (filter + radial derivative); more means a product fell back to its own instruction.
    if groups:
        kernel, count, steps = None, None, {}
        for l in lines:
            m = re.match(r"^(_ZN\w+):", l)
            if m:
                kernel, count = m.group(1), None
            t = l.strip()
            if "; arrival_fence" in t:
                if count:
                    steps.setdefault(kernel, set()).add(count)
                count = 0
            elif t.startswith("v_mfma") and count is not None:
                count += 1
            elif t.startswith("s_endpgm"):
                if count:
                    steps.setdefault(kernel, set()).add(count)
                count = None
        for k, v in steps.items():
            is_bwd = "k_edge_bwd" in k
            nf = int(re.search(r"mfmaILi(\d)E", k).group(1))
            limit = (4 if is_bwd else 2) * {4: 3, 2: 2, 1: 1}[nf]   # 2 per tile; tiles per table: 3 / 2 / 1 (EdgeGeo::NT)
            tag = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.split("(")[0].replace("void vssr::", "")
            print(f"    {tag:44s} matrix instructions per 16-slot step: {sorted(v)} (limit {limit})")
            if max(v) > limit:
                bad += 1
                print(f"VIOLATION in {k}: {max(v)} matrix instructions per step, expected <= {limit}")
    print(f"{os.path.basename(hip)}: {groups} MFMA groups checked, {dense_blocks} dense MFMA pairs, {bad} violations")
    return bad


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in DEFAULT_FILES]
    sys.exit(1 if sum(check(f) for f in files) else 0)
