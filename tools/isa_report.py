#!/usr/bin/env python
"""Static report on the gfx950 code of the engine's kernels (no GPU needed).

    python tools/isa_report.py [dfsph wcsph pbd runtime system] > profiles/rNN_isa_report.txt

For every kernel: VGPRs, SGPRs, occupancy (waves/SIMD), scratch bytes, LDS bytes, code bytes; for the
sweep kernels additionally the instruction mix of the main row loop (the innermost depth-1 loop with
the most instructions): VALU (packed fp32 / transcendental counted separately), SALU, branches,
vector-memory loads split into row loads and gathers with a uniform base (`saddr` form).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cpp-fluid-particles_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S"]


def demangle(names):
    filt = "c++filt"
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def loop_mix(body):
    """instruction mix of the largest depth-1 inner loop of a kernel body (list of lines)"""
    heads = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=1" in l]
    best = None
    for h in heads:
        label = body[h].split(":")[0].strip().lstrip(".L")          # e.g. BB19_51
        tagged = [i for i in range(h, len(body)) if ("Header=%s " % label) in body[i] + " "]
        last = tagged[-1] if tagged else h
        end = last + 1
        while end < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[end]):   # to the end of that block
            end += 1
        seg = [l.strip() for l in body[h:end] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        if best is None or len(seg) > len(best):
            best = seg
    if not best:
        return None
    mix = {"total": len(best), "valu": 0, "valu_pk": 0, "valu_trans": 0, "salu": 0, "branch": 0, "vmem_row": 0,
           "gather_saddr": 0, "gather_vaddr": 0, "waitcnt": 0, "lds": 0, "row_loads_split": 0}
    # the row loads of a trip must be in flight TOGETHER: a "load, wait, load, wait" sequence (what the compiler falls back to
    # when the kernel is one VGPR over its budget; r04: +16 % on the dominant launch) shows as waits between the row loads
    seen_row, pending_wait = 0, False
    for l in best:
        op = l.split()[0]
        if op.startswith("global_load") and op.endswith("dword") and "off" in l:
            if seen_row and pending_wait:
                mix["row_loads_split"] += 1
            seen_row += 1; pending_wait = False
        elif op == "s_waitcnt" and "vmcnt" in l and seen_row:
            pending_wait = True
        elif op.startswith("global_load"):
            break                                    # the gathers begin: the row loads of this trip are over
    for l in best:
        op = l.split()[0]
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            mix["branch"] += 1
        elif op == "s_waitcnt":
            mix["waitcnt"] += 1
        elif op.startswith("s_"):
            mix["salu"] += 1
        elif op.startswith("global_load") or op.startswith("buffer_load"):
            if op.endswith("dword") and "off" in l:
                mix["vmem_row"] += 1
            elif re.search(r",\s*s\[\d+:\d+\]", l):
                mix["gather_saddr"] += 1
            else:
                mix["gather_vaddr"] += 1
        elif op.startswith("ds_"):
            mix["lds"] += 1
        elif op.startswith("v_"):
            mix["valu"] += 1
            if op.startswith("v_pk_"):
                mix["valu_pk"] += 1
            if re.match(r"v_(sqrt|rcp|rsq|exp|log|sin|cos)_", op):
                mix["valu_trans"] += 1
    return mix


def report(unit):
    src = os.path.join(CSRC, unit + ".hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, unit + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s+; @", l)]
    names = demangle([n for _, n in starts])
    print("== %s.hip" % unit)
    print("%-86s %5s %5s %4s %7s %6s %7s" % ("kernel", "vgpr", "sgpr", "occ", "scratch", "lds", "code B"))
    for k, (i, n) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = lines[i:end]
        def meta(key):
            for l in body:
                m = re.match(r";\s*%s:?\s*=?\s*(\d+)" % key, l.strip())
                if m:
                    return int(m.group(1))
            return -1
        short = re.sub(r"sphx::", "", names[n])
        short = re.sub(r"HIP_vector_type<float, (\d)u>", r"float\1", short)
        print("%-86s %5d %5d %4d %7d %6d %7d" % (short[:86], meta("NumVgprs"), meta("NumSGPRsForWavesPerEU"), meta("Occupancy"),
                                                 meta("ScratchSize"), meta("LDSByteSize"), meta("codeLenInByte")))
        if re.search(r"k_run_op|k_rate|k_dfsph_head|k_build_list", short):
            m = loop_mix(body)
            if m:
                print("      main loop (4 row entries per trip): %d instr | VALU %d (packed %d, transcendental %d) | SALU %d | "
                      "branches %d | waitcnt %d | row loads %d (%s) | gathers uniform-base %d, per-lane-base %d | LDS %d"
                      % (m["total"], m["valu"], m["valu_pk"], m["valu_trans"], m["salu"], m["branch"], m["waitcnt"],
                         m["vmem_row"], "issued together" if m["row_loads_split"] == 0 else "SERIALISED by %d waits" % m["row_loads_split"],
                         m["gather_saddr"], m["gather_vaddr"], m["lds"]))
    print()


if __name__ == "__main__":
    units = sys.argv[1:] or ["dfsph", "wcsph", "pbd", "runtime", "system"]
    print("static gfx950 code report (tools/isa_report.py; hipcc -S with the product's flags)")
    print("main-loop counts are STATIC: each of the 4 pair terms of a trip appears twice, as the exact fast path that")
    print("normally runs (~60 VALU) and as the out-of-line plain-operator variant (~105 VALU) taken wave-uniformly when")
    print("some lane's pair needs it.  'gathers per-lane-base' should be 0: every neighbour gather uses saddr + 32-bit offset.\n")
    for u in units:
        report(u)
