#!/usr/bin/env python3
"""Static census of the trace kernel's ISA by region (round 5).

Compiles rl_api.hip device-only with -DRL_MARK -- the region timers' boundaries (RL_T0 / RL_T1 in rl_kernels.hip.h) become
comments of the assembly -- and counts, between the markers of one instantiation, the instructions by kind: vector, scalar,
branches, LDS, waits, and the v_mov copies (the phi copies of a loop or a join are where the compiler wastes a wave's issue
slots: the stash hand-out's sixteen, the hit completion's dozen were found this way).  Static counts: a region that is
inlined several times (the round handlers) is counted as often as it appears.  No GPU needed.

Usage: tools/region_census.py [mangled-kernel-prefix]   (default: the bench's rl_trace_kernel<STAGE_ALL, FUSED, no CYL>)
       tools/region_census.py --resources               (registers, spills, scratch of every trace instantiation)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "robigo_luculenta_amd", "csrc")
FLAGS = ["-O2", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-munsafe-fp-atomics", "-fno-slp-vectorize", "-mllvm", "-disable-machine-licm", '-DRL_BUILD_ID="census"', "-DRL_MARK",
         "--cuda-device-only", "-S"]  # the Makefile's flags


def assembly():
    out = os.path.join(tempfile.mkdtemp(prefix="rl_census_"), "rl_api.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, "rl_api.hip"], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kind(op):
    if op.startswith("v_mov_b"):
        return "v_mov"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "s_load", "scratch_")):
        return "mem"
    return "vector" if op.startswith("v_") else "scalar"


def census(lines, prefix):
    inside, region = False, "prologue"
    counts = collections.OrderedDict()
    for l in lines:
        if l.startswith(prefix) and l.rstrip().split(":")[0].startswith(prefix) and ":" in l:
            inside = True
            continue
        if not inside:
            continue
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        m = re.match(r"; RL_MARK (begin|end) (\w+)", t)
        if m:
            region = m.group(2) if m.group(1) == "begin" else "behind " + m.group(2)
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        counts.setdefault(region, collections.Counter())[kind(t.split()[0])] += 1
    return counts


def resources(lines):
    name, rows = None, {}
    for l in lines:
        m = re.match(r"\s+\.name:\s+(_Z\S+)", l)
        if m:
            name = m.group(1)
        for k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
            m = re.match(r"\s+\.%s:\s+(\d+)" % k, l)
            if m and name:
                rows.setdefault(name, {})[k] = int(m.group(1))
    for n, r in rows.items():
        m = re.match(r"_Z\d+(rl_trace_kernel(?:_open)?)ILi(\d)ELb(\d)ELb(\d)E", n)
        if m:
            print("%-34s sgpr %3d (spilled %2d)  vgpr %3d (spilled %d)  scratch %d" % ("%s<%s,%s,%s>" % m.groups(), r["sgpr_count"],
                  r["sgpr_spill_count"], r["vgpr_count"], r["vgpr_spill_count"], r["private_segment_fixed_size"]))


if __name__ == "__main__":
    lines = assembly()
    if "--resources" in sys.argv:
        resources(lines)
        sys.exit(0)
    prefix = next((a for a in sys.argv[1:] if not a.startswith("-")), "_Z15rl_trace_kernelILi2ELb1ELb0E")
    kinds = ["vector", "v_mov", "scalar", "branch", "lds", "mem", "wait"]
    print("%-22s %7s " % ("region (static)", "total") + " ".join("%7s" % k for k in kinds))
    total = collections.Counter()
    for region, c in census(lines, prefix).items():
        print("%-22s %7d " % (region, sum(c.values())) + " ".join("%7d" % c[k] for k in kinds))
        total.update(c)
    print("%-22s %7d " % ("all", sum(total.values())) + " ".join("%7d" % total[k] for k in kinds))
