#!/usr/bin/env python3
"""Per-kernel resource table of a device-only compile of rl_api.hip (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: tools/resource_usage.py [-DNAME=VALUE ...] [--dir csrc-dir]   -- extra flags go to hipcc behind the Makefile's."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
csrc = os.path.join(ROOT, "robigo_luculenta_amd", "csrc")
if "--dir" in args:
    i = args.index("--dir"); csrc = args[i + 1]; del args[i:i + 2]
make = open(os.path.join(csrc, "Makefile")).read()
flags = re.search(r"^FLAGS = (.*?)\n(?!\s)", make, re.S | re.M).group(1).replace("\\\n", " ")
flags = flags.replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "").split()
run = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ['-DRL_BUILD_ID="x"', "--cuda-device-only", "-c", "-o", "/dev/null", "rl_api.hip",
                      "-Rpass-analysis=kernel-resource-usage"] + args, cwd=csrc, capture_output=True)
if run.returncode:
    sys.exit(run.stderr.decode()[-3000:])
name, rows = None, {}
for line in run.stderr.decode().splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); rows[name] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[", line)
    if m and name:
        rows[name][m.group(1).strip()] = int(m.group(2))
print("%-44s %5s %5s %6s %6s %7s %4s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "occ"))
for k, v in rows.items():
    m = re.match(r"_Z(?:15rl_trace_kernel|20rl_trace_kernel_open)ILi([012])ELb([01])ELb([01])E", k)
    short = k[:44]
    if m:
        short = ("open " if "kernel_open" in k else "plain") + " stage=%s fused=%s cyl=%s" % m.groups()
    print("%-44s %5d %5d %6d %6d %7d %4d" % (short, v.get("VGPRs", -1), v.get("SGPRs", -1), v.get("VGPRs Spill", 0), v.get("SGPRs Spill", 0),
                                            v.get("ScratchSize", 0), v.get("Occupancy", 0)))
