#!/bin/bash
# What one more instruction of a kind costs the trace kernel: A/B of builds that add a block of instructions of ONE kind to every
# iteration of the persistent loop (RL_EXP_EXTRA in rl_kernels.hip.h; timing only, the results are unchanged) against the tree's
# library.  Build the alternatives first:
#   for i in 1 2 3 4 5 6 7 8 9 10 11 12; do make -C robigo_luculenta_amd/csrc OUT=../librl_alt_x$i.so EXTRA=-DRL_EXP_EXTRA=$i; done
# Usage (through gpurun): bash tools/issue_cost_probes.sh <tag>   -> gpurun_out/<tag>/issue_cost_probes.txt
set -u
TAG=${1:-probes}
OUT=gpurun_out/$TAG; mkdir -p $OUT
N=3 BENCH_ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-live-counters --no-others" bash tools/ab3.sh x1 x2 x5 x7 x6 x4 x11 x8 x9 x3 x10 x12 > $OUT/ab.txt 2>&1
python - $OUT/ab.txt > $OUT/issue_cost_probes.txt <<'PY'
import collections, sys
v = collections.defaultdict(list)
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) >= 3 and f[1] == "demo":
        v[f[0]].append(float(f[2]))
base = sum(v["tree"]) / len(v["tree"])
what = {"x1": (256, "s_add_u32 (one dependent chain)"), "x2": (256, "v_add_u32 (four chains)"), "x5": (256, "v_add_u32 (one dependent chain)"),
        "x7": (256, "v_fma_f32 (four chains)"), "x6": (128, "v_pk_fma_f32 (four chains)"), "x4": (256, "s_nop 0"),
        "x11": (128, "s_and_saveexec_b64 / s_or_b64 exec (64 pairs)"), "x8": (64, "s_branch, taken"), "x9": (64, "s_cbranch_scc1, not taken"),
        "x3": (16, "ds_read_b32 + s_waitcnt lgkmcnt(0): exposed LDS round trips"), "x10": (4, "batches of nine ds_bpermute_b32 + one wait (+ ~14 VALU each)"),
        "x12": (4, "batches of 2 x ds_read_b128 + ds_read_b32 gathers + one wait (+ ~14 VALU each)")}
print("built-in scene, 1920x1080, %.0f Mrays/s without a probe (mean of %d runs); each probe adds N instructions of one kind to every iteration of the" % (base, len(v["tree"])))
print("persistent loop (64 rays, ~4.0 k instructions, ~31 k cycles of a wave's time):")
print("%-78s %6s %9s %9s %14s" % ("probe", "N", "Mrays/s", "slower", "per instruction"))
for k, (n, name) in what.items():
    if not v[k]:
        continue
    m = sum(v[k]) / len(v[k])
    slow = base / m - 1.0
    print("%-78s %6d %9.0f %8.2f %% %12.4f %%" % (name, n, m, 100 * slow, 100 * slow / n))
PY
cat $OUT/issue_cost_probes.txt
