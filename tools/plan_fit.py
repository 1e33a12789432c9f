#!/usr/bin/env python3
"""Least-squares fit behind rl_scene.cpp's plan_cost(): what a group bound, a (group, ray) pair and a (cluster, ray) pair cost
the trace kernel, from the measured throughput of builds that force one cluster size and one group size each.

The table below was measured on MI355X in round 3 (tools/ab.sh over libraries built with -DRL_CLUSTER_K=k -DRL_GROUP_GC=g,
two bench.py runs each, member loop rolled for every size); groups / pairs per ray are counted by the host mirror on the
segments of 20,000 paths of the same scene (tests/host_mirror: mirror_cull_counts).  Model, picoseconds per ray of a whole
MI355X:  T = T0(scene) + a_g * groups + pairs_g * (a_s + a_s4 * (G - 3)) + pairs_c * (a_c + a_m * K).
Usage: python tools/plan_fit.py   (prints the coefficients and the residuals; no GPU needed)"""
import numpy as np

# scene (0 built-in 1080p, 1 the 513-object scene from LDS), K, G, groups, group pairs per ray, cluster pairs per ray, Mrays/s
DATA = [
    (0, 8, 3, 13, 2.04, 1.87, 16206), (1, 8, 3, 21, 2.82, 2.51, 13582), (0, 8, 4, 10, 2.11, 1.87, 16147), (1, 8, 4, 16, 2.58, 2.51, 13856),
    (0, 10, 3, 11, 2.01, 1.73, 16328), (1, 10, 3, 17, 2.99, 2.43, 13652), (0, 10, 4, 8, 2.18, 1.73, 16219), (1, 10, 4, 13, 2.50, 2.43, 13898),
    (0, 12, 3, 9, 1.96, 1.92, 16088), (1, 12, 3, 14, 2.47, 2.45, 13844), (0, 12, 4, 7, 2.06, 1.92, 15911), (1, 12, 4, 11, 2.36, 2.45, 13835),
    (0, 14, 3, 8, 1.72, 1.58, 16490), (1, 14, 3, 12, 2.43, 2.42, 13686), (0, 14, 4, 6, 1.61, 1.58, 16498), (1, 14, 4, 9, 2.24, 2.42, 13728),
    (0, 16, 3, 7, 1.94, 1.58, 16168), (1, 16, 3, 11, 2.37, 2.32, 13593), (0, 16, 4, 5, 1.86, 1.58, 16192), (1, 16, 4, 8, 2.40, 2.32, 13520),
]


def main():
    d = np.array(DATA, dtype=float)
    t = 1e6 / d[:, 6]  # ps per ray
    x = np.column_stack([d[:, 0] == 0, d[:, 0] == 1, d[:, 3], d[:, 4], d[:, 4] * (d[:, 2] - 3), d[:, 5], d[:, 5] * d[:, 1]]).astype(float)
    coef = np.linalg.lstsq(x, t, rcond=None)[0]
    print("T0 built-in %.2f ps, T0 513 objects %.2f ps; per group %.3f; per group pair %.3f (+ %.3f per cluster beyond three); "
          "per cluster pair %.3f + %.4f per member" % tuple(coef))
    pred = x @ coef
    for row, ti, pi in zip(d, t, pred):
        print("scene %d  %2d per cluster, %d per group: measured %.2f ps, model %.2f, residual %+.2f" % (row[0], row[1], row[2], ti, pi, pi - ti))
    print("rms residual %.3f ps; spread of the measurements %.3f / %.3f ps" % (np.sqrt(np.mean((pred - t) ** 2)), t[d[:, 0] == 0].std(), t[d[:, 0] == 1].std()))
    print("plan_cost() uses 0.59, 2.13 + 0.83 (G - 3), 0.25 + 0.40 K: the fit of the counts before they were rounded to two digits for this\n"
          "table (the constant of the cluster pair moves by 0.09, worth 0.01 ps between two plans); the member term is the rolled loop's, the\n"
          "unrolled loops of the sizes on offer are ~0.05 cheaper per member, which does not change a choice")


if __name__ == "__main__":
    main()
