// Host side of the single-launch factorisation (dag.hpp): the task list = a static schedule.
//
// Order (every dependency of a task points backwards in the list, or to the leaf lane whose own dependencies do):
//   TRSM(i, 0), i >= 2
//   for k = 0 .. nb-1:
//     updates that bring tile column k+1 up to date (panel k included)                  -- it feeds the leaf lane next
//     TRSM(i, k+1), i >= k+3          -- LOOK-AHEAD: the solves of panel k+1 run beside the bulk of panel k's updates
//                                        (the leaf lane factors (k+1, k+1) meanwhile), so a step never starts with
//                                        every worker waiting for the panel's solves
//     updates of the other window columns k+2 .. k+window (kept current panel by panel: they feed the leaf lane and
//       the next solves within a few steps)
//     the column that enters the window `lead` steps from now is brought up to date (its catch-up on entry stays short)
//     far columns, nearest first, as many as fit beside one step of the leaf lane: each one brought up to date with ALL
//       the panels it is behind (one update per tile, as deep as the column is behind)
// Far columns are updated LAZILY (left-looking): an update task costs ~10 us + ~17 us per panel of depth
// (tools/dag_probe.hip level 9), so a tile that collects 8-16 panels per visit runs near the MFMA rate and is
// read-modified-written that many times less often; and a column is never given two updates in flight (the second would
// only be claimed by a worker that spins until the first has finished).  What is emitted per step is what the workers can
// do while the leaf lane runs one step (`step_us`): the critical chain (~58 us per 128 columns, latency-bound) is not held
// back by bulk work it does not need yet, and the bulk fills its shadow.  Only columns at least `depth` panels behind
// are picked (half of it in a second pass when the workers would otherwise idle).
// While the trailing matrix is small (fewer than `deep_min` columns left) every column is kept current: deep catch-up
// updates would only add latency to a chain-bound phase.  depth = 1: every update applies one panel.
// The leaf lane applies panel k to the diagonal tile (k+1, k+1) itself, so the workers stop one panel short there.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

struct DagSchedule {
    long depth = 8, window = 4, deep_min = 36, workers = 255, lead = 2;
    double step_us = 58., task_us = 10., panel_us = 17., trsm_us = 16., fill = 1.0;
};

inline void dag_build_tasks(long nb, long nr, const DagSchedule& sc, std::vector<int4>& t) {
    const long window = sc.window < 1 ? 1 : sc.window, depth = sc.depth < 1 ? 1 : sc.depth;
    t.clear();
    std::vector<long> applied(nb, 0);          // panels already scheduled onto tile column j
    double spent = 0.;                         // worker time of this step's tasks (us)
    auto column = [&](long k, long j) {        // bring column j up to date with every finished panel (through k)
        const long k0 = applied[j], k1 = k + 1;
        if (k0 >= k1) return;
        for (long i = j; i < nr; ++i) {
            long b = k1;
            if (i == j && j == k + 1) b = k;                 // the leaf lane's share
            if (b > k0) {
                t.push_back(make_int4((int)i, (int)j, (int)k0, (int)((b << 1) | 1)));
                spent += sc.task_us + sc.panel_us * (double)(b - k0);
            }
        }
        applied[j] = k1;
    };
    for (long i = 2; i < nr; ++i) t.push_back(make_int4((int)i, 0, 0, 0));
    for (long k = 0; k < nb; ++k) {
        spent = 0.;
        if (k + 1 < nb) {
            column(k, k + 1);
            for (long i = k + 3; i < nr; ++i) { t.push_back(make_int4((int)i, (int)(k + 1), 0, 0)); spent += sc.trsm_us; }
        }
        for (long j = k + 2; j < nb && j <= k + window; ++j) column(k, j);
        if (depth == 1 || nb - k <= sc.deep_min) {
            for (long j = k + window + 1; j < nb; ++j) column(k, j);
            continue;
        }
        if (k + window + sc.lead < nb) column(k, k + window + sc.lead);
        const double budget = sc.fill * sc.step_us * (double)sc.workers;
        for (long need = depth; need >= 1 && spent < budget; need = need > 1 ? (need + 1) / 2 - (need == 1) : 0) {
            for (long j = k + window + 1; j < nb && spent < budget; ++j)
                if (k + 1 - applied[j] >= need) column(k, j);
            if (need == 1) break;
        }
    }
}
