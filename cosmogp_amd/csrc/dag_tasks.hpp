// Host side of the single-launch factorisation (dag.hpp): the task list.
//
// Order (every dependency of a task points backwards in the list, or to the leaf lane whose own dependencies do):
//   TRSM(i, 0), i >= 2
//   for k = 0 .. nb-1:
//     updates that bring tile column k+1 up to date (panel k included)          -- it feeds the leaf lane next
//     TRSM(i, k+1), i >= k+3          -- LOOK-AHEAD: the solves of panel k+1 run beside the bulk of panel k's updates
//                                        (the leaf lane factors (k+1, k+1) meanwhile), so a step never starts with
//                                        every worker waiting for the panel's solves
//     updates of the columns further right
// depth = 1: every update applies one panel, UPD(i, j, k, k+1).  depth = d > 1: a column more than d panels right of
// the current one collects d panels per update (UPD(i, j, k-d+1, k+1) when k+1 is a multiple of d: fewer, deeper
// read-modify-writes of C), the d columns next to the panel are kept current panel by panel.
// The leaf lane applies panel k to the diagonal tile (k+1, k+1) itself, so the workers stop one panel short there.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

inline void dag_build_tasks(long nb, long nr, long depth, std::vector<int4>& t) {
    t.clear();
    std::vector<long> applied(nb, 0);          // panels already scheduled onto tile column j
    auto column = [&](long k, long j) {        // step k: updates of column j
        long k0 = -1;
        const long k1 = k + 1;
        if (depth == 1 || j - k <= depth) k0 = applied[j];
        else if ((k + 1) % depth == 0) k0 = applied[j];
        if (k0 < 0 || k0 >= k1) return;
        for (long i = j; i < nr; ++i) {
            long b = k1;
            if (i == j && j == k + 1) b = k;                 // the leaf lane's share
            if (b > k0) t.push_back(make_int4((int)i, (int)j, (int)k0, (int)((b << 1) | 1)));
        }
        applied[j] = k1;
    };
    for (long i = 2; i < nr; ++i) t.push_back(make_int4((int)i, 0, 0, 0));
    for (long k = 0; k < nb; ++k) {
        if (k + 1 < nb) {
            column(k, k + 1);
            for (long i = k + 3; i < nr; ++i) t.push_back(make_int4((int)i, (int)(k + 1), 0, 0));
        }
        for (long j = k + 2; j < nb; ++j) column(k, j);
    }
}
