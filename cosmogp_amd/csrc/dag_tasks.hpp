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
//     far columns (right of the window), LAZILY (left-looking), see below
// Far columns: an update task costs ~10 us + ~17 us per panel of depth (tools/dag_probe.hip level 9), so a tile that
// collects 8+ panels per visit runs near the MFMA rate and is read-modified-written that many times less often.  But
// the updates of ONE tile are serial, so a tile that is p panels behind needs ~17 p us of latency to catch up whatever
// the number of idle workers: the further right a column is, the more it may lag.  Per step, for the far columns from
// the nearest one outwards:
//   (a) the `lead` columns next to the window are caught up (at most `dmax` panels per task), so that a column enters the
//       window at most ~`lead` panels behind;
//   (b) a column more than  dmax + slope * (distance beyond the lead zone)  panels behind gets one task of `dmax` panels:
//       the lag a column can still work off, `slope` panels per step, before the window reaches it;
//   (c) then, while the workers' time beside one step of the leaf lane (`step_us` x workers x `fill`) is not used up,
//       tiles at least `depth` panels behind get a task (nearest column first).
// The critical chain (~58 us per 128 columns, latency-bound: potrf + solve + update of the next diagonal tile on one CU) is
// then not held back by bulk work nobody needs yet, and that work fills the idle time of the chain-bound last part of
// the factorisation instead.  depth = 1: every column is kept current (one panel per update).
// The leaf lane applies panel k to the diagonal tile (k+1, k+1) itself, so the workers stop one panel short there.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

struct DagSchedule {
    long depth = 8, window = 4, lead = 2, dmax = 8, slope = 1, workers = 255;
    double step_us = 58., task_us = 10., panel_us = 17., trsm_us = 16., fill = 0.7;
    // tri_row0 >= 0: the row tiles from tri_row0 on hold IDENTITY rows (cgp_fit_grad, cgp_batch_loo: they become W = L^-T,
    // upper triangular): row tile tri_row0 + t is zero left of column tile t, so its tiles (., j < t) do not exist and the
    // others start with t panels "applied" (they are zero contributions)
    long tri_row0 = -1;
    long first(long i) const { return tri_row0 >= 0 && i >= tri_row0 ? i - tri_row0 : 0; }
};

// `steps` (optional) receives the segment boundaries of the list: [0] = 0, [1] = end of the solves of panel 0, then three per
// step k: end of the updates of column k+1, end of the look-ahead solves of panel k+1, end of the step.
inline void dag_build_tasks(long nb, long nr, const DagSchedule& sc, std::vector<int4>& t, std::vector<size_t>* steps = nullptr) {
    const long window = std::max<long>(1, sc.window), depth = std::max<long>(1, sc.depth), dmax = std::max(depth, sc.dmax);
    t.clear();
    std::vector<long> applied((size_t)nr * nb, 0);         // panels already scheduled onto tile (i, j)
    for (long i = 0; i < nr; ++i)
        for (long j = 0; j < nb; ++j) applied[(size_t)i * nb + j] = j < sc.first(i) ? nb + 1 : sc.first(i);   // nb + 1: no such tile
    double spent = 0.;                                     // worker time of this step's tasks (us)
    // one task for tile (i, j): panels [applied, min(k + 1, applied + cap)); returns false if there was nothing to do
    auto tile = [&](long k, long i, long j, long cap) {
        long& a = applied[(size_t)i * nb + j];
        long b = std::min(k + 1, a + cap);
        if (i == j && j == k + 1) b = std::min(b, k);       // the leaf lane's share
        if (b <= a) return false;
        t.push_back(make_int4((int)i, (int)j, (int)a, (int)((b << 1) | 1)));
        spent += sc.task_us + sc.panel_us * (double)(b - a);
        a = b;
        return true;
    };
    auto column = [&](long k, long j, long cap) { for (long i = j; i < nr; ++i) tile(k, i, j, cap); };
    if (steps) steps->assign(1, 0);
    for (long i = 2; i < nr; ++i)
        if (sc.first(i) == 0) t.push_back(make_int4((int)i, 0, 0, 0));
    if (steps) steps->push_back(t.size());
    for (long k = 0; k < nb; ++k) {
        if (k > 0 && steps) steps->push_back(t.size());
        spent = 0.;
        if (k + 1 < nb) column(k, k + 1, nb);
        if (steps) steps->push_back(t.size());
        if (k + 1 < nb)
            for (long i = k + 3; i < nr; ++i)
                if (sc.first(i) <= k + 1) { t.push_back(make_int4((int)i, (int)(k + 1), 0, 0)); spent += sc.trsm_us; }
        if (steps) steps->push_back(t.size());
        for (long j = k + 2; j < nb && j <= k + window; ++j) column(k, j, nb);
        if (depth == 1) {
            for (long j = k + window + 1; j < nb; ++j) column(k, j, nb);
            continue;
        }
        for (long j = k + window + 1; j < nb; ++j) {
            const long beyond = j - (k + window + sc.lead);        // <= 0: lead zone
            const long allowed = beyond <= 0 ? 0 : dmax + sc.slope * beyond;
            for (long i = j; i < nr; ++i)
                if (k + 1 - applied[(size_t)i * nb + j] > allowed) tile(k, i, j, dmax);
        }
        const double budget = sc.fill * sc.step_us * (double)sc.workers;
        for (long j = k + window + 1; j < nb && spent < budget; ++j)
            for (long i = j; i < nr && spent < budget; ++i)
                if (k + 1 - applied[(size_t)i * nb + j] >= depth) tile(k, i, j, dmax);
    }
    if (steps) steps->push_back(t.size());
}

// counters of one object before the launch: 0, except the identity row tiles (DagSchedule::tri_row0)
inline void dag_initial_state(long nb, long nr, const DagSchedule& sc, std::vector<unsigned>& st) {
    st.assign((size_t)nr * nb, 0u);
    for (long i = 0; i < nr; ++i)
        for (long j = sc.first(i); j < nb; ++j) st[(size_t)i * nb + j] = (unsigned)sc.first(i);
}

// The list for `nmat` same-shape objects: the single-object list of every object (object index in the upper half of .x),
// interleaved step by step.  Per step: every object's updates of column k+1 (they release the leaf lanes), then every
// object's remaining updates, then every object's look-ahead solves of panel k+1 -- by then the leaf lanes have factored
// their diagonal tiles, so no worker sits in a solve waiting for a factor while other objects have work (order = 0 keeps the
// single-object order inside a step: solves before the remaining updates).  The solves of panel k+1 and the updates with
// panel k touch different tiles, so moving one behind the other keeps every dependency pointing backwards; every tile sees
// its updates in the same order and the same grouping as in the single-object list, so an object's arithmetic does not
// depend on `nmat`.
inline void dag_build_tasks_batch(long nb, long nr, long nmat, const DagSchedule& sc, std::vector<int4>& t, int order = 1) {
    std::vector<int4> one;
    std::vector<size_t> m;
    dag_build_tasks(nb, nr, sc, one, &m);
    t.clear();
    t.reserve(one.size() * (size_t)nmat);
    auto emit = [&](size_t lo, size_t hi) {
        for (long b = 0; b < nmat; ++b)
            for (size_t q = lo; q < hi; ++q) {
                int4 d = one[q];
                d.x |= (int)(b << 16);
                t.push_back(d);
            }
    };
    emit(m[0], m[1]);
    for (size_t s = 1; s + 3 < m.size(); s += 3) {
        // segments [m[s], m[s+1]) column k+1, [m[s+1], m[s+2]) solves, [m[s+2], m[s+3]) the rest
        if (order == 0 || nmat == 1) emit(m[s], m[s + 3]);
        else { emit(m[s], m[s + 1]); emit(m[s + 2], m[s + 3]); emit(m[s + 1], m[s + 2]); }
    }
}
