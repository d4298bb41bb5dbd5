// host check of the task list: every dependency points backwards; every tile gets every panel exactly once
#include "../../cosmogp_amd/csrc/dag_tasks.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    int bad = 0;
    for (long nb : {2, 3, 8, 17, 32, 64, 128})
        for (long tall : {0, 1, 20, -1})      // -1: the gradient / leave-one-out layout: one dense tall tile row + nb identity row tiles
            for (long depth : {1, 2, 4, 8, 16})
                for (long window : {1, 2, 4, 8})
                    for (long dm : {0, 12, 36})
                        for (double fill : {0.3, 1.0, 2.0}) {
                            DagSchedule sc; sc.depth = depth; sc.window = window; sc.fill = fill; sc.lead = 1 + (depth + window) % 4; sc.dmax = depth + (window % 3) * 4; sc.slope = 1 + dm / 20;
                            const long nr = tall >= 0 ? nb + tall : 2 * nb + 1;
                            if (tall < 0) sc.tri_row0 = nb + 1;
                            // several objects side by side (every object is checked with its own counters and leaf lane),
                            // both interleaving orders
                            for (int variant = 0; variant < (nb <= 32 ? 3 : 1); ++variant) {
                            const long nmat = variant == 0 ? 1 : 3;
                            std::vector<int4> tall_list;
                            dag_build_tasks_batch(nb, nr, nmat, sc, tall_list, variant == 2 ? 0 : 1);
                            if (nmat == 1) { std::vector<int4> ref; dag_build_tasks(nb, nr, sc, ref); if (ref.size() != tall_list.size()) { printf("SIZE MISMATCH\n"); ++bad; } }
                            for (long ob = 0; ob < nmat; ++ob) {
                            std::vector<int4> t;
                            for (const int4& d : tall_list) if ((d.x >> 16) == ob) { int4 e = d; e.x &= 0xffff; t.push_back(e); }
                            std::vector<unsigned> st0;
                            dag_initial_state(nb, nr, sc, st0);
                            std::vector<int> state(st0.begin(), st0.end());       // simulated in list order with the leaf lane as far ahead as it can be
                            int leaf_k = 0;                            // leaf has finished potrf(leaf_k - 1) and near trsm
                            // the leaf lane exactly as dag_leaf_step orders its waits and publishes (sub = how far step leaf_k got)
                            int sub = 0;
                            auto leaf_advance = [&]() {
                                for (;;) {
                                    const int k = leaf_k;
                                    if (k >= nb) return;
                                    if (sub == 0) {            // factor (k, k): workers' updates through panel k-2, X from the leaf itself
                                        if (state[k * nb + k] < (k > 0 ? k - 1 : 0)) return;
                                        if (k > 0 && state[k * nb + (k - 1)] != k) return;
                                        sub = 1;
                                    }
                                    if (sub == 1) {            // wait for the tile below, THEN publish the factor
                                        if (k + 1 < nr && state[(k + 1) * nb + k] < k) return;
                                        state[k * nb + k] = k + 1;
                                        sub = 2;
                                    }
                                    if (sub == 2) {            // wait for the next diagonal tile, then solve and publish X
                                        if (k + 1 < nr) {
                                            if (k + 1 < nb && state[(k + 1) * nb + k + 1] < k) return;
                                            state[(k + 1) * nb + k] = k + 1;
                                        }
                                        sub = 0;
                                        ++leaf_k;
                                    }
                                }
                            };
                            bool ok = true;
                            for (size_t q = 0; q < t.size() && ok; ++q) {
                                leaf_advance();
                                const int i = t[q].x, j = t[q].y, k0 = t[q].z, type = t[q].w & 1, k1 = t[q].w >> 1;
                                if (type == 0) {
                                    if (state[j * nb + j] < j + 1 || state[i * nb + j] != j) ok = false;
                                    state[i * nb + j] = j + 1;
                                } else {
                                    if (state[i * nb + k1 - 1] < k1 || state[j * nb + k1 - 1] < k1 || state[i * nb + j] != k0) ok = false;
                                    state[i * nb + j] = k1;
                                }
                                if (!ok) printf("FORWARD DEP nb=%ld tall=%ld d=%ld w=%ld dm=%ld fill=%.1f at task %zu (%d,%d,%d,%d type %d)\n", nb, tall, depth, window, dm, fill, q, i, j, k0, k1, type);
                            }
                            leaf_advance();
                            if (ok) for (long i = 0; i < nr; ++i) for (long j = 0; j < nb && j <= i; ++j) if (j >= sc.first(i) && state[i * nb + j] != j + 1) { ok = false; printf("INCOMPLETE nb=%ld tall=%ld d=%ld w=%ld dm=%ld tile (%ld,%ld) state %d\n", nb, tall, depth, window, dm, i, j, state[i * nb + j]); break; }
                            bad += !ok;
                            }
                            }
                        }
    printf("%d bad configurations\n", bad);
    return bad != 0;
}
