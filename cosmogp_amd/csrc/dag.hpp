// Single-launch blocked Cholesky of one object, or of `nmat` same-shape objects side by side (gfx950): a persistent grid,
// one 512-thread workgroup per CU, that walks the tile DAG of the right-looking factorisation with dependency counters in
// HBM instead of kernel boundaries.  With several objects every object has its own leaf lane (workgroup b for object b),
// counters and buffers (constant strides); the workers serve ONE list in which the objects' lists are interleaved step by
// step, so the chain-bound tail of one object is filled with the bulk updates of the others.
//
//   tiles           128 x 128 (CGP_NB); row tiles i = 0 .. nr-1 (the tall rows of the fit included), column tiles
//                   ("panels") k = 0 .. nb-1
//   leaf lane       workgroup b < nmat, for object b.  For k = 0, 1, ...: factor the diagonal tile (k, k) in LDS (potrf128_lds, the K3
//                   code), solve the tile below it, X = A(k+1, k) L_kk^-T, with the factor still in LDS (the K4 code),
//                   and apply X X^T to the next diagonal tile, again in LDS: the critical chain of the factorisation never
//                   leaves this CU and never waits for a launch.
//   worker lanes    workgroups nmat .. G-1 take tasks from ONE list in ticket order (a device-scope atomic):
//                     TRSM(i, k)          X = A(i, k) L_kk^-T                         for i >= k + 2
//                     UPD(i, j, k0, k1)   A(i, j) -= sum_{k0 <= k < k1} L(i, k) L(j, k)^T
//                   Every dependency of a task sits EARLIER in the list (or on the leaf lane, whose own dependencies sit
//                   earlier still), and a workgroup only takes a ticket once it is resident: whatever the dispatch order
//                   or the number of resident workgroups, the first unfinished task always has an owner that can run, so
//                   the spins below cannot deadlock.  They are bounded anyway (wall-clock limit -> abort word -> every
//                   lane leaves), so a broken invariant shows as an error code, never as a hang.
//   state[i][j]     number of panels applied to tile (i, j); the final operation of the tile (factor / solve) makes it
//                   j + 1.  One writer at a time by construction (updates of a tile are totally ordered by k).
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility; one workgroup per CU): every byte another workgroup
// reads later is stored write-through (`sc1`, 16-byte buffer stores), every storing wave drains (`s_waitcnt vmcnt(0)`),
// the workgroup's barrier, then ONE lane stores the counter (agent-scope relaxed atomic = `sc1`); the consumer polls that
// word with `sc1` loads from one wave, its barrier, and EVERY load of handed-off bytes is an `sc1` buffer load to
// registers (L1 bypassed, so no acquire fence).  No tile is ever read by a workgroup other than its current writer before
// its final value is published, so no stale copy can exist in another XCD's L2 either.
#pragma once
#include "kernels.hpp"

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));

struct DagParams {
    double* M;              // the fit's matrix buffer (rows [0, Np): K -> L, rows [Np, R): tall rows)
    long ld;                // leading dimension = Np
    int nb, nr;             // column tiles, row tiles
    int nmat;               // objects factored side by side (leaf lanes); object b: M + b sM, wsW + b sW, ldp + b sLd, info + b,
    long sM, sW, sLd;       //   state + b nr nb
    const int4* tasks;      // {i | (object << 16), j, k0, (k1 << 1) | type}   type 0 = TRSM (panel j), 1 = UPD
    int ntasks;
    unsigned* state;        // [nmat][nr * nb]
    unsigned* ctrl;         // [0] ticket  [1] abort code  [2] first failing wait (diagnostic)  [3] progress of leaf lane 0  [4] workgroups arrived (roles)
    unsigned* sticky;       // [0] abort code  [1] failing wait  [2] ticket head then: the FIRST give-up of any launch on this handle since
                            // the host last looked (never reset by a launch: graph replays and chunked sequences reuse `ctrl`)
    double* wsW;            // [nb][8][256] T^-T blocks of the diagonal tiles (also read by later solves)
    double* ldp;            // [nb] logdet parts
    int* info;
    long long timeout;      // wall_clock64 ticks (100 MHz) a single wait may last
    unsigned long long* prof;   // diagnostics (tools/dag_probe.hip), may be null: [k][8] leaf-lane time stamps (100 MHz), then
                                // per worker [G][4] ticks spent waiting / in TRSM / in UPD, tasks run
    int dbg;                // diagnostics (tools/dag_probe.hip): 0 = run; 1..3 = the leaf lane stops after its initial load /
                            // first factorisation / first write-back and the workers leave at once; 4 = workers leave at once
};

#define DAG_THREADS 512
#ifndef DAG_UPD_KS
#define DAG_UPD_KS 20
#endif
#define DAG_LDS_DOUBLES (CGP_NB * CGP_LDA + 8 * 256 + 160)
#define DAG_LDS_BYTES (DAG_LDS_DOUBLES * 8 + 128)      // + control words

enum { DAG_ABORT_TIMEOUT = 1 };

// buffer resource over [base, base + 4 GB); the base is made wave-uniform explicitly (it always is: tile coordinates come
// from kernel arguments, loop counters and broadcast task descriptors) so that no waterfall loop is generated
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dag_rsrc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0xffffffffu, 0x00020000);
}
// 16-byte / 8-byte loads and stores that bypass this CU's L1 and write through the XCD's L2 (aux 16 = sc1)
__device__ __forceinline__ d2 ldc2(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
}
__device__ __forceinline__ void stc2(__amdgpu_buffer_rsrc_t r, unsigned off, d2 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, off, 0, 16);
}
__device__ __forceinline__ double ldc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16));
}
// every shared word is a GLOBAL (never flat) agent-scope access
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned dag_ld_flag(const unsigned* p) {
    return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dag_st_flag(unsigned* p, unsigned v) {
    __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// progress stamp of the leaf lane (diagnostics after a give-up): ctrl[3]
__device__ __forceinline__ void dag_stamp(const unsigned* ctrl, unsigned v) {
    if (threadIdx.x == 0) dag_st_flag((unsigned*)ctrl + 3, v);
}

// Every thread of the workgroup calls this.  Wave 0 polls up to three counters (lane l polls word l) until each of the first
// `nw` is >= its expected value; the result travels through an LDS word and the workgroup barrier.  Words nw .. nw + nopt - 1
// are OPTIONAL: they are read along but never waited for.  Returns 0: aborted, 1: done, 3: done and every optional word had
// reached its value as well.
__device__ __forceinline__ int dag_wait_impl(const unsigned* ctrl, long long timeout, int nw, const unsigned* w0, unsigned e0,
                                        const unsigned* w1, unsigned e1, const unsigned* w2, unsigned e2, int code, int nopt) {
    extern __shared__ __attribute__((aligned(16))) double smem_w[];
    int* okword = (int*)(smem_w + DAG_LDS_DOUBLES);
    const int tid = threadIdx.x;
    if (tid < 64) {
        const unsigned* w = tid == 0 ? w0 : (tid == 1 ? w1 : w2);
        const unsigned e = tid == 0 ? e0 : (tid == 1 ? e1 : e2);
        const bool active = tid < nw + nopt, required = tid < nw;
        int ok = 1;
        long long t0 = 0;
        unsigned v = e;
        for (unsigned spins = 0;; ++spins) {
            v = active ? dag_ld_flag(w) : e;
            if (__all((int)(!required || v >= e))) break;
            if (spins == 0) t0 = wall_clock64();
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 15) == 15) {
                unsigned ab = dag_ld_flag(ctrl + 1);
                if (ab != 0) { ok = 0; break; }
                if (wall_clock64() - t0 > timeout) {
                    if (tid == 0) {
                        unsigned exp0 = 0u, exp1 = 0u;
                        __hip_atomic_compare_exchange_strong((gu32*)(ctrl + 1), &exp0, (unsigned)DAG_ABORT_TIMEOUT, __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_compare_exchange_strong((gu32*)(ctrl + 2), &exp1, (unsigned)code, __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ok = 0;
                    break;
                }
            }
        }
        if (ok && __all((int)(v >= e))) ok = 3;
        if (tid == 0) *okword = ok;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*okword);
}

// The control steps of a lane are CALLS (a few hundred cycles per task): the compiler then neither merges the spin loops
// and counter stores of different task kinds nor sinks a counter store out of the loop that must publish it (it did both
// when they were inlined into the ticket loop, and the result hung on the GPU).
__device__ __noinline__ int dag_wait(const unsigned* ctrl, long long timeout, int nw, const unsigned* w0, unsigned e0,
                                     const unsigned* w1, unsigned e1, const unsigned* w2, unsigned e2, int code) {
    return dag_wait_impl(ctrl, timeout, nw, w0, e0, w1, e1, w2, e2, code, 0) ? 1 : 0;
}
// the same with one optional word behind the `nw` required ones: 0 aborted, 1 done, 3 done and the optional word was there too
__device__ __noinline__ int dag_wait_opt(const unsigned* ctrl, long long timeout, const unsigned* w0, unsigned e0,
                                         const unsigned* w1, unsigned e1, int code) {
    return dag_wait_impl(ctrl, timeout, 1, w0, e0, w1, e1, nullptr, 0, code, 1);
}

// publish: every wave drains its write-through stores, the barrier, then ONE lane stores the counter
__device__ __noinline__ void dag_publish(unsigned* word, unsigned value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) dag_st_flag(word, value);
    asm volatile("" ::: "memory");
}

// next ticket: one lane adds to the head word and stages the descriptor in LDS (cw[2..5]); returns the ticket, the same in
// every lane (>= ntasks: the list is exhausted)
__device__ __noinline__ int dag_fetch(unsigned* ctrl, const int4* tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) double smem_f[];
    int* cw = (int*)(smem_f + DAG_LDS_DOUBLES);
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add((gu32*)ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cw[1] = (int)t;
        if (t < (unsigned)ntasks) {
            const int4 d = tasks[t];
            cw[2] = d.x; cw[3] = d.y; cw[4] = d.z; cw[5] = d.w;
        }
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(cw[1]);
}

// ---------------------------------------------------------------------------------------------------------------
// UPD: C (128 x 128 at Cg) -= A (128 x K at Ag) B^T (128 x K at Bg); all three row-major with leading dimension ld.
// 8 waves as 2 x 4, wave tile 64 x 32, KB = 16 stages double-buffered in LDS through registers like K5 (kernels.hpp).
// Column slot (J, lp) of a wave's 32 columns stands for column 8 (J >> 1) + 2 lp + (J & 1): a lane's values for J, J + 1 are
// two adjacent doubles, so the read-modify-write of C moves 16 bytes per lane and 64 contiguous bytes per row.
// ---------------------------------------------------------------------------------------------------------------
#ifdef DAG_UPD_STAMPS
__device__ unsigned long long g_upd_acc[8];        // tools/dag_probe.hip: time of one update task by phase (100 MHz ticks), workgroup 1
#define DAG_UPD_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 1) st_[i] = wall_clock64(); } while (0)
#else
#define DAG_UPD_STAMP(i) do {} while (0)
#endif
__device__ __forceinline__ void dag_upd_tile(double* Cg, const double* Ag, const double* Bg, long ld, int K) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
#ifdef DAG_UPD_STAMPS
    long long st_[6] = {0, 0, 0, 0, 0, 0};
#endif
    DAG_UPD_STAMP(0);
    // KS = 20 doubles = 10 sixteen-byte slots per row: the four 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS) then
    // hit 16 different slots both for the A fragments (rows lr, slot 10 lr + lq) and for the B fragments (4 lp + lq); with
    // KS = 18 the A reads were 2-way conflicts (SQ_LDS_BANK_CONFLICT = 20 % of the LDS cycles of round 2's kernel)
    constexpr int TM = 128, TN = 128, WN = 4, KB = 16, KS = DAG_UPD_KS, CPR = KB / 2;
    constexpr int MT = 4, NJ = 8, PA = 2, PB = 2, RSTEP = DAG_THREADS / CPR;      // 64 rows per staging pass
    double* As = smem;                      // [2][TM * KS]
    double* Bs = smem + 2 * TM * KS;        // [2][TN * KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4, lp = lane & 3, lb = (lane >> 2) & 3;
    const int srow = tid / CPR, sch = (tid % CPR) * 2;
    const __amdgpu_buffer_rsrc_t ra_ = dag_rsrc(Ag), rb_ = dag_rsrc(Bg), rc_ = dag_rsrc(Cg);
    const unsigned ldb = (unsigned)(ld * 8);

    // The accumulators start from -C and the result is stored negated: C - sum A B^T without a second copy of the tile in
    // registers (sign changes are exact, so this is the same rounding sequence as accumulating onto +C with -A).
    // acc[i][2 jj], acc[i][2 jj + 1] <-> row 64 wm + 16 i + 4 lb + lq, columns 32 wn + 8 jj + 2 lp + (0, 1)
    const unsigned coff = (unsigned)(wm * 64 + 4 * lb + lq) * ldb + (unsigned)(wn * 32 + 2 * lp) * 8u;
    double acc[MT][NJ];
    {
        d2 cv[MT][NJ / 2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ / 2; ++jj) cv[i][jj] = ldc2(rc_, coff + (unsigned)(16 * i) * ldb + (unsigned)(8 * jj) * 8u);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ / 2; ++jj) { acc[i][2 * jj] = -cv[i][jj].x; acc[i][2 * jj + 1] = -cv[i][jj].y; }
    }

    d2 ra[2][PA], rb[2][PB];
    const int nk = K / KB;
    auto gload = [&](auto par, int kt) {
        constexpr int P = decltype(par)::value;
#if defined(DAG_ABLATE) && (DAG_ABLATE == 1 || DAG_ABLATE == 3)
        if (kt > 1) return;             // timing experiment (tools/dag_probe.hip): no global loads inside the loop
#endif
        const unsigned k1 = (unsigned)(kt * KB + sch) * 8u;
#pragma unroll
        for (int q = 0; q < PA; ++q) ra[P][q] = ldc2(ra_, (unsigned)(srow + RSTEP * q) * ldb + k1);
#pragma unroll
        for (int q = 0; q < PB; ++q) rb[P][q] = ldc2(rb_, (unsigned)(srow + RSTEP * q) * ldb + k1);
    };
    auto lstore = [&](auto par, int buf) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int q = 0; q < PA; ++q) *(d2*)(&As[buf * TM * KS + (srow + RSTEP * q) * KS + sch]) = ra[P][q];
#pragma unroll
        for (int q = 0; q < PB; ++q) *(d2*)(&Bs[buf * TN * KS + (srow + RSTEP * q) * KS + sch]) = rb[P][q];
    };
    // fragments of half a stage (8 of its 16 k): two register sets, the reads of one half are issued among the MFMAs of the
    // half before it
    d2 fa[2][MT], fb[2][NJ];
#if defined(DAG_ABLATE) && DAG_ABLATE >= 2
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[q][i] = (d2){(double)(lane + i), (double)(q + 1)};
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[q][j] = (d2){(double)(lane - j), (double)(q + 2)};
    }
#endif
    auto fread = [&](auto setc, int buf, int t) {
        constexpr int S = decltype(setc)::value;
#if defined(DAG_ABLATE) && DAG_ABLATE >= 2
        if (buf + t >= 0) return;       // timing experiment (tools/dag_probe.hip): no fragment reads
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i)
            fa[S][i] = *(const d2*)(&As[buf * TM * KS + (wm * 64 + i * 16 + lr) * KS + 8 * t + 2 * lq]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            fb[S][j] = *(const d2*)(&Bs[buf * TN * KS + (wn * 32 + 8 * (j >> 1) + 2 * lp + (j & 1)) * KS + 8 * t + 2 * lq]);
    };
    auto fmma = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[S][i].x, fb[S][j].x, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma4(fa[S][i].y, fb[S][j].y, acc[i][j]);
    };
    // issue order asked of the scheduler for {12 LDS reads, 64 MFMAs}: one read every five MFMAs
    auto interleave_reads = [&]() {
#pragma unroll
        for (int r = 0; r < MT + NJ; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    gload(P0{}, 0);
    gload(P1{}, 1);
    __builtin_amdgcn_sched_barrier(0);
    lstore(P0{}, 0);
    __syncthreads();
    DAG_UPD_STAMP(1);
    fread(P0{}, 0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        // stage in buffer 0; stage kt + 1 goes to buffer 1 meanwhile
        gload(P0{}, min(kt + 2, nk - 1));
        fread(P1{}, 0, 1);
        fmma(P0{});
        lstore(P1{}, 1);
        __builtin_amdgcn_sched_group_barrier(0x020, PA + PB, 0);
        interleave_reads();
        __builtin_amdgcn_sched_group_barrier(0x200, PA + PB, 0);
        __syncthreads();                 // buffer 1 complete; every wave's reads of buffer 0 have returned
        fread(P0{}, 1, 0);
        fmma(P1{});
        // stage in buffer 1; stage kt + 2 goes to buffer 0 meanwhile
        gload(P1{}, min(kt + 3, nk - 1));
        fread(P1{}, 1, 1);
        fmma(P0{});
        lstore(P0{}, 0);
        interleave_reads();
        __builtin_amdgcn_sched_group_barrier(0x020, PA + PB, 0);
        interleave_reads();
        __builtin_amdgcn_sched_group_barrier(0x200, PA + PB, 0);
        __syncthreads();
        fread(P0{}, 0, 0);
        fmma(P1{});
        interleave_reads();
    }
    DAG_UPD_STAMP(2);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ / 2; ++jj) {
            d2 o;
            o.x = -acc[i][2 * jj];
            o.y = -acc[i][2 * jj + 1];
            stc2(rc_, coff + (unsigned)(16 * i) * ldb + (unsigned)(8 * jj) * 8u, o);
        }
#ifdef DAG_UPD_STAMPS
    DAG_UPD_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DAG_UPD_STAMP(4);
    if (threadIdx.x == 0 && blockIdx.x == 1) {
        g_upd_acc[0] += st_[1] - st_[0]; g_upd_acc[1] += st_[2] - st_[1]; g_upd_acc[2] += st_[3] - st_[2]; g_upd_acc[3] += st_[4] - st_[3];
        g_upd_acc[4] += 1;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The K4 solve of 128 rows (8 waves x 16 rows) against a diagonal block whose factor sits in LDS: Ls (strictly lower
// 16 x 16 tiles, row stride CGP_LDA) and Ws (the eight T^-T blocks).  Rows at Ap-band `rp` (buffer resource of the tile's
// first row and column), X is stored back write-through and returned in S (K4's register layout: S[jb][0..3] =
// columns 16 jb + 4 lu + (0..3) of row 16 wave + 4 lb + lq).
// ---------------------------------------------------------------------------------------------------------------
// S (already loaded: S[jb][0..3] = columns 16 jb + 4 lu + (0..3) of row 16 wave + 4 lb + lq, at byte offset aoff of rp) holds the
// not yet solved column tiles of this wave's rows; a finished X_jb replaces S[jb] in place and is stored write-through.
__device__ __forceinline__ void dag_trsm_rows_regs(double* Ls, double* Ws, __amdgpu_buffer_rsrc_t rp, unsigned aoff, double (&S)[8][4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    // wave-private 16 x 16 tile in the part of Ls that holds no strictly-lower tile (see k_trsm_panel)
    const int xg = wave < 6 ? 0 : 1, xt = wave - 6 * xg;
    double* Xs = Ls + xg * 16 * CGP_LDA + 16 + 18 * xt;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        wave_lds_fence();
        *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu) = (d2){S[jb][0], S[jb][1]};
        *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu + 2) = (d2){S[jb][2], S[jb][3]};
        wave_lds_fence();
        double x[4] = {0., 0., 0., 0.};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d2 a = *(const d2*)(Xs + lr * CGP_LDA + 8 * t + 2 * lq);
            const double* w0 = Ws + jb * 256 + (8 * t + 2 * lq) * 16 + 4 * lu;
            d2 p0 = *(const d2*)(w0), p1 = *(const d2*)(w0 + 2), p2 = *(const d2*)(w0 + 16), p3 = *(const d2*)(w0 + 18);
            x[0] = mfma4(a.x, p0.x, x[0]); x[1] = mfma4(a.x, p0.y, x[1]);
            x[2] = mfma4(a.x, p1.x, x[2]); x[3] = mfma4(a.x, p1.y, x[3]);
            x[0] = mfma4(a.y, p2.x, x[0]); x[1] = mfma4(a.y, p2.y, x[1]);
            x[2] = mfma4(a.y, p3.x, x[2]); x[3] = mfma4(a.y, p3.y, x[3]);
        }
        stc2(rp, aoff + (unsigned)(jb * 16) * 8u, (d2){x[0], x[1]});
        stc2(rp, aoff + (unsigned)(jb * 16 + 2) * 8u, (d2){x[2], x[3]});
#pragma unroll
        for (int J = 0; J < 4; ++J) S[jb][J] = x[J];
        if (jb < 7) {
            wave_lds_fence();
            *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu) = (d2){x[0], x[1]};
            *(d2*)(Xs + (4 * lb + lq) * CGP_LDA + 4 * lu + 2) = (d2){x[2], x[3]};
            wave_lds_fence();
            d2 a[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = -*(const d2*)(Xs + lr * CGP_LDA + 8 * t + 2 * lq);
#pragma unroll
            for (int i = jb + 1; i < 8; ++i) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    d2 bb[4];
#pragma unroll
                    for (int J = 0; J < 4; ++J)
                        bb[J] = *(const d2*)(Ls + (i * 16 + 4 * lu + J) * CGP_LDA + jb * 16 + 8 * t + 2 * lq);
#pragma unroll
                    for (int J = 0; J < 4; ++J) S[i][J] = mfma4(a[t].x, bb[J].x, S[i][J]);
#pragma unroll
                    for (int J = 0; J < 4; ++J) S[i][J] = mfma4(a[t].y, bb[J].y, S[i][J]);
                }
            }
        }
    }
}

__device__ __forceinline__ void dag_trsm_rows(double* Ls, double* Ws, __amdgpu_buffer_rsrc_t rp, unsigned ldb, double (&S)[8][4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    const unsigned aoff = (unsigned)(wave * 16 + 4 * lb + lq) * ldb + (unsigned)(4 * lu) * 8u;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        d2 t0 = ldc2(rp, aoff + (unsigned)(jb * 16) * 8u), t1 = ldc2(rp, aoff + (unsigned)(jb * 16 + 2) * 8u);
        S[jb][0] = t0.x; S[jb][1] = t0.y; S[jb][2] = t1.x; S[jb][3] = t1.y;
    }
    dag_trsm_rows_regs(Ls, Ws, rp, aoff, S);
}

// worker TRSM(i, k): stage L_kk (rows 16..127 hold strictly-lower tiles) and its T^-T blocks, then solve the tile
__device__ __noinline__ void dag_trsm_task(double* M, long ld, const double* wsW, int i, int k) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    struct { double* M; long ld; const double* wsW; } p = {M, ld, wsW};
    double* Ls = smem;
    double* Ws = smem + CGP_NB * CGP_LDA;
    const int tid = threadIdx.x;
    const unsigned ldb = (unsigned)(p.ld * 8);
    const __amdgpu_buffer_rsrc_t rl = dag_rsrc(p.M + (long)(k * CGP_NB) * p.ld + (long)k * CGP_NB);
    const __amdgpu_buffer_rsrc_t rw = dag_rsrc(p.wsW + (long)k * (8 * 256));
    // rows 16..127, columns 0..(row tile start): 112 rows x 64 chunks of 16 B, only chunks left of the diagonal tile
    constexpr int NCH = 112 * 64 / DAG_THREADS;          // 14
    d2 v[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int idx = tid + q * DAG_THREADS, r = 16 + (idx >> 6), c = (idx & 63) * 2;
        v[q] = (c >> 4) < (r >> 4) ? ldc2(rl, (unsigned)r * ldb + (unsigned)c * 8u) : (d2){0., 0.};
    }
    d2 w[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) w[q] = ldc2(rw, (unsigned)(tid + q * DAG_THREADS) * 16u);
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int idx = tid + q * DAG_THREADS, r = 16 + (idx >> 6), c = (idx & 63) * 2;
        if ((c >> 4) < (r >> 4)) *(d2*)(Ls + r * CGP_LDA + c) = v[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *(d2*)(Ws + 2 * (tid + q * DAG_THREADS)) = w[q];
    __syncthreads();
    double xs[8][4];
    dag_trsm_rows(Ls, Ws, dag_rsrc(p.M + (long)(i * CGP_NB) * p.ld + (long)k * CGP_NB), ldb, xs);
}

// ---------------------------------------------------------------------------------------------------------------
// leaf lane
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dag_tile_of(int t, int& ti, int& tj) {      // t-th lower tile of an 8 x 8 tile grid, row-major
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= t) ++i;
    ti = i;
    tj = t - i * (i + 1) / 2;
}

// the K3 factorisation as a CALL: its register allocation (183 VGPRs, a long unrolled pivot chain) stays its own instead of
// being merged with the solve / update code around it.  The dynamic LDS is addressed through its own symbol, so the
// accesses stay ds_* instructions.
__device__ __noinline__ int dag_potrf_call() {
    extern __shared__ __attribute__((aligned(16))) double smem_[];
    return potrf128_lds(smem_, smem_ + CGP_NB * CGP_LDA, smem_ + CGP_NB * CGP_LDA + 8 * 256);
}

// Steps of the leaf lane as CALLS as well: nothing but the step counter lives in registers across them (the diagonal
// tile and the factor travel through LDS), so the unrolled pivot chain of the factorisation and the solve / update code
// do not compete for registers.
//
// dag_leaf_step(k), called with the factor of (k, k) in LDS:
//   1. write L_kk and its T^-T blocks back (write-through), and -- once the workers have brought the tile below and the
//      next diagonal tile up to date, ONE wait for both words -- fetch the rows of the tile below (S) and this wave's share
//      of the next diagonal tile (36 lower 16 x 16 tiles dealt round robin over the 8 waves) while those stores drain;
//   2. publish the factor; solve X = A(k+1, k) L_kk^-T with the factor still in LDS (the K4 code);
//   3. X into LDS (over the factor), publish X; next diagonal tile -= X X^T, the waves' tiles interleaved so that the LDS
//      fragment reads of one hide behind the MFMAs of the others; the result replaces X in LDS.
// Returns 0 = aborted, 1 = done and the next diagonal tile is in LDS, 2 = done, no further diagonal tile.
__device__ __noinline__ int dag_leaf_step(double* M, long ld, int nb_, int nr_, unsigned* state, const unsigned* ctrl, long long timeout,
                                          double* wsW, double* ldp, int* info, int k, int bad, unsigned long long* prof) {
    extern __shared__ __attribute__((aligned(16))) double smem_[];
    struct { double* M; long ld; int nb; unsigned* state; } p = {M, ld, nb_, state};
    double* As = smem_;
    double* Ws = smem_ + CGP_NB * CGP_LDA;
    double* Lc = Ws + 8 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4, lu = lane & 3, lb = (lane >> 2) & 3;
    const unsigned ldb = (unsigned)(p.ld * 8);
    const int nb = p.nb;
    const bool below = k + 1 < nr_, diag = k + 1 < nb;

    // ---- 1. write-back of the factor
    {
        const __amdgpu_buffer_rsrc_t rd = dag_rsrc(p.M + (long)(k * CGP_NB) * p.ld + (long)k * CGP_NB);
        constexpr int NCH = CGP_NB * 64 / DAG_THREADS;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int idx = tid + q * DAG_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            if ((c >> 4) <= (r >> 4)) stc2(rd, (unsigned)r * ldb + (unsigned)c * 8u, *(const d2*)(As + r * CGP_LDA + c));
        }
        const __amdgpu_buffer_rsrc_t rw = dag_rsrc(wsW + (long)k * (8 * 256));
#pragma unroll
        for (int q = 0; q < 2; ++q) stc2(rw, (unsigned)(tid + q * DAG_THREADS) * 16u, *(const d2*)(Ws + 2 * (tid + q * DAG_THREADS)));
        double lg = 0.;
        if (tid < CGP_NB) lg = 2. * log(As[tid * CGP_LDA + tid]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
        double* red = Lc + 128;
        if (lane == 0 && wave < 2) red[wave] = lg;
    }
    // The tile below with panels 0..k-1 applied.  ORDER MATTERS for the no-deadlock argument of the ticket list: the factor is
    // published after THIS wait only (its update precedes every task that needs the factor in the list); the next diagonal
    // tile is waited for after the factor is out, because its last update is listed behind the solves of panel k.
    int diag_ready = 0;              // the next diagonal tile was already up to date when the tile below was: no second wait
    if (below) {
        const int r = diag ? __builtin_amdgcn_readfirstlane(dag_wait_opt(ctrl, timeout, p.state + (long)(k + 1) * nb + k, (unsigned)k,
                                                                         p.state + (long)(k + 1) * nb + (k + 1), (unsigned)k, 0x10000 + k))
                           : __builtin_amdgcn_readfirstlane(dag_wait(ctrl, timeout, 1, p.state + (long)(k + 1) * nb + k, (unsigned)k, nullptr,
                                                                     0, nullptr, 0, 0x10000 + k));
        if (!r) return 0;
        diag_ready = r == 3;
    }
    if (prof && tid == 0) prof[0] = wall_clock64();
    // this wave's share of the next diagonal tile: its lower triangle as 20 blocks of 32 rows x 16 columns (block row R =
    // rows 32 R .. 32 R + 31, column tiles c <= 2 R + 1), dealt round robin over the 8 waves (3 / 3 / 3 / 3 / 2 / 2 / 2 / 2:
    // five per SIMD); two row fragments share the four column fragments of a block, which keeps the X X^T update below
    // the LDS read rate (as 36 separate 16 x 16 tiles it was LDS-bound)
    constexpr int NT = 3;
    double S[8][4], dacc[NT][2][4];
    int bR[NT], bC[NT];
    const unsigned aoff = (unsigned)(wave * 16 + 4 * lb + lq) * ldb + (unsigned)(4 * lu) * 8u;
    const __amdgpu_buffer_rsrc_t rp = dag_rsrc(p.M + (long)((k + 1) * CGP_NB) * p.ld + (long)k * CGP_NB);
    if (below) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            d2 t0 = ldc2(rp, aoff + (unsigned)(jb * 16) * 8u), t1 = ldc2(rp, aoff + (unsigned)(jb * 16 + 2) * 8u);
            S[jb][0] = t0.x; S[jb][1] = t0.y; S[jb][2] = t1.x; S[jb][3] = t1.y;
        }
    }
    // ---- 2. publish the factor (the stores above have drained once the loads behind them have returned)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const double* red = Lc + 128;
        ldp[k] = red[0] + red[1];
        if (bad != 0 && info[0] == 0) info[0] = k * CGP_NB + bad;
        dag_st_flag(p.state + (long)k * nb + k, (unsigned)(k + 1));
    }
    if (prof && tid == 0) prof[1] = wall_clock64();
    if (!below) return 2;
    // this wave's share of the next diagonal tile (workers have applied panels 0..k-1): the loads land during the solve
    if (diag) {
        if (!diag_ready &&
            !__builtin_amdgcn_readfirstlane(dag_wait(ctrl, timeout, 1, p.state + (long)(k + 1) * nb + (k + 1), (unsigned)k, nullptr, 0,
                                                     nullptr, 0, 0x20000 + k)))
            return 0;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            // block b = wave + 8 n of the list R = 3 (8 blocks), 2 (6), 1 (4), 0 (2)
            const int b = min(wave + 8 * n, 19);
            const int R = b < 8 ? 3 : (b < 14 ? 2 : (b < 18 ? 1 : 0));
            const int c = b - (R == 3 ? 0 : (R == 2 ? 8 : (R == 1 ? 14 : 18)));
            bR[n] = __builtin_amdgcn_readfirstlane(R);
            bC[n] = __builtin_amdgcn_readfirstlane(c);
        }
        const __amdgpu_buffer_rsrc_t rn = dag_rsrc(p.M + (long)((k + 1) * CGP_NB) * p.ld + (long)(k + 1) * CGP_NB);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (wave + 8 * n < 20) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int J = 0; J < 4; ++J)
                        dacc[n][r][J] = ldc1(rn, (unsigned)(bR[n] * 32 + r * 16 + 4 * lb + lq) * ldb + (unsigned)(bC[n] * 16 + lu + 4 * J) * 8u);
            }
        }
    }
    dag_trsm_rows_regs(As, Ws, rp, aoff, S);
    // ---- 3. X over the factor in LDS, publish X
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                     // every wave is done with L_kk, every X store has drained
    if (tid == 0) dag_st_flag(p.state + (long)(k + 1) * nb + k, (unsigned)(k + 1));
    if (prof && tid == 0) prof[2] = wall_clock64();
    if (!diag) return 2;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        double* dst = As + (wave * 16 + 4 * lb + lq) * CGP_LDA + jb * 16 + 4 * lu;
        *(d2*)(dst) = (d2){S[jb][0], S[jb][1]};
        *(d2*)(dst + 2) = (d2){S[jb][2], S[jb][3]};
    }
    __syncthreads();
    // tile (k+1, k+1) -= X X^T : k outer, this wave's blocks inner
#pragma unroll 1
    for (int kc = 0; kc < 16; ++kc) {
        d2 fa[NT][2], fb[NT][4];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const double* Arow = As + (bR[n] * 32) * CGP_LDA + 8 * kc;
            const double* Brow = As + (bC[n] * 16) * CGP_LDA + 8 * kc;
#pragma unroll
            for (int r = 0; r < 2; ++r) fa[n][r] = -*(const d2*)(Arow + (r * 16 + lr) * CGP_LDA + 2 * lq);
#pragma unroll
            for (int J = 0; J < 4; ++J) fb[n][J] = *(const d2*)(Brow + (4 * J + lu) * CGP_LDA + 2 * lq);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
            if (n < 2 || wave < 4) {             // waves 4..7 own two blocks (wave-uniform)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int J = 0; J < 4; ++J) dacc[n][r][J] = mfma4(fa[n][r].x, fb[n][J].x, dacc[n][r][J]);
            }
#pragma unroll
        for (int n = 0; n < NT; ++n)
            if (n < 2 || wave < 4) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int J = 0; J < 4; ++J) dacc[n][r][J] = mfma4(fa[n][r].y, fb[n][J].y, dacc[n][r][J]);
            }
    }
    __syncthreads();                 // every wave has read X
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        if (wave + 8 * n < 20) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int J = 0; J < 4; ++J)
                    As[(bR[n] * 32 + r * 16 + 4 * lb + lq) * CGP_LDA + bC[n] * 16 + lu + 4 * J] = dacc[n][r][J];
        }
    }
    __syncthreads();
    if (prof && tid == 0) prof[3] = wall_clock64();
    return 1;
}

__device__ void dag_leaf(const DagParams& p, double* smem, int b) {
    double* As = smem;
    const int tid = threadIdx.x;
    const unsigned ldb = (unsigned)(p.ld * 8);
    double* const M = p.M + (long)b * p.sM;
    unsigned* const state = p.state + (long)b * p.nr * p.nb;
    double* const wsW = p.wsW + (long)b * p.sW;
    double* const ldp = p.ldp + (long)b * p.sLd;
    int* const info = p.info + b;
    unsigned long long* const prof = b == 0 ? p.prof : nullptr;
    {
        // tile (0, 0) as assembled, straight into LDS (tiles on and below the block diagonal)
        const __amdgpu_buffer_rsrc_t rd = dag_rsrc(M);
        constexpr int NCH = CGP_NB * 64 / DAG_THREADS;      // 16
        d2 v[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int idx = tid + q * DAG_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            v[q] = (c >> 4) <= (r >> 4) ? ldc2(rd, (unsigned)r * ldb + (unsigned)c * 8u) : (d2){0., 0.};
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int idx = tid + q * DAG_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            if ((c >> 4) <= (r >> 4)) *(d2*)(As + r * CGP_LDA + c) = v[q];
        }
        __syncthreads();
    }
    for (int k = 0; k < p.nb; ++k) {
        if (b == 0) dag_stamp(p.ctrl, 4u * k + 1u);
        if (p.dbg == 1) return;
        if (prof && tid == 0) prof[k * 8 + 0] = wall_clock64();
        const int bad = dag_potrf_call();
        if (prof && tid == 0) prof[k * 8 + 1] = wall_clock64();
        if (b == 0) dag_stamp(p.ctrl, 4u * k + 2u);
        if (p.dbg == 2) return;
        if (dag_leaf_step(M, p.ld, p.nb, p.nr, state, p.ctrl, p.timeout, wsW, ldp, info, k, bad,
                          prof ? prof + k * 8 + 2 : nullptr) != 1) break;
    }
    if (b == 0) dag_stamp(p.ctrl, 0xffffu);
}

// A lane that leaves after a give-up (its own or another lane's) copies the launch's abort words into the handle's sticky
// words, first give-up wins: `ctrl` is cleared by the next launch (graph replays, chunked sequences), the sticky words only
// by the host once it has reported them (dag_check).
__device__ __forceinline__ void dag_note_abort(const DagParams& p) {
    if (threadIdx.x != 0 || !p.sticky) return;
    const unsigned ab = dag_ld_flag(p.ctrl + 1);
    if (ab == 0) return;
    unsigned exp0 = 0u;
    if (__hip_atomic_compare_exchange_strong((gu32*)p.sticky, &exp0, ab, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        dag_st_flag(p.sticky + 1, dag_ld_flag(p.ctrl + 2));
        dag_st_flag(p.sticky + 2, dag_ld_flag(p.ctrl));
        dag_st_flag(p.sticky + 3, dag_ld_flag(p.ctrl + 3));
    }
}

// this workgroup's arrival number in the launch (ctrl[4]; the launch's control block starts zeroed)
__device__ __noinline__ int dag_role(unsigned* ctrl) {
    extern __shared__ __attribute__((aligned(16))) double smem_r[];
    int* cw = (int*)(smem_r + DAG_LDS_DOUBLES);
    if (threadIdx.x == 0) cw[6] = (int)__hip_atomic_fetch_add((gu32*)(ctrl + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return cw[6];
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DAG_THREADS) void k_chol_dag(DagParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int* cw = (int*)(smem + DAG_LDS_DOUBLES);       // [0] wait result, [1..5] ticket + descriptor
    // Roles by ARRIVAL, not by block id (round 4): the first nmat workgroups to become resident are the leaf lanes.  With roles
    // by block id a launch could end up with workers resident and a leaf lane not -- the dispatcher deals blocks to the XCDs
    // independently, block 0 can queue for a CU of its XCD while later blocks start elsewhere -- and when several persistent
    // launches compete for the CUs (four host threads, tests/test_dag.py) two of them could hold each other's CUs with
    // workers that spin for leaf lanes that never start: the bounded spins then gave up after 4 s.  By arrival, every
    // resident worker belongs to a launch whose leaf lanes are all resident, so it can always make progress.
    const int role = __builtin_amdgcn_readfirstlane(dag_role(p.ctrl));
    if (role < p.nmat) {
        if (p.dbg != 9) dag_leaf(p, smem, role);   // 9: update-task benchmark of tools/dag_probe.hip, no leaf lane
        dag_note_abort(p);
        return;
    }
    const int nb = p.nb;
    const long sS = (long)p.nr * nb;
    if (p.dbg != 0 && p.dbg < 5) return;
    long long acc_wait = 0, acc_trsm = 0, acc_upd = 0, acc_n = 0, tm0 = 0, tm1 = 0, acc_wt = 0, acc_w1 = 0;
    for (;;) {
        if (p.prof) tm0 = wall_clock64();
        const int t = __builtin_amdgcn_readfirstlane(dag_fetch(p.ctrl, p.tasks, p.ntasks));
        if (t >= p.ntasks) break;
        const int ib = __builtin_amdgcn_readfirstlane(cw[2]), j = __builtin_amdgcn_readfirstlane(cw[3]);
        const int k0 = __builtin_amdgcn_readfirstlane(cw[4]), w = __builtin_amdgcn_readfirstlane(cw[5]);
        const int type = w & 1, k1 = w >> 1, i = ib & 0xffff, ob = ib >> 16;
        unsigned* const state = p.state + (long)ob * sS;
        double* const M = p.M + (long)ob * p.sM;
        if (type == 0) {
            // TRSM(i, k = j): needs the factor of (k, k) and every update of (i, k)
            if (!__builtin_amdgcn_readfirstlane(dag_wait(p.ctrl, p.timeout, 2, state + (long)j * nb + j, (unsigned)(j + 1),
                                                         state + (long)i * nb + j, (unsigned)j, nullptr, 0, 0x30000 + t))) break;
            if (p.prof) tm1 = wall_clock64();
            if (p.dbg != 5 && p.dbg != 7) dag_trsm_task(M, p.ld, p.wsW + (long)ob * p.sW, i, j);
            dag_publish(state + (long)i * nb + j, (unsigned)(j + 1));
            if (p.prof) { acc_wait += tm1 - tm0; acc_wt += tm1 - tm0; acc_trsm += wall_clock64() - tm1; ++acc_n; }
        } else {
            // UPD(i, j, k0, k1): panels k0..k1-1 of rows i and j final, tile (i, j) updated through k0-1
            if (!__builtin_amdgcn_readfirstlane(dag_wait(p.ctrl, p.timeout, 3, state + (long)i * nb + (k1 - 1), (unsigned)k1,
                                                         state + (long)j * nb + (k1 - 1), (unsigned)k1, state + (long)i * nb + j,
                                                         (unsigned)k0, 0x40000 + t))) break;
            if (p.prof) tm1 = wall_clock64();
            if (p.dbg != 5 && p.dbg != 6)
                dag_upd_tile(M + (long)(i * CGP_NB) * p.ld + (long)j * CGP_NB, M + (long)(i * CGP_NB) * p.ld + (long)k0 * CGP_NB,
                             M + (long)(j * CGP_NB) * p.ld + (long)k0 * CGP_NB, p.ld, (k1 - k0) * CGP_NB);
            dag_publish(state + (long)i * nb + j, (unsigned)k1);
            if (p.prof) { acc_wait += tm1 - tm0; if (k1 - k0 == 1) acc_w1 += tm1 - tm0; acc_upd += wall_clock64() - tm1; ++acc_n; }
        }
    }
    if (p.prof && threadIdx.x == 0) {
        unsigned long long* w = p.prof + (long)p.nb * 8 + (long)blockIdx.x * 4;
        w[0] = acc_wait; w[1] = acc_trsm; w[2] = acc_upd; w[3] = acc_n | ((unsigned long long)(acc_wt / 16) << 20) | ((unsigned long long)(acc_w1 / 16) << 42);
    }
    if (p.dbg >= 5 && threadIdx.x == 0) dag_st_flag(p.ctrl + 5, 0x4000u);
    dag_note_abort(p);
}
