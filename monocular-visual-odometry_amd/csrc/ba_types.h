// csrc/ba_types.h -- data shared by the bundle-adjustment kernel (ba_kernels.hip) and its host side (ba_host.cpp):
// the device-resident window descriptor, the LDS carve-up and the summation plan.
//
// Canonical arithmetic: every cross-edge / cross-landmark sum of the solver has ONE defined order, so
// that a run is bit-reproducible and can be restated bit for bit by a scalar CPU program (the parity tests do that):
//   * the window is cut into G contiguous landmark ranges ("workgroup ranges", balanced by edge count);
//   * inside a range a Gram-type sum (pose blocks M^T M, Schur blocks U^T U) is ONE chain of fused multiply-adds over
//     its rows / columns in storage order (this is what a sequence of v_mfma_f64_16x16x4_f64 computes: the matrix
//     core adds the four k-slots of an instruction as IEEE FMAs in k order), optionally split into `nsplit`
//     consecutive column ranges whose chain results are added in range order;
//   * range results are added in range order g = 0 .. G-1;
//   * scalar sums (chi2, predicted decrease) are per-thread partials (thread = edge / landmark index mod 512),
//     a 64-lane xor butterfly (32, 16, .. 1), the 8 waves in order, the ranges in order;
//   * the reduced system is eliminated in the order Eigen::LDLT takes its rows (largest |diagonal entry| of the input first,
//     first maximum on ties: g2o's LinearSolverDense), right-looking, r = 1 / d, l = c r, fma updates;
//   * the predicted decrease of the solver's STALE step (scored when a solve fails): landmark part per range with lane =
//     landmark mod 64 and one butterfly, added over the ranges like a Schur entry; pose part as one butterfly; then + 1e-3.
#ifndef MVO_BA_TYPES_H
#define MVO_BA_TYPES_H
#include <stddef.h>
#include <stdint.h>

typedef unsigned long long ba_u64;

#define BA_THREADS 512
#define BA_WAVES 8
#define BA_MAX_POSES 20
#define BA_MSTRIDE 15               // doubles per edge in M: two rows [A~ (6) | e~] at +0 and +7, one pad (odd pitch)
#define BA_XS 7                     // doubles per landmark in H_ll / C (6 used): odd pitch, so that one-lane-per-landmark
//                                  accesses spread over the LDS banks
#define BA_SXS 9                    // doubles per edge in the staged [X~ (6) | e~ (2)] rows of the landmark-block phase (odd)
#define BA_EDGE_SLOTS 2             // edges per thread: the first one keeps its Jacobian rows in registers, the second one (ranges
//                                  with more than 512 edges) in LDS; a range holds <= 1024 edges
#define BA_PANEL_DOUBLES 1040        // block LDL^T: rows [-l] and [c] of a 4-column panel, 64 rows each, two buffers; the 4 x 4 diagonal block
#define BA_E2S 21                   // doubles per edge in that LDS area: a0 (6) | a1 (6) | x (6) | e~ (2), odd pitch
#define BA_LDS_BUDGET (157 * 1024)  // dynamic part; the static part (descriptor, flags: < 1.5 KB) comes on top (160 KB per CU)
#define BA_MAX_WGS 256
#define BA_MAX_BATCH 16             // windows per launch (8 x 32 or 16 x 16 workgroups = one workgroup per CU)
#define BA_NPHASE 16
#define BA_TRACE_MAX 512            // LM trials recorded per solve (50 iterations x at most 10 trials)
#define BA_HP_PASSES 6               // staging passes of the pose-block rows whose slice tables are kept for the whole solve
#define BA_HP 28                    // packed lower triangle of the 7 x 7 pose block [H_pp | -b_p; . | e^T e]

struct BaStatsDev {
    int iterations, trials, terminated, error;
    int failed_solves;  // trials whose factorisation was "not positive" (g2o: the linear solve fails)
    int stale_steps;    // of those: the solver's stale x was applied and ACCEPTED (negative predicted decrease, see ba_window)
    double chi2_initial, chi2_final, lambda_final;
    long long phase[BA_NPHASE];  // shader-clock cycles per phase as seen by thread 0 of workgroup 0
    long long solve_ticks;       // duration of the solve on the 100 MHz clock (workgroup 0, load to write-back)
};
// phase ids: 0 LIN, 1 pose-block chains, 2 landmark blocks + pose-block exchange, 3 T1 + U, 4 Schur chains,
// 5 publish + slice reduction + gather, 6 assemble, 7 reduced solve, 8 back-substitution + update, 9 chi2,
// 10 chi2 exchange + decision, 11 whole kernel

// One LM trial as recorded for the parity tests: the damping it was solved with, the robust chi2 it reached, the gain
// ratio and whether the step was accepted.
struct BaTraceRow {
    double lambda, chi2, rho, accepted;
};

struct BaDev {
    int F, L, E, G, nfree, n, NT, npair, fix_points, max_it, maxEg, maxLg;
    int max_dup;  // largest rank of an observation among those of its (landmark, pose): 0 = no duplicate observations
    int nlow;    // packed entries of the reduced system: lower triangle (i >= j, i < n) then the rhs row (n, j)
    int npk;     // nlow rounded up to 16: row pitch of the partial exchange
    int slice;   // packed entries every workgroup reduces in stage 1
    int groups;  // K: the Schur exchange sums the partials of the workgroups g = k mod K first (one XCD each), then the K group sums
    int nsplit;  // consecutive column pieces of a Schur chain (results added in piece order) = nseq x npar
    int npar;    // pieces that run side by side on different waves (one chunk of the U area holds npar pieces)
    int nseq;    // chunks of the U area that are built and consumed one after the other
    int uarea;   // doubles of the multi-purpose U / staging area of a workgroup
    int npt;     // passes of the landmark-block phase: pass h stages [X~ | e~] of the edges of the h-th slice of a range's landmarks
    int panel;   // 1: the solver area has the panel of the block LDL^T behind it
    int e2_edges;  // edges per range whose Jacobian rows live in LDS (all of them, or those behind the first 512)
    int alias_sl;   // 1: the reduced system (SL) lives in the U area (ba_solver_doubles)
    int uv_global;  // 1: the measurements (u, v) of a range stay in device memory (read once per trial) instead of LDS
    double* uv_dev;  // their device copy (E x 2; a window of the resident service has its inputs in pinned host memory)
    double* dxl_dev;  // L x 3 in device memory: the landmark part of the solver's x (the last successful solve's step)
    int slots;   // kernel flavour the plan was made for: 0 = all rows in LDS, 1 / 2 = first 512 edges in registers
    int ldu;     // rows of the U buffer = 16 NT
    int nhp;     // pose-block exchange entries per workgroup = BA_HP nfree + 1 (last: max |diag H_ll|)
    double f, cx, cy, delta;
    double lc00, lc01, lc11;  // upper Cholesky factor of the information matrix: Omega = Lc^T Lc
    const double* poses_in;   // F x 16
    double* poses_out;        // F x 16
    const double* pts_in;     // L x 3
    double* pts_out;          // L x 3
    const int* wg_pt_start;   // G + 1   (landmark ranges)
    const int* wg_edge_start; // G + 1   (edges sorted by owner workgroup, then pose)
    const int* wg_pose_start; // G x (F + 1): absolute edge index where pose p starts inside workgroup g
    const int* e_pose;        // E
    const int* e_point;       // E (global landmark index)
    const double* e_uv;       // E x 2
    const int* pt_edge_start; // L + 1 -> pt_edge_list
    const int* pt_edge_list;  // E absolute edge indices, grouped by landmark
    const short* eof;         // L x nfree: LOCAL index of the first edge (landmark, pose slot), -1 if none
    const short* dup_rank;    // E: rank of the edge among the observations of its (landmark, pose), ascending edge order
    const int* pose_slot;     // F
    const int* slot_pose;     // nfree
    const short* pk_of_tile;  // npair x 256: packed index of tile entry (pair, r, c), -1 if not needed
    // cross-workgroup exchange: 8-byte granules {tag : 32 | half a double : 32}, two per value
    ba_u64* xP;  // G x npk x 2   Schur partials
    ba_u64* xR;  // npk x 2       the same entries summed over the workgroups
    ba_u64* xH;  // G x nhp x 2   pose-block partials (+ max diagonal)
    ba_u64* xC;  // 2 (parity) x G x 2 x 2   chi2 / predicted-decrease partials
    BaStatsDev* stats;
    // pinned host mirrors written by the kernel itself at the end of the solve (fetching needs no copy dispatch)
    BaStatsDev* h_stats;
    double* h_poses;  // F x 16
    double* h_pts;    // L x 3 (null when the landmarks are fixed)
    BaTraceRow* trace;  // BA_TRACE_MAX rows in pinned host memory, or null
};

struct BaBatch {  // kernel argument: the windows of one launch
    const BaDev* win[BA_MAX_BATCH];
    unsigned tag_base[BA_MAX_BATCH];  // launch sequence number of the window << 12: exchange tags never repeat
    int nwin;
    int stride;      // blockIdx -> (window = b % stride, workgroup = b / stride); 8 when every window has <= 32 workgroups
    int use_mfma;    // 0: validation path (the same fma chains on the vector ALU)
    int same_l2_ok;  // 0: always publish write-through (test hook); 1: plain stores when a window sits on one XCD
};

// ---- resident solver service (k_ba_service): one mailbox per slot in pinned host memory, written by the host's scheduler
// (fields first, `seq` last) and polled by workgroup 0 of the slot; `done_seq` is written by the device when the job's
// results are in the window's pinned mirrors.
#define BA_SERVICE_SLOTS 16
#define BA_SERVICE_IDLE_TICKS 300000000ull  // 3 s of the 100 MHz clock without a job: the slot leaves by itself
struct BaMail {
    ba_u64 seq;       // job number of this slot (monotonic)
    ba_u64 desc;      // the window's descriptor (pinned host address, device-readable)
    ba_u64 flags;     // tag base | use_mfma << 32 | same_l2_ok << 33
    ba_u64 stop;      // 1: leave
    ba_u64 done_seq;  // device -> host
    ba_u64 beat;      // heartbeat of the host's scheduler (bumped while the grid is resident): an idle slot stays while it moves
    ba_u64 pad[2];
};
struct BaServiceArgs {
    BaMail* mail;          // BA_SERVICE_SLOTS mailboxes (pinned host memory)
    ba_u64* cmd;           // device memory: 8 words per slot (job republished to the slot's workgroups)
    ba_u64* arrived;       // device memory: one counter per slot
    ba_u64 first_seq[BA_SERVICE_SLOTS];  // seq of every slot at launch
    int nslots, wgs_per_slot;
};

// ---- LDS carve-up of one workgroup (doubles unless noted); shared by the kernel and the planner
// the matrix part of the solver area: the reduced system (embedded into 32 / 64 rows for the register / block solvers), which also
// stages the split-chain tiles and the slice reduction (never live together)
__host__ __device__ inline size_t ba_solver_matrix_doubles(int n, int nlow, int G, int npair, int npar) {
    size_t a = n + 1 <= 32 ? 32 * 33 : (n + 1 <= 64 ? 64 * 65 : (size_t)(n + 1) * (size_t)(n + 2));
    const size_t sp = (size_t)npair * (size_t)(npar > 1 ? npar - 1 : 0) * 256;
    const size_t sl = (size_t)((nlow + G - 1) / (G > 0 ? G : 1)) * (size_t)G;
    if (sp > a) a = sp;
    if (sl > a) a = sl;
    return a;
}
// `alias`: the matrix part lives in the U area (windows whose Schur chains have ONE column piece per chunk -- no split tiles while
// U is live --: the matrix, the slice staging and L are only ever needed between the last chain and the back-substitution, when
// the U area is dead; the 64-row class saves 33 KB that way: two chunks of U instead of three for the BA10 window)
__host__ __device__ inline size_t ba_solver_doubles(int n, int nlow, int G, int npair, int npar, int panel, int alias = 0) {
    return (alias ? 0 : ba_solver_matrix_doubles(n, nlow, G, npair, npar)) + 3 * 64 +
           (panel ? BA_PANEL_DOUBLES : 0);  // + two column-broadcast buffers + scratch (+ the panel of the block solver)
}
// per-pose state: q t (8) + backup (8), R (9), t (3), H_pp (36), b_p (6), dx (6), solution (6)
__host__ __device__ inline size_t ba_pose_doubles(int F) { return (size_t)(F > 0 ? F : 1) * 82; }
// The multi-purpose area must hold: one chunk of U (ucols columns at the odd pitch ldu + 1), the staged [X~ | e~] rows of
// all edges, the back-substitution terms (3 per edge), the rows [A~ | e~] of the largest pose of any range (the pose-block
// chains stage whole poses), and the pose-block exchange staging.
__host__ __device__ inline size_t ba_uarea_doubles(int ucols, int ldu, int max_pt_edges, int maxEpose, int nhp, int G, int fix_points) {
    size_t hrows = 4096 / (size_t)(nhp > 0 ? nhp : 1);  // pose-block exchange staging: about 4096 values at a time
    if (hrows < 1) hrows = 1;
    if (hrows > (size_t)G) hrows = (size_t)G;
    size_t a = (size_t)nhp * hrows;
    const size_t m = (size_t)BA_MSTRIDE * (size_t)(maxEpose > 0 ? maxEpose : 1);
    if (m > a) a = m;
    if (!fix_points) {
        const size_t u = (size_t)ucols * (size_t)(ldu + 1), sx = (size_t)BA_SXS * max_pt_edges;  // (edges of one landmark-block pass)
        if (u > a) a = u;
        if (sx > a) a = sx;
    }
    return a;
}
__host__ __device__ inline size_t ba_lds_bytes(int F, int n, int nlow, int nhp, int G, int npair, int npar, int nfree, int maxEg,
                                               int maxLg, int fix_points, size_t uarea, int e2_edges, int panel, int uv_global = 0,
                                               int alias_sl = 0) {
    size_t d = ba_solver_doubles(n, nlow + nhp, G, npair, npar, panel, alias_sl) + (size_t)nlow + 2 * (size_t)nhp + 17 +
               (uv_global ? 0 : (size_t)maxEg * 2) + (size_t)maxLg * 3;
    d += ba_pose_doubles(F) + 2 * (size_t)G + 8;  // pose state, per-workgroup exchange values
    if (!fix_points) d += (size_t)maxLg * (3 + BA_XS + 3 + BA_XS + 3);
    d += uarea + (size_t)e2_edges * BA_E2S;
    size_t shorts = (size_t)maxEg * 5 + (size_t)maxLg + 1 + (size_t)maxLg * (nfree > 0 ? nfree : 1);
    return d * 8 + ((shorts * 2 + 15) & ~(size_t)15) + 64;
}

#endif
