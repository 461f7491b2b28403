/*
 * chd_phys.h — C ABI of the MI355X-native physics stage (libchd_phys.so).
 *
 * This library replaces the child process that the reference launches per video,
 *     subprocess.run(['./phys_optim', '--in_dir', ..., '--nframes', F, '--out_dir', ...,
 *                     '--w_com_lin', ..., '--w_com_ang', ..., '--w_ee', ..., '--w_smooth', ..., '--w_dur', ...])
 * (reference: scripts/run_phys_mocap.py:159-174; flags: towr_phys_optim/phys_optim.cpp:23-31),
 * by an in-process, batched call: many independent sequences are solved by one
 * persistent HIP kernel launch on one gfx950 device (one resident workgroup per compute
 * unit; the workgroups take sequences from a queue until the batch is drained, so a batch
 * may hold thousands of sequences of mixed length).
 *
 * Conventions: plain C, every function returns 0 on success and <0 on error (the
 * message is available through chd_phys_last_error), no exception crosses the boundary,
 * the library never retains caller pointers after a call returns, a handle is bound to
 * one HIP device and is not thread-safe (distinct handles may run concurrently, one
 * per GPU).  There is NO CPU fallback: chd_phys_create fails when no gfx950 device
 * can be opened.
 */
#ifndef CHD_PHYS_H
#define CHD_PHYS_H

#ifdef __cplusplus
extern "C" {
#endif

#define CHD_PHYS_ABI_VERSION 2   /* 2 (round 5): chd_config.reserved[0] became pipeline_chunk and `factorisation` is ignored (both since round 4, unversioned then),
                                  * chd_phys_solve_batch is pipelined, chd_phys_get_call_stats and chd_debug_get_state were added.  Callers compare
                                  * chd_phys_version() with the header they were built against (phys_optim.load_library, cli/phys_optim_main.cpp do). */
#define CHD_N_EE 4          /* NLP end-effector order: 0 L-toe, 1 R-toe, 2 L-heel, 3 R-heel (phys_optim.cpp:505-513) */
#define CHD_N_STAGES 6      /* 1.1, 1.2, 2.1, 2.2, 3, 4   (phys_optim.cpp:544-749) */
#define CHD_N_SNAPSHOTS 3   /* sol_out_no_dynamics / sol_out_dynamics / sol_out_durations (run_phys_mocap.py:182) */

/* Replaces the gflags of phys_optim (phys_optim.cpp:23-31) and the IPOPT options set at
 * phys_optim.cpp:567-578, 640, 652, 706, 743. */
typedef struct chd_config {
  double w_com_lin;        /* --w_com_lin  default 0.4  */
  double w_com_ang;        /* --w_com_ang  default 1.7  */
  double w_ee;             /* --w_ee       default 0.3  */
  double w_smooth;         /* --w_smooth   default 0.1  */
  double w_dur;            /* --w_dur      default 0.1  */
  int max_iter[CHD_N_STAGES];   /* 7000, 7000, 7000, 2500, 2000, 7000 */
  double tol;              /* IPOPT "tol", 1e-3 (phys_optim.cpp:578) */
  int threads_per_sequence;     /* workgroup size of the solver kernel: 0 or 512 (the phases are written for eight wavefronts;
                                   anything else is refused by chd_phys_create) */
  int stall_window;             /* > 0: a stage whose optimality error has not halved within this many iterations ends with
                                   status -2 (stage 3 then takes the stage-4 fallback) instead of running to max_iter; the hit
                                   is reported in chd_seq_out.stage_stalled.  0 (default) = off: IPOPT has no such rule */
  int max_workgroups;           /* resident workgroups of the solver launch; 0 = one per compute unit */
  int lds_kilobytes;            /* dynamic LDS per workgroup; 0 = all of a compute unit's (156 KB); smaller values narrow the
                                   factorisation panels (tuning / test knob) */
  int factorisation;            /* ignored since round 4 (kept for the ABI).  Rounds 2-3 carried two alternatives to the right-looking panel
                                 * factorisation -- left-looking matrix-core tiles gathered from the factor storage (1), a front held in the
                                 * accumulator registers (2); both were correct and measured slower on the MI355X, and were removed */
  int pipeline_chunk;           /* chd_phys_solve_batch / chd_phys_solve_dirs cut their B sequences into chunks of this many: the host builds the
                                 * tables of chunk k + 1 (and reads / writes the files of its neighbours) while the device solves chunk k, and up to
                                 * four chunks' launches share the device.  0 = automatic (a first chunk of 256, the rest in three equal chunks of 256 .. 4 096); < 0 = one chunk: set-up,
                                 * solve and fetch in turn, as rounds 1-3 did */
  int damping_rule;             /* 0 (default): the shipped rule -- the Levenberg damping is halved after a clean step, x 4 after a backtrack, x 1.5 after a second-model iteration.
                                 * 1: additionally x 4 after an ACCEPTED step that delivered less than a quarter of the merit reduction its quadratic model promised
                                 * (Levenberg-Marquardt's ratio test).  Measured on the MI355X (profiles/r06_globalisation_study.md): FEW or LONG sequences per call gain -- 500 walks in
                                 * one call 437 -> 602 sequences/s, one 600-frame sequence 5.6 -> 4.2 s, the pipeline's clips 21 -> 26 videos/s -- a launch of thousands of 90-frame walks
                                 * loses 7 % (2.4 % more iterations, a longer tail).  (Was reserved[0], always 0: layout and ABI version unchanged.) */
  int reserved[1];
} chd_config;

/* One sequence = the content of phys_optim_in_<char>/{skel,motion,terrain,contact}_info.txt
 * (reader being replaced: phys_optim.cpp:155-267).  All arrays are caller-owned, contiguous
 * fp64, row-major F x 3 (inertia: F x 6 = Ixx Iyy Izz Ixy Ixz Iyz). */
typedef struct chd_seq_in {
  int F;                       /* --nframes */
  double dt;                   /* motion_info.txt first token */
  const double* hip_l;         /* skel_info.txt  (phys_optim.cpp:176) */
  const double* hip_r;         /*                (:177) */
  double leg_len, heel_len, heel_dist, mass;   /* (:179-182) */
  const double* inertia;       /* (:183-187) */
  const double* com;           /* motion_info.txt (:199) */
  const double* euler;         /* (:200) extrinsic-xyz Euler, radians */
  const double* ltoe;          /* (:201) file order L-toe, L-heel, R-toe, R-heel */
  const double* lheel;         /* (:202) */
  const double* rtoe;          /* (:203) */
  const double* rheel;         /* (:204) */
  double normal[3];            /* terrain_info.txt (:216-218) */
  double point[3];             /* (:219-221) */
  int start_contact[4];        /* contact_info.txt, file order L-toe, L-heel, R-toe, R-heel (:236-264) */
  int n_phases[4];
  const double* durations[4];
} chd_seq_in;

/* One output snapshot = one sol_out_*.txt (writer being replaced: phys_optim.cpp:63-143).
 * Arrays are caller-allocated with room for `capacity` samples (>= F + 2 is always enough);
 * n_samples is the number the reference's `while (t <= T + 1e-5)` loop produces. */
typedef struct chd_snapshot {
  int capacity;
  int n_samples;               /* out */
  int num_frames_header;       /* out: int((T+1e-5)/dt)+1 (phys_optim.cpp:71) */
  double* base_lin;            /* capacity x 3 */
  double* base_ang_deg;        /* capacity x 3, degrees (phys_optim.cpp:97) */
  double* ee_pos;              /* 4 x capacity x 3, NLP ee order */
  double* ee_force;            /* 4 x capacity x 3 */
  unsigned char* contact;      /* 4 x capacity, 0/1 */
} chd_snapshot;

typedef struct chd_seq_out {
  chd_snapshot snap[CHD_N_SNAPSHOTS];
  int stage_status[CHD_N_STAGES];   /* 0 solved, 1 acceptable, -1 max-iter, -2 numerical failure, -3 internal (band overflow), 9 not run;
                                       all stages -4: the sequence was rejected at set-up (see build_error) and not solved */
  int stage_iters[CHD_N_STAGES];
  int stage_stalled[CHD_N_STAGES];  /* 1 = the stage was ended by the stall guard (chd_config.stall_window) */
  int stage_factorizations[CHD_N_STAGES];   /* KKT factorisations of the stage (>= iterations: inertia retries, rejected steps) */
  double stage_kkt_error[CHD_N_STAGES];
  double stage_constr_viol[CHD_N_STAGES];
  double stage_objective[CHD_N_STAGES];
  int dynamics_succeed;        /* success_log.txt line 1 (phys_optim.cpp:655) */
  int durations_succeed;       /* success_log.txt line 2 (phys_optim.cpp:709, :747) */
  /* problem sizes of the largest stage, for the roofline accounting */
  int n_vars, n_rows, kkt_dim, kkt_halfband, kkt_border;
  long long nnz_jac;           /* structural non-zeros of the constraint Jacobian, stage 2.2 */
} chd_seq_out;

typedef struct chd_handle chd_handle;   /* bound to one HIP device */
typedef struct chd_batch chd_batch;     /* device-resident problem data + results of one batch */

int chd_phys_version(void);
void chd_config_default(chd_config* cfg);

int chd_phys_create(const chd_config* cfg, int device_id, chd_handle** out);
void chd_phys_destroy(chd_handle* h);
const char* chd_phys_last_error(const chd_handle* h);   /* owned by the handle; "" if none */

/* Split interface (what the benchmark times is chd_batch_solve alone: inputs are resident
 * in HBM when it starts).
 *   upload : builds the per-sequence NLP structure tables on the host and copies inputs +
 *            tables to the device.  A sequence whose set-up fails (fewer than 8 frames, contact
 *            schedules of different total time, degenerate floor normal ...) is rejected on its
 *            own: the call still succeeds if at least one sequence is solvable, the rejected ones
 *            come back from fetch with stage_status -4 (chd_phys_last_error names the last one);
 *   solve  : runs stages 1.1 .. 3 (+4 where stage 3 failed) for every sequence on the device;
 *   fetch  : copies the three snapshots and the per-stage statistics back. */
int chd_batch_upload(chd_handle* h, int B, const chd_seq_in* in, chd_batch** out);
int chd_batch_solve(chd_handle* h, chd_batch* b);
int chd_batch_fetch(chd_handle* h, chd_batch* b, chd_seq_out* out /* B entries */);
void chd_batch_free(chd_handle* h, chd_batch* b);
/* Timing / accounting of the last chd_batch_solve on this batch (HIP events on the library's
 * stream): kernel_ms[0] = stages 1.1..3 launch, kernel_ms[1] = stage-4 fallback launch (0 if
 * not needed), host_ms = host work between the two launches; total_iters = sum of interior-
 * point iterations, alg_bytes = SURVEY 8(d) algorithmic bytes summed over all iterations. */
typedef struct chd_batch_stats {
  double kernel_ms[2];
  double host_ms;
  long long total_iters;
  long long total_factorizations;
  double alg_bytes;
  int n_fallback;              /* sequences that needed stage 4 */
  double phase_ms[24];         /* in-kernel wall-clock per phase, summed over sequences: 0 evaluation (f, grad, c, J, H), 1 evaluation (values only),
                                  2 factorisation, 3 substitution, 4 KKT mat-vec, 5 whole sequence,
                                  6-12 factorisation sub-phases, as the first wavefront sees them (6 copy, 8 panel load, 9 row solves,
                                  11 its look-ahead: three tiles + the next diagonal block, 10 write-back + wait for the other wavefronts' tiles, 12 border) */
  double max_seq_ms;           /* slowest single sequence (in-kernel wall clock) */
  int n_stalled;               /* stages ended by the stall guard (0 unless chd_config.stall_window > 0) */
  int n_rejected;              /* sequences rejected at set-up (not solved; stage_status -4) */
  int n_workgroups;            /* resident workgroups of the main launch */
} chd_batch_stats;
int chd_batch_get_stats(chd_handle* h, chd_batch* b, chd_batch_stats* out);

/* The whole call a user makes -- what replaces the loop `for video: subprocess.run(['./phys_optim', ...])` of run_phys_mocap.py:80-174 when the
 * inputs are already in memory: set-up (the reference does it inside the child process: phys_optim.cpp:428-540, nlp_formulation.cpp:79-203), upload,
 * solve, fetch -- pipelined over chunks of the batch (chd_config.pipeline_chunk), so that the host work hides behind the device's. */
int chd_phys_solve_batch(chd_handle* h, int B, const chd_seq_in* in, chd_seq_out* out);
/* Accounting of the last chd_phys_solve_batch / chd_phys_solve_dirs on this handle (wall-clock milliseconds unless stated). */
typedef struct chd_call_stats {
  double wall_ms;              /* the whole call */
  double prep_ms;              /* reading the input files (solve_dirs) */
  double setup_cpu_ms;         /* table builder, thread time summed over all sequences: / n_sequences = set-up cost of one sequence on one core */
  double setup_wall_ms;        /* table builder, wall time on host_threads threads (overlaps the device's work from the second chunk on) */
  double upload_ms;            /* allocation, host-to-device copies, launches */
  double wait_for_pool_ms;     /* the host had the next chunk ready and waited for a workspace pool: the device is the bottleneck */
  double finish_ms;            /* writing the output files (solve_dirs; overlaps the device's work) */
  double kernel_ms;            /* sum of the chunks' kernel times (launches overlap: this can exceed wall_ms) */
  double sequence_ms;          /* in-kernel wall clock summed over the sequences */
  double max_seq_ms;
  double alg_bytes;
  long long total_iters, total_factorizations;
  int n_sequences, n_chunks, chunk, host_threads, n_fallback, n_stalled, n_rejected;
} chd_call_stats;
int chd_phys_get_call_stats(chd_handle* h, chd_call_stats* out);

/* Drop-in for B invocations of ./phys_optim: reads the four input files of every in_dirs[i],
 * solves the batch, writes sol_out_no_dynamics.txt, sol_out_dynamics.txt, sol_out_durations.txt
 * and success_log.txt into out_dirs[i] (which must exist, as for the reference,
 * phys_optim.cpp:23).  status[i] (optional) = 0 ok, -1 unreadable inputs, -2 outputs not writable, -3 rejected at
 * set-up (inconsistent inputs; nothing written -- the reference's child process would have died on that video alone).
 * The return value is 0 as long as one directory could be solved. */
int chd_phys_solve_dirs(chd_handle* h, int B, const char* const* in_dirs, const char* const* out_dirs,
                        const int* nframes, int* status);

/* Test / profiling hooks (used by tests/ to compare single pieces of the hot path with the
 * oracle; not needed by a drop-in user).
 *   chd_debug_eval: evaluates f, grad, c and the dense Jacobian J (m x n row-major) and the
 *   Gauss-Newton objective Hessian H (n x n) of `stage` at x (NULL = initial guess) for
 *   sequence `seq` of an uploaded batch, on the device.  Any output pointer may be NULL. */
int chd_debug_sizes(chd_handle* h, chd_batch* b, int seq, int stage, int* n, int* m, int* kkt_dim, int* halfband, int* border);
/* chd_debug_linsolve: factor / solve self test of the sequence's KKT matrix of `stage` at the initial state, with dw * Dw on the
 * variable diagonal and -dval on the row diagonal (`which` is ignored since round 4: one factorisation is left); `reps` factorisations are timed.  rhs, x: kkt_dim doubles (KKT ordering).  info (14 doubles): replaced
 * pivots, clock ticks (100 MHz) of the factorisations, of the solve, the factorisation that ran, ten in-kernel phase timers. */
int chd_debug_linsolve(chd_handle* h, chd_batch* b, int seq, int stage, double dw, double dval, int which, int reps,
                       const double* rhs, double* x, double* info);
int chd_debug_eval(chd_handle* h, chd_batch* b, int seq, int stage, const double* x,
                   double* x_out, double* f, double* grad, double* c, double* J, double* H);
/* chd_debug_get_state: the point behind output snapshot `snapshot` (0: after stage 1.2, 1: after 2.2, 2: after 3 or its stage-4 fallback) of sequence `seq`
 * of a SOLVED batch, in the NLP's own variables -- node_vars: the node variables in variable-set order (the first *n_node_vars entries of every stage's x;
 * room for chd_seq_out.n_vars doubles is enough), phase_durations: all phase durations of the four end-effectors, NLP order, concatenated (n_phases[e] each;
 * room for 4 * 64).  tests/test_quality_gate.py recomputes objective and constraint violation at this point with the oracle's model functions. */
int chd_debug_get_state(chd_handle* h, chd_batch* b, int seq, int snapshot, double* node_vars, int* n_node_vars, double* phase_durations, int* n_phases);

#ifdef __cplusplus
}
#endif
#endif /* CHD_PHYS_H */
