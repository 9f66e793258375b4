/*
 * v2e_amd.h -- C ABI of libv2e_amd.so, the MI355X (gfx950) implementation of the
 * v2e hot path.  Plain pointers and sizes only; no torch types.  Every device
 * pointer is memory owned by the caller (PyTorch-ROCm tensors in the Python host);
 * the library allocates only its private scratch at create().
 *
 * All functions return 0 on success or a negative V2E_E* code; nothing throws
 * across the boundary.  Unless stated otherwise a call only enqueues work on the
 * given hipStream_t (passed as void*) and returns.
 *
 * Reference interfaces replaced (paths relative to the SensorsINI/v2e tree):
 *   v2ecore/emulator.py:439-511   EventEmulator._init            -> v2e_emu_init_state
 *   v2ecore/emulator.py:656-775   generate_events, front half    -> v2e_emu_count
 *     (lin_log emulator_utils.py:18-45, rescale_intensity_frame :48-54,
 *      low_pass_filter :57-109, subtract_leak_current :114-134,
 *      compute_event_map :137-173, generate_shot_noise :297-351)
 *   v2ecore/emulator.py:791-942   iteration loop, refractory, event list,
 *      shot-noise events, base update (get_event_list_from_coords :1024-1059)
 *                                                                -> v2e_emu_rank + v2e_emu_emit
 *   v2ecore/emulator.py:867-869   randperm shuffle               -> v2e_emu_permute
 *   v2ecore/model.py:10-226       UNet / down / up               -> v2e_unet_forward
 *   v2ecore/model.py:229-300      backWarp                       -> v2e_slomo_prep / _fuse
 *   v2ecore/slomo.py:404-433      per-t blend, warps, fusion     -> v2e_slomo_prep / _fuse
 */
#ifndef V2E_AMD_H
#define V2E_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2E_OK 0
#define V2E_EINVAL (-22)
#define V2E_ENOMEM (-12)
#define V2E_EHIP (-5)      /* a HIP runtime call failed; see v2e_last_error() */
#define V2E_ENODEV (-19)
#define V2E_EOVERFLOW (-75)

/* frame element types accepted by the emulator (emulator.py:663 converts all to f64) */
#define V2E_DT_U8 0
#define V2E_DT_F32 1
#define V2E_DT_F64 2

#define V2E_RNG_TAPE 0   /* random numbers / timestamps / permutations supplied by host */
#define V2E_RNG_PHILOX 1 /* counter-based Philox4x32-10 in-kernel (include/v2e_detmath.h) */

/* record flags */
#define V2E_FLAG_EVENTS_DROPPED 1u /* event buffer capacity exceeded; counts are still exact */
#define V2E_FLAG_ITERS_CLAMPED 2u  /* max events per pixel exceeded max_iters: frame NOT emitted */
#define V2E_FLAG_SYNC_TIMEOUT 4u   /* in-kernel workgroup rendezvous timed out: results invalid */

/*
 * DVS model parameters, read at every call (the reference reads plain attributes
 * lazily, so callers may change them between frames, e.g. set_dvs_params()).
 * Scalars that the reference computes in Python doubles stay doubles here.
 */
typedef struct v2e_emu_params {
    int32_t f64_state;      /* 1: lp/base planes are double (cutoff_hz > 0), 0: float   */
    int32_t scalar_thres;   /* 1: sigma_thres == 0, thresholds are Python floats        */
    int32_t rng_mode;       /* V2E_RNG_TAPE / V2E_RNG_PHILOX                            */
    int32_t shuffle;        /* philox mode: apply the keyed bijection per iteration     */
    double pos_thres_nominal, neg_thres_nominal;
    double pos_thres_scalar, neg_thres_scalar; /* used when scalar_thres                */
    double sigma_thres;
    double cutoff_hz;
    double leak_rate_hz;
    double leak_jitter_fraction;
    double noise_rate_cov_decades;
    double refractory_period_s;
    double shot_noise_rate_hz;
    double shot_noise_inten_factor; /* emulator.py:210, 0.25 */
    float pos_pre_scalar, neg_pre_scalar; /* nominal/thres when scalar_thres (host torch.div) */
    uint64_t seed;          /* philox key */
    int32_t log_input;      /* hdr=True (emulator.py:304, 666): frames are already log intensity, no lin_log;
                               the state is float64 even without a cutoff (f64_state must be 1) */
    int32_t photoreceptor_noise; /* emulator.py:694-703: low-passed Gaussian noise on the photoreceptor output instead of
                                    Poisson shot events (frame-at-a-time API only); pass shot_noise_rate_hz = 0 with it */
    double photoreceptor_noise_vrms; /* emulator_utils.py:177-290, evaluated by the host */
} v2e_emu_params;

/* Per-(frame, clip) result record, written on device. */
typedef struct v2e_frame_rec {
    int32_t max_events;  /* max_num_events_any_pixel (emulator.py:773) */
    uint32_t flags;
    uint32_t n_signal;   /* signal events after refractory filtering   */
    uint32_t n_events;   /* signal + shot                              */
    uint32_t n_on, n_off;/* totals incl. shot (emulator.py:1038-1040)  */
    uint64_t ev_offset;  /* first row of this frame in the clip's event region */
} v2e_frame_rec;

typedef struct v2e_emu v2e_emu;

const char *v2e_last_error(void);
int v2e_version(void);

/*
 * H, W: sensor size; n_clips: independent pixel arrays advanced in lock-step by one
 * launch (state planes are [n_clips][npx_pad], npx_pad = v2e_emu_npx_pad(H,W));
 * max_iters: cap on events per pixel per frame (hist scratch is sized by it).
 */
int v2e_emu_create(int H, int W, int n_clips, int max_iters, int device, v2e_emu **out);
int v2e_emu_destroy(v2e_emu *h);
int64_t v2e_emu_npx_pad(int H, int W);

/* Device pointers to the persistent per-pixel state (caller-owned, [n_clips][npx_pad]).
 * lp/base are double planes when params.f64_state else float planes. */
int v2e_emu_bind_state(v2e_emu *h, void *lp, void *base, float *ts_mem,
                       float *pos_thres, float *neg_thres, float *noise_rate);

/*
 * First frame (emulator.py:681-717 + _init :439-511).  frame: [n_clips][H*W] of
 * `dtype`.  Tape mode: thres_pos/thres_neg hold the torch.normal draws (pre-clamp)
 * and noise_rate the final exp() values (or NULL when unused); Philox mode: pass
 * NULL and they are generated in-kernel.
 */
int v2e_emu_init_state(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype,
                       double t_frame, const float *thres_pos, const float *thres_neg,
                       const float *noise_rate, void *stream);

/* Photoreceptor-noise state (emulator.py:684, 694-703): pn_arr is the float64 plane [n_clips][npx_pad]
 * `photoreceptor_noise_arr` (caller-owned, zero on the first frame); randn_tape the float32 draws
 * torch.randn(shape) of the coming frame in tape mode (NULL in Philox mode).  Both are consumed by the next
 * v2e_emu_count when params.photoreceptor_noise is set; NULL pn_arr switches the feature off. */
/* SCIDVS pixel (emulator.py:56-80, 719-725, 747): device planes [n_clips][npx_pad] scidvs_highpass and
 * scidvs_previous_photo (state dtype) and scidvs_tau_arr (float32; filled by v2e_emu_init_state in Philox mode, by the
 * caller in tape mode).  first_frame_idx: the frame index at which scidvs_previous_photo is taken from the frame itself
 * (emulator.py:720-722).  All NULL switches it off.  Runs on the count / rank / scan / emit kernels. */
int v2e_emu_set_scidvs(v2e_emu *h, void *highpass, void *previous_photo, float *tau, uint32_t first_frame_idx);

/* Model-state planes the reference keeps as attributes and the kernels otherwise never store: log_new_frame (emulator.py:666),
 * c_minus_s_frame (:753, zero without CSDVS) and diff_frame (:749-754), float64 [n_clips][npx_pad] each (caller-owned), written
 * by every v2e_emu_count from then on -- what show_dvs_model_state / record_single_pixel_states (emulator.py:756-764, 985-1006)
 * read.  All three or none (NULL switches it off).  Frame-at-a-time kernels only. */
int v2e_emu_set_model_state_planes(v2e_emu *h, double *log_new_frame, double *c_minus_s_frame, double *diff_frame);

int v2e_emu_set_pnoise(v2e_emu *h, void *pn_arr, const float *randn_tape);

/* Centre-surround DVS (cs_lambda_pixels; emulator.py:245-272, 707-716, 753-754, 1061-1124).  The surround plane
 * cs_surround_frame [n_clips][npx_pad] (state dtype, caller-owned) is what v2e_emu_count subtracts from the photoreceptor
 * output: diff = (photoreceptor + noise - surround) - base; NULL switches the feature off.  Per frame the caller
 *   1. v2e_emu_lp_preview: the frame's lp_log_frame (the low-pass of emulator.py:685-690 applied to the state as it is,
 *      written to lp_out, the state untouched),
 *   2. v2e_csdvs_update:   steps the diffuser against it (below),
 *   3. v2e_emu_count ...:  the frame itself (recomputes the same lp_log_frame).
 * On the first frame the surround is a copy of lp_log_frame and base_log_frame is zero (emulator.py:1062-1063, 715). */
int v2e_emu_set_csdvs(v2e_emu *h, const void *surround);
int v2e_emu_lp_preview(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype, const double *t_prev,
                       const double *t_frame, uint32_t frame_idx, void *lp_out, void *stream);
/* EventEmulator._update_csdvs (emulator.py:1098-1124) on device planes of H x W values (float64 if f64 else float32):
 * up to num_steps Euler steps  h += alpha_p (p - h) + alpha_h conv2d(ReplicationPad2d(1)(h.float()), [[0,1,0],[1,-4,1],[0,1,0]])
 * stopping after the first step whose max |change| is <= max_change_to_stop (1e-5 in the reference).  The float32 5-point
 * sum is taken in kernel order ((((top + left) - 4 centre) + right) + bottom) -- what torch's CPU convolution does on planes
 * of 200 x 200 and more (DAVIS346 included; on small planes its backend sums in another order, see csdvs.hip).  h_scratch: a
 * second plane of the same size.  steps_taken: the reference's `steps`; last_max_change (may be NULL): its `max_change`.
 * Synchronises the stream once per 32 steps. */
/* CSDVS inside a device-resident run (v2e_emu_run; one clip): per frame of the COMING run the Euler parameters the host
 * computes as emulator.py:1066-1096 does (alpha_p, alpha_h, num_steps), a scratch surround plane and a plane for the frame's
 * lp_log_frame (state dtype, [npx_pad] each, caller-owned), and steps_taken_dev (device int32 [n_frames]: the reference's `steps`
 * per frame, readable after the run).  The run then steps the diffuser on the stream before every frame: every one of the frame's
 * num_steps launches is enqueued and the ones after the step that settles (max |change| <= max_change_to_stop, evaluated on the
 * device) return at once -- no host step, capturable in the run's hipGraph.  n_frames = 0 clears it. */
int v2e_emu_set_csdvs_run(v2e_emu *h, void *h_scratch, void *lp_scratch, const double *alpha_p, const double *alpha_h,
                          const int *num_steps, int n_frames, double max_change_to_stop, int *steps_taken_dev);
int v2e_csdvs_update(const void *p_plane, void *h_plane, void *h_scratch, int H, int W, int f64, double alpha_p, double alpha_h,
                     int num_steps, double max_change_to_stop, int *steps_taken, double *last_max_change, void *stream);

/*
 * Front half of one time step for all clips: photoreceptor, IIR, leak, event
 * counts, shot-noise decisions, global max.  leak_randn/shot_rand: [n_clips][npx_pad]
 * host-drawn tapes (tape mode) or NULL (Philox).  frame_idx: Philox counter word and
 * record slot (rec slot = frame_idx % ring).  t_prev/t_frame: per-clip host arrays.
 */
int v2e_emu_count(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype,
                  const double *t_prev, const double *t_frame, uint32_t frame_idx,
                  const float *leak_randn, const float *shot_rand, void *stream);

/* Blocking: copy the per-clip records of `frame_idx` to host (syncs the stream). */
int v2e_emu_read_rec(v2e_emu *h, uint32_t frame_idx, v2e_frame_rec *recs_host, void *stream);

/* Shot-noise decisions as their own pass (tape mode only: the reference draws `rand`
 * after the per-iteration randperms, emulator.py:868 then :906 / emulator_utils.py:338). */
int v2e_emu_shot(v2e_emu *h, const v2e_emu_params *p, const void *frame, int dtype,
                 uint32_t frame_idx, const float *shot_rand, void *stream);

/* Grow the per-iteration scratch if max_events exceeds the create()-time max_iters
 * (blocking; frame-at-a-time API only, after v2e_emu_read_rec). */
int v2e_emu_reserve_iters(v2e_emu *h, int max_events, void *stream);

/*
 * Back half, step 1: refractory filter (emulator.py:830-846) and per-(iteration,
 * polarity) histograms + exclusive scan; no state is modified.  ts_table: device
 * [n_clips][n_ts] float32 timestamps (torch.linspace drawn by the host, tape mode)
 * or NULL for the in-kernel formula.  Totals are readable with
 * v2e_emu_read_iter_counts afterwards.
 */
int v2e_emu_rank(v2e_emu *h, const v2e_emu_params *p, uint32_t frame_idx, const float *ts_table,
                 int n_ts, void *stream);

/*
 * Back half, step 2 (after v2e_emu_rank): dense [N,4] float32 (t,x,y,p) list in
 * reference order (emulator.py:861-923), base/ts_mem update (:936-942), record.
 * events: device [n_clips][cap][4] float32.  ev_offset0: per-clip host array of the
 * row at which this frame's events start (NULL: continue after the previous frame).
 */
int v2e_emu_emit(v2e_emu *h, const v2e_emu_params *p, uint32_t frame_idx,
                 const float *ts_table, int n_ts, float *events, uint64_t cap,
                 const uint64_t *ev_offset0, void *stream);

/* Blocking: per-iteration (on_i, off_i) totals of the last emitted frame,
 * host array [n_clips][2*(max_events+1)]; the last pair is the shot-noise pair. */
int v2e_emu_read_iter_counts(v2e_emu *h, uint32_t frame_idx, int n_iters, uint32_t *counts_host,
                             void *stream);

/* out[dst0 + j] = in[src0 + idx[j]], j < n   (events_curr_iter[idx], emulator.py:869) */
int v2e_emu_permute(v2e_emu *h, const float *events_in, float *events_out, const int32_t *idx,
                    uint64_t row0, uint64_t n, void *stream);

/* One frame of the frame-at-a-time API in one call (Philox mode, one clip; replaces the sequence v2e_emu_count /
 * read_rec / rank / read_iter_counts / emit + the row copy of EventEmulator.generate_events, emulator.py:619-1022): the
 * frame (host pointer if frame_on_host, else device) is counted, ranked and scanned, the frame record and per-key totals
 * come back in ONE read, the rows are emitted into events_dev[0 .. n) and copied to pinned host memory owned by the
 * handle (*events_host, valid until the next call; NULL when n = 0).  out8 = {n_events, n_on, n_off, n_signal, M, 0, 0, 0}.
 * Returns 0; 1 if M > the handle's max_iters (counted, nothing emitted: grow with v2e_emu_reserve_iters, then
 * v2e_emu_rank / v2e_emu_emit); 2 if n_events > cap (same); negative on error.  Synchronises `stream`. */
int v2e_emu_frame(v2e_emu *h, const v2e_emu_params *p, const void *frame, int frame_on_host, int dtype, double t_prev,
                  double t_frame, uint32_t frame_idx, float *events_dev, uint64_t cap, uint32_t *out8,
                  const float **events_host, void *stream);
/* Where the NEXT v2e_emu_frame call puts its rows: a caller-owned PINNED host buffer of cap_rows rows of 4 floats (hipHostMalloc
 * / hipHostRegister memory, e.g. a torch pin_memory tensor), written by the frame's last kernel directly -- *events_host then
 * equals pinned_rows and the caller needs no copy.  One frame only; a frame with more rows than cap_rows is delivered in the
 * handle's own buffer as before (*events_host says which).  NULL / 0 cancels. */
int v2e_emu_frame_host_rows(v2e_emu *h, float *pinned_rows, uint64_t cap_rows);

/*
 * Philox-mode, fully device-resident multi-frame run: frames [n_frames][n_clips][H*W],
 * t_prev/t_frame host arrays [n_frames][n_clips].  Records for frame f go to
 * recs_dev[f][clip] (device, caller-owned).  No host synchronisation inside.
 * use_graph, low two bits: 0 plain launches, 1 capture the launch sequence into a hipGraph
 * (cached while every baked-in pointer/size is unchanged), 2 instrumented (see
 * v2e_emu_last_profile).  Pipeline: k_chain -- K frames per launch with the per-pixel state in
 * registers (32 on small grids and without a refractory period, 8 on large grids where the
 * refractory rule may force a redo, 1 where a clip's workgroups cannot be co-resident), the rule-off
 * speculation of a launch validated (and redone from a checkpoint on a miss) by the next launch --
 * with the event list built behind it in batches (k_ctot, k_cframe, k_cemit) on a second stream that
 * joins `stream` before the call's work ends; on small grids the state-independent part of every
 * frame (lin-log, low-pass coefficient, Philox draws, leak step, shot decisions) is precomputed by
 * k_ahead on a third.  |128: one frame per launch (clips on which the refractory rule is active on
 * most frames).  |16 selects the unfused count/rank/scan/emit kernels, one frame at a time (kept for
 * A/B measurements; also what runs photoreceptor noise, float64 log-encoded frames and more than
 * 1024 events per pixel and frame); |256 insists on k_chain (error where it cannot run).  All give
 * identical results.
 * Round 5: the run's frame scalars, first frame index and the ADDRESS of `frames` reach the device through one small kernel that
 * reads a pinned staging set (no copy-engine transfer), and the k_chain pipeline's kernels read the frames through that device
 * variable: with use_graph & 1 a cached graph is replayed over whatever `frames` buffer a call names (events / recs_dev are still
 * baked in: callers alternate between fixed buffer sets), so a caller need not copy frames into one fixed buffer per run.
 * Round 6, PIPELINED RUNS (use_graph = 0 | 1024; wherever the k_chain pipeline runs): plain launches on four streams, no
 * graph and no join at the run's end.  `stream` carries the chain's launches and nothing else, so consecutive runs' chains follow one
 * another in one hardware queue; the run's upload, zero fills and k_ahead records go to a stream of the handle that runs ahead of the
 * chain (beside the run before; one k_ahead launch and one wait per run where the handle's ring of frame slots holds the whole run), the
 * emission's tables and rows to two more that finish beside the run after; everything two runs in flight would share exists twice.  `stream` then orders the PIXEL STATE only: the run's event rows and records are complete behind v2e_emu_run_wait
 * (host) / v2e_emu_run_join (a stream), which the caller must call before reading them; every other entry point of the handle joins by
 * itself.  Callers alternate two sets of `events` / `recs_dev` buffers (a run that names the buffers of the run before it waits for that
 * run).  | 2048: the caller vouches that `frames` are resident (nothing enqueued on `stream` still writes them): the run's head then
 * does not wait for `stream`, i.e. for the chain of the run before.  Results are identical to every other mode.
 */
int v2e_emu_run(v2e_emu *h, const v2e_emu_params *p, const void *frames, int dtype, int n_frames,
                const double *t_prev, const double *t_frame, uint32_t frame_idx0, float *events,
                uint64_t cap, v2e_frame_rec *recs_dev, int use_graph, void *stream);
/* `stream` waits (device side, no host block) for every piece of every pipelined run enqueued on the handle so far. */
int v2e_emu_run_join(v2e_emu *h, void *stream);
/* The scratch set (0 / 1) the run enqueued last took if the library enqueued it in pieces (the ticket of that run, valid until the
 * second-next such run), or -1: that run is whole on its stream. */
int v2e_emu_run_ticket(v2e_emu *h);
/* Host-blocking: every piece of the last pipelined run with that ticket has completed (its rows and records can be read). */
int v2e_emu_run_wait(v2e_emu *h, int ticket);
/* The records [n_frames][n_clips] of the last pipelined run with that ticket in pinned host memory of the handle (copied there behind
 * the run's last rows; valid behind v2e_emu_run_wait, until the second-next pipelined run); *n_recs = how many. */
const v2e_frame_rec *v2e_emu_run_recs(v2e_emu *h, int ticket, uint64_t *n_recs);

/* After an instrumented v2e_emu_run (blocking): summed milliseconds per kernel class and frames.
 * k_chain pipeline (a HIP event before and after every chain launch, on its stream): ms_count = first
 * launch's start to last launch's end (gaps included), ms_rank = the sum of the chain kernels' own
 * durations, ms_emit = the k_cemit launches (v2e_emu_last_profile_pipe), ms_scan 0;
 * |16: k_count, k_rank, k_scan, k_emit (a hipEvent before every launch). */
int v2e_emu_last_profile(v2e_emu *h, double *ms_count, double *ms_rank, double *ms_scan,
                         double *ms_emit, int *launches);

/* The launch schedule v2e_emu_run uses for a run of n_frames frames (no GPU needed; for tests and tooling): returns the
 * number of chain launches and, if out != NULL, writes per launch 8 int32: the frames it advances (f0, nf; nf = 0: the
 * tail launch that only validates), the frames of its predecessor it validates (pf0, pnf), the emission batch it must wait
 * for before reusing ring slots, the k_ahead batch it needs, the k_ahead batch enqueued behind it, the emission batch
 * that is final once it is enqueued; -1 = none.  Batch b = frames [b * frames_per_batch, (b + 1) * frames_per_batch). */
int v2e_emu_chain_plan(int n_frames, int frames_per_launch, int frames_per_batch, int ring_batches, int has_refractory,
                       int fused_records, int32_t *out, int cap);

/* Emission batches timed by the last instrumented run, the frames per emission batch, and the
 * number of chain launches ms_count covers (step_launches may be NULL). */
int v2e_emu_last_profile_pipe(v2e_emu *h, int *emit_batches, int *frames_per_batch, int *step_launches);
/* The chain launches of the last instrumented run one by one, in launch order (microseconds; the last one is the tail launch
 * that only validates): *n = how many there were, the first min(cap, *n) written to us (may be NULL). */
int v2e_emu_last_profile_launches(v2e_emu *h, float *us, int cap, int *n);

/* Device time stamps of the chain kernel, per launch, in the configuration that is actually timed (graph replay, side streams running):
 * every k_chain launch of the runs enqueued after v2e_emu_launch_stamps(h, runs > 0, NULL, 0, NULL, NULL) leaves the wall-clock time
 * of its first workgroup's start and of its last workgroup's end (two atomics per workgroup; s_memrealtime, 100 MHz) in a ring of
 * `runs` runs.  With out_ns != NULL or n_runs != NULL the call first synchronises the device and copies the last min(cap_runs, runs
 * stamped) runs out, oldest first: out_ns[run][launch][2] = {start, end} in nanoseconds of the device clock, *launches_per_run
 * entries per run (0 / 0: the run had no such launch); then `runs` takes effect (0 switches the stamps off; a changed value
 * re-allocates the ring and forgets what it held).  What bench.py's roofline object divides the chain's algorithmic bytes by. */
int v2e_emu_launch_stamps(v2e_emu *h, int runs, uint64_t *out_ns, int cap_runs, int *n_runs, int *launches_per_run);

/* The event writer of the handle's k_chain pipeline: 1 = k_cpull (a thread per output row; needs 32 bytes of pixel ballots per frame
 * of the table sets, group and key: 281 MB at 346x260 with max_iters = 64), 0 = k_cemit (the push writer: same rows, 1.5x the bytes
 * written), -1 = no such run yet.  Chosen when the pipeline's scratch is allocated: the pull where its tables fit a quarter of the free
 * device memory AND an absolute budget (24 GB per scratch set; V2E_AMD_PULL_BUDGET_MB), V2E_AMD_EMIT_PULL=0 forces the push writer. */
int v2e_emu_event_writer(v2e_emu *h);

/* Which pipeline the last v2e_emu_run on this handle used: kind 0 = unfused count/rank/scan/emit, 3 = k_chain (K frames
 * per launch, state in registers; records from k_ahead), 4 = k_chain with the per-frame records built inside the chain;
 * frames_per_launch of the dependency chain and frames per emission batch. */
int v2e_emu_last_pipeline(v2e_emu *h, int *kind, int *frames_per_launch, int *frames_per_batch);

/* ------------------------------------------------------------- SuperSloMo */

/* one 2-D convolution layer of the UNet (model.py: nn.Conv2d + leaky_relu 0.1) */
typedef struct v2e_conv_desc {
    const float *weight; /* device, pre-packed [Cin][k][k][Cout] (v2e_pack_conv_weight) */
    const float *bias;   /* device [Cout] */
    int32_t cin, cout, ksize;
    int32_t split_kind;  /* what weight_s3 holds: 0 three bf16 pieces (v2e_pack_conv_weight_s3); 2 | (scale_log2 << 8): two float16
                          * pieces of the weights times 2^scale_log2 (v2e_pack_conv_weight_h2) */
    const void *weight_s3; /* device, split weights or NULL: f32-MFMA kernel only */
} v2e_conv_desc;

/* repack torch [Cout][Cin][k][k] -> [Cin][k][k][Cout] on device */
int v2e_pack_conv_weight(const float *w_oihw, float *w_packed, int cout, int cin, int k, void *stream);

/*
 * The same weights split exactly into three bf16 pieces (w = p0 + p1 + p2) for the bf16-matrix-core convolution that
 * keeps f32 accuracy (six piece products per multiply, f32 accumulation; v2e_amd/csrc/slomo_s3.h):
 * [ceil(Cin/16)][k*k][3][2][Cout][8 bf16] = 6 bytes per weight of the padded tensor (a last partial 16-channel chunk
 * is padded with zero weights): w_s3 must hold ceil(cin/16)*16 * k*k * cout * 6 bytes.
 */
int v2e_pack_conv_weight_s3(const float *w_oihw, void *w_s3, int cout, int cin, int k, void *stream);
/* The same with TWO float16 pieces (w = h0 + h1 + r, |r| <= 2^-22 |w|) for the three-product convolution (conv_math "fp16x2":
 * half the matrix-core work, a product good to ~2^-21; v2e_amd/csrc/slomo_s3.h): [ceil(Cin/16)][k*k][2][2][Cout][8 f16] =
 * 4 bytes per weight of the padded tensor.  The weights are split times 2^scale_log2 (choose it so that the layer's largest
 * |w| 2^scale_log2 is below 2^14: nothing overflows float16 and no weight piece falls into its subnormal range); the
 * convolution divides it out again exactly.  Set v2e_conv_desc.split_kind = 2 | (scale_log2 << 8) with it. */
int v2e_pack_conv_weight_h2(const float *w_oihw, void *w_h2, int cout, int cin, int k, int scale_log2, void *stream);
/* Range guard of the two-float16-piece convolutions inside v2e_unet_forward: while a device flag is set here (per host thread;
 * NULL switches it off), a forward pass ORs 1 into it if a convolution's input holds an inf or a NaN (seen in the range slot its
 * producer left: the largest |output| of every convolution is tracked on the device, and a two-piece convolution stages its
 * activations times the power of two that puts that maximum in [2^13, 2^14) -- finite activations of any magnitude stay in
 * float16's normal range).  The caller zeroes the flag, runs the pass, reads it back, and on 1 redoes it with the exact three-piece
 * weights -- what SloMoEngine's default conv math "auto" does.  A two-piece v2e_conv2d_lrelu called on its own stages unscaled and
 * is not watched. */
int v2e_conv_set_range_flag(int *device_flag);

/* activations in the same split form: x [n][c][h][w] f32 -> xs [3 pieces][n][c/8][h][w][8 bf16] (6 bytes per element;
 * c a multiple of 8).  A convolution takes such an input with pre = 3 (3x3 layers with split weights). */
int v2e_split3_nchw(const float *x, void *xs, int n, int c, int h, int w, void *stream);

/*
 * y = leaky_relu(conv2d(cat(x0, x1), W) + b, 0.1), stride 1, zero pad (k-1)/2, NCHW f32.
 * x1 may be NULL (c1 = 0).  pre: 0 none, 1 avg_pool2d(x0,2) fused on load (x0 is
 * [N][c0][2H][2W]), 2 bilinear x2 upsample (align_corners=False) fused on load
 * (x0 is [N][c0][H/2][W/2]), 3 x0 is a pre-split tensor (v2e_split3_nchw; 3x3 layers with split weights only).
 */
int v2e_conv2d_lrelu(const float *x0, int c0, const float *x1, int c1, int pre,
                     const v2e_conv_desc *conv, float *y, int n, int h, int w, void *stream);

/* 23-conv UNet forward (model.py:198-226).  convs: 23 descriptors in forward order;
 * workspace: device scratch of v2e_unet_workspace_bytes(n,h,w) bytes. */
int64_t v2e_unet_workspace_bytes(int n, int h, int w, int cin);
int v2e_unet_forward(const float *x, int cin, const v2e_conv_desc *convs, int cout, float *y,
                     int n, int h, int w, void *workspace, void *stream);

/*
 * slomo.py:405-419 for `n_t` time points of one batch: flow blend, two backWarps and
 * assembly of the 12-channel interpolation input
 * [I0,I1,F01(2),F10(2),Ft1(2),Ft0(2),g1,g0] -> x12 [n_t*b][12][h][w].
 * flow: flow-UNet output [b][4][h][w]; tcoef: DEVICE array [n_t][6] float32 =
 * {fCoeff[0..3], wCoeff[0..1]} of slomo.py:406-407,429, evaluated by the host in doubles.
 */
int v2e_slomo_prep(const float *i0, const float *i1, const float *flow, const float *tcoef, int n_t,
                   int b, int h, int w, float *x12, void *stream);

/* slomo.py:421-433: refine flows, visibility, two backWarps, fusion -> out [n_t*b][1][h][w] */
int v2e_slomo_fuse(const float *i0, const float *i1, const float *x12, const float *intrp,
                   const float *tcoef, int n_t, int b, int h, int w, float *out, void *stream);

/* slomo.py:352-368 (auto_upsample): max over the batch and both flow pairs of the SQUARED flow magnitude vx*vx + vy*vy (float32,
 * each operation rounded) of flow [b][4][h][w] = [F_0_1 x, y, F_1_0 x, y], as float32 bits in *out_bits (device); the host takes
 * one float32 sqrt of it (sqrt is monotone: the same number as the reference's max of sqrt planes).  0xffffffff: a NaN flow. */
int v2e_slomo_max_speed2(const float *flow, int b, int h, int w, uint32_t *out_bits, void *stream);

/* ------------------------------------------- SloMo <-> emulator hand-off */

/*
 * Pillow-exact 8-bit resampling of n images [ih][iw] -> [oh][ow] (dataloader.py:142 LANCZOS,
 * slomo.py:439 BILINEAR).  Tables from v2e_amd/resample.py (libImaging/Resample.c
 * precompute_coeffs + normalize_coeffs_8bpc): bounds [out][2] = (first tap, taps), coef
 * [out][ksize] 22-bit fixed point.  tmp: device scratch [n][ih][ow] (horizontal pass output).
 */
int v2e_resample_u8(const uint8_t *in, uint8_t *tmp, uint8_t *out, int n, int ih, int iw, int oh, int ow,
                    const int32_t *hbounds, const int32_t *hcoef, int hksize, const int32_t *vbounds,
                    const int32_t *vcoef, int vksize, void *stream);

/* ToTensor + Normalize(mean, 1): out = in / 255 - mean (slomo.py:148-162) */
int v2e_u8_to_f32_norm(const uint8_t *in, float *out, int64_t n, float mean, void *stream);

/* revNormalize + ToPILImage: out = (uint8)(int)((in + mean) * 255) (slomo.py:437); reorder != 0:
 * in is [U][B][hw] (interpolation batch order), out is [B][U][hw] (time order, slomo.py:441) */
int v2e_f32_to_u8_trunc(const float *in, uint8_t *out, int U, int B, int hw, float mean, int reorder, void *stream);

/* ------------------------------------------- stage-1 pre-processing (v2e.py:687-738) -- PARITY UNPINNED:
 * OpenCV 4.x's published INTER_AREA / BGR2GRAY 8-bit algorithms restated (see v2e_amd/csrc/preproc.hip); cv2 is in neither
 * tree nor this image, so the restatement has not been compared with cv2 itself. */

/* cv2.resize(src, (dw, dh), interpolation=cv2.INTER_AREA) of n uint8 images [sh][sw][cn] (cn = 1 or 3, interleaved), shrinking
 * only.  Integer scale factors: pass xofs = NULL (box sums).  Otherwise the tables of computeResizeAreaTab, grouped per
 * destination index (v2e_amd/preproc.py area_tab): entries [xofs[dx], xofs[dx + 1]) of (xsi, xalpha) feed column dx, in order;
 * the same for rows. */
int v2e_resize_area_u8(const uint8_t *src, uint8_t *dst, int n, int sh, int sw, int dh, int dw, int cn, const int32_t *xofs,
                       const int32_t *xsi, const float *xalpha, const int32_t *yofs, const int32_t *ysi, const float *yalpha, void *stream);
/* cv2.cvtColor(src, cv2.COLOR_BGR2GRAY) of npx interleaved BGR uint8 pixels.  gray_shift 15: OpenCV 4.x's RGB2Gray<uchar>
 * ((3735 B + 19235 G + 9798 R + 2^14) >> 15); 14: OpenCV 3.x's ((1868 B + 9617 G + 4899 R + 2^13) >> 14). */
int v2e_bgr2gray_u8(const uint8_t *src_bgr, uint8_t *dst, int64_t npx, int gray_shift, void *stream);

/* ----------------------------------------------------- event sinks */

/* AEDAT-2.0 records of aedat2_output.py:155-173: out_bytes gets n x 8 bytes (big-endian int32 address,
 * big-endian int32 timestamp in us).  Layout constants as set in aedat2_output.py:41-77 for the sensor
 * (346x260 / 240x180: xshift 12, yshift 22, pshift 11, flipx = flipy = 1; 640x480: 1, 11, 0).
 * noise_from >= 0: events [noise_from, n) get the special-event bit (label_signal_noise). */
int v2e_events_pack_aedat2(const float *events, void *out_bytes, int64_t n, int sizex, int sizey, int xshift,
                           int yshift, int pshift, int flipx, int flipy, int64_t noise_from, void *stream);

/* HDF5 "events" rows of emulator.py:955-965: uint32 [n][4] = (t*1e6 in float32, x, y, p with -1 -> 0) */
int v2e_events_pack_h5(const float *events, uint32_t *out, int64_t n, void *stream);

/* Lossless 8-byte wire format of an event row [t, x, y, p] (emulator.py:1024-1059 builds them as float32[4]):
 * bits 63..32 the float32 bits of t, 31..18 x, 17..4 y, bit 0 (p > 0); x, y < 16384.  Used by the multi-GPU
 * all-gather of the event streams (half the bytes over xGMI); unpack restores the float32 rows exactly. */
int v2e_events_pack64(const float *events, uint64_t *out, int64_t n, void *stream);
int v2e_events_unpack64(const uint64_t *in, float *events, int64_t n, void *stream);

/* Lossless 4-byte wire format for sensors up to 2048 x 1024: the rows of a run come in blocks of one time stamp (all events
 * of one (frame, iteration) share it: emulator.py:793-796, 861-870), so t travels once per block.  payload[i] = x | y << 11
 * | (p > 0) << 21; runs[0] = number of blocks R in its low 32 bits (never more than cap_runs) and the overflow flags below in
 * bits 62 (coordinate) and 63 (table too small), so that a receiver of the table sees them too; runs[1 + r] = float32 bits of t << 32 | index of the block's first event
 * (in order).  cap_runs: entries the run table holds besides runs[0] (a bound is sum over frames of max(iterations, 1));
 * scratch: device uint32 [v2e_events_pack32_scratch_words(n)], scratch[0] after the call: bit 0 a coordinate did not fit,
 * bit 1 the run table was too small (the caller then sends pack64).  unpack32 restores the float32 rows bit for bit. */
int v2e_events_pack32(const float *events, int64_t n, uint32_t *payload, uint64_t *runs, int64_t cap_runs,
                      uint32_t *scratch, void *stream);
int64_t v2e_events_pack32_scratch_words(int64_t n);
int v2e_events_unpack32(const uint32_t *payload, int64_t n, const uint64_t *runs, float *events, void *stream);

/* EventRenderer.accumulate_event_frame (renderer.py:368-400, hist2d_numba_seq v2e_utils.py:474-486):
 * current_frame (float64 [bins_y][bins_x], device) = clip(current_frame + hist(ON) - hist(OFF), +-full_scale).
 * scratch_diff: device int32 [bins_y][bins_x], zero on entry, left zero. */
int v2e_events_accumulate_frame(const float *events, int64_t n, double *current_frame, int32_t *scratch_diff,
                                int bins_y, int bins_x, double y_lo, double y_hi, double x_lo, double x_hi,
                                double full_scale, void *stream);

/* Finished DVS frame in 0..1: (current_frame + full_scale) / (2 full_scale) in float64 (renderer.py:245-247, normalize_frame). */
int v2e_frame_normalize(const double *current_frame, double *out, int n, double full_scale, void *stream);

/* EventRenderer.render_events_to_frames on a device-resident packet (renderer.py:161-366; v2e_amd/csrc/render.hip).
 * v2e_render_area_segments: the AREA_COUNT windows of a packet of n events (renderer.py:253-266, 292-298): area_counts
 * (device int32 [nw][nh], carried from packet to packet, updated), seg_end[k] = end index (exclusive) of complete window k
 * (the event that filled a cell; it opens the next window and is counted again there), out2[0] = number of complete windows,
 * out2[1] = 1 if the walk ended on a trigger at the packet's last event. */
int v2e_render_area_segments(const float *events, int64_t n, int32_t *area_counts, int nw, int nh, double area_dimension,
                             int area_count, int32_t *seg_end, int64_t cap, int32_t *out2, void *stream);
/* v2e_render_packet: every event i < n_used (= n - 1: the packet's last event is never accumulated, renderer.py:303-306)
 * goes into the frame(s) of its exposure window -- mode 1 DURATION: bounds[0..n_bounds) the frame start times T_k, frame k =
 * [T_k, T_k+1] (both ends inclusive, as searchsorted left / right), the open frame from T_{n_bounds-1}; 2 COUNT: frame
 * i / count_per_frame; 3 AREA_COUNT: seg_end[0..n_seg); 4 SOURCE: frame 0 -- as hist(ON) - hist(OFF) over bins_y x bins_x bins
 * (hist2d_numba_seq, v2e_utils.py:474-486).  frames_out [n_complete][bins_y][bins_x] float64 = (clip(+-full_scale) +
 * full_scale) / (2 full_scale) (renderer.py:245-247, 396-400); cur_out (if has_open) = the clipped, un-normalised frame the
 * packet ends in.  diff: device int32 scratch [(n_complete + has_open)][bins_y * bins_x]. */
int v2e_render_packet(const float *events, int64_t n_used, int mode, const double *bounds, int n_bounds,
                      int64_t count_per_frame, const int32_t *seg_end, int n_seg, int n_complete, int has_open, int32_t *diff,
                      double *frames_out, double *cur_out, int bins_y, int bins_x, double y_lo, double y_hi, double x_lo,
                      double x_hi, double full_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* V2E_AMD_H */
