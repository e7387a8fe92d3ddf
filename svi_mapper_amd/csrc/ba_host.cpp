// ba_host.cpp — C ABI of the bundle adjustment (include/svi_hot.h): graph construction with the
// reference's rules, structure analysis, and the Levenberg-Marquardt driver.
//
// Reference behaviour mirrored here
//   Cg2oOptimizer::_setAndgetPose                 src/optimization/Cg2oOptimizer.cpp:1229-1290
//   Cg2oOptimizer::_getEdgeLinearAcceleration     :982-997
//   Cg2oOptimizer::_setLandmarkMeasurementsWORLD  :1383-1466  (+ factories :999-1073)
//   Cg2oOptimizer::_optimizeUnLimited             :954-980
//   Cg2oOptimizer::_applyOptimizationToLandmarks  :1468-1512 (pruning rule)
//   g2o OptimizationAlgorithmLevenberg::solve / SparseOptimizer::optimize  
// The linear algebra is NOT g2o's: landmarks are eliminated per 3x3 block (Schur complement) and
// the reduced camera system is factorised tile-sparse on the GPU; the increment is the same up to
// round-off (SURVEY.md "Quick facts").
#include "ba_host.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>

#include "ba_math.h"

namespace svi {
// launchers defined in ba_kernels.hip / ba_chol.hip
void ba_linearize_lm(const BaDev& d, int cur, void* st);
void ba_linearize_pose(const BaDev& d, int cur, void* st);
void ba_linearize_aux(const BaDev& d, int cur, int rank, void* st);
void ba_chi2_aux(const BaDev& d, int which, int rank, void* st);
void ba_pose_finalize(const BaDev& d, const int* red_slot, int rank, int n_ranks, int with_invert, double lambda, void* st);
void ba_publish(const BaDev& d, int n, double* h_scal, int* h_status, int seq, void* st);
void ba_lin_post(const BaDev& d, int n_ranks, void* st);
void ba_invert_landmarks(const BaDev& d, double lambda, void* st);
void ba_schur(const BaDev& d, const StageSignals* sg, unsigned long long seq, int reserve_per_se, int launch_wgs, void* st);
void ba_assemble(const BaDev& d, int sub0, int sub1, int accumulate, void* st);
void ba_update_poses(const BaDev& d, int cur, double lambda, int scale_mode, int rank, void* st);
void ba_backsub_chi2(const BaDev& d, int cur, double lambda, void* st);
void ba_chi2_only(const BaDev& d, int which, void* st);
void ba_reduce_trial_scalars(const BaDev& d, int aux_state, int with_lin, int n_pub, double* h_scal, int* h_status, int seq, void* st);
void ba_debug_jacobians(const BaDev& d, int cur, const int* e_orig, double* err, double* Jp, double* Jl, void* st);
void ba_debug_aux_jacobians(const BaDev& d, int cur, double* se3_err, double* se3_Ji, double* se3_Jj, double* acc_err, double* acc_J, void* st);
void ba_configure_kernels(int TS);
int chol_potrf_probe(int tile, int reps, int stop_after, double* ms);
int chol_factor_solve(const CholPlan& p, double* S, double* Lt, double* Linv, double* g, double* x, double lambda, int n,
                      int* status, void* st, const PoseTail* tail, int* tail_done);
// polls a workgroup of the one-launch backward substitution grants a pending entry before it gives up (ba_chol.hip reads it;
// svi_debug_set_backsolve_spin_limit: the tests shrink it to provoke the time-out path)
std::atomic<int> g_backsolve_spin_limit{1 << 22};
int chol_factor_range(const CholPlan& p, double* S, double* Lt, double* Linv, double* g, double* x, double lambda, int n, int* status,
                      void* st, const PoseTail* tail, int* tail_done, int st_begin, int st_end, int solve, double* Su, double* gu);
int build_structure(svi_ba* ba); // ba_structure.cpp
int reupload_state(svi_ba* ba);

void PhaseTimer::begin(int phase, hipStream_t s)
{
    if (!on) return;
    if (used == pool.size()) {
        Rec r{phase, nullptr, nullptr};
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        pool.push_back(r);
    }
    pool[used].phase = phase;
    (void)hipEventRecord(pool[used].a, s);
}
void PhaseTimer::end(hipStream_t s)
{
    if (!on) return;
    (void)hipEventRecord(pool[used].b, s);
    ++used;
}
void PhaseTimer::collect()
{
    // A record whose events have not completed yet (a sweep enqueued behind the numbers the host has just read may still be
    // running) stays in the pool for the next collect - it is neither dropped nor waited for; records marked "discarded"
    // (phase < 0: a speculated sweep whose trial was rejected or whose block ended) are skipped.
    size_t keep = 0;
    for (size_t i = 0; i < used; ++i) {
        if (pool[i].phase < 0) continue;
        float t = 0.f;
        const hipError_t e = hipEventElapsedTime(&t, pool[i].a, pool[i].b);
        if (e == hipSuccess) { ms[pool[i].phase] += t; calls[pool[i].phase]++; }
        else if (e == hipErrorNotReady) { if (keep != i) std::swap(pool[keep], pool[i]); ++keep; }
        else (void)hipGetLastError(); // (the record is lost; the error must not surface in an unrelated call)
    }
    if (keep) (void)hipGetLastError(); // hipErrorNotReady is sticky in hipGetLastError
    used = keep;
}
void PhaseTimer::drop_last()
{
    if (on && used > 0) pool[used - 1].phase = -1;
}
void PhaseTimer::release()
{
    for (auto& r : pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    pool.clear();
    used = 0;
}
} // namespace svi

using namespace svi;

namespace {

const double kIdentity12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};

// the device structures of the last initialize() are no longer valid (the buffers themselves stay: ba_structure.cpp re-uses them)
void invalidate_device(svi_ba* ba)
{
    ba->initialized = false;
    ba->d = BaDev{};
    ba->plan = CholPlan{};
}

void free_device(svi_ba* ba)
{
    for (auto& b : ba->pool) b.release();
    ba->pool.clear();
    ba->raw_log.release();
    ba->raw_flags.release();
    ba->raw_uploaded = 0; ba->raw_cap = 0;
    if (ba->h_scal) { (void)hipHostFree(ba->h_scal); ba->h_scal = nullptr; }
    if (ba->h_status) { (void)hipHostFree(ba->h_status); ba->h_status = nullptr; }
    invalidate_device(ba);
}

#define SVI_TRY(x) do { int rc_ = (x); if (rc_ != SVI_OK) return rc_; } while (0)

int allreduce(svi_ba* ba, double* buf, size_t count)
{
    if (ba->opt.n_ranks <= 1) return SVI_OK;
    if (!ba->ar) return fail(SVI_ERR_STATE, "n_ranks > 1 but no all-reduce hook set (svi_ba_set_allreduce)");
    ba->timer.begin(SVI_PH_ALLREDUCE, ba->stream);
    const int rc = ba->ar(ba->ar_user, buf, count, ba->stream);
    ba->timer.end(ba->stream);
    if (rc != 0) return fail(SVI_ERR_COMM, "all-reduce hook returned %d", rc);
    return SVI_OK;
}

// ---------------------------------------------------------------------------------------------
// LM pieces
// ---------------------------------------------------------------------------------------------
// waits until the device has published sequence number `seq` (h_scal / h_status[0] are valid then)
int wait_published(svi_ba* ba, int seq)
{
    volatile int* flag = ba->h_status + 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spin & 0xFFF) == 0xFFF) { // every few thousand polls: has the stream died or drained without publishing?
            const hipError_t q = hipStreamQuery(ba->stream);
            if (q == hipSuccess) { if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break; SVI_HIP(hipStreamSynchronize(ba->stream)); if (*flag == seq) break; return fail(SVI_ERR_HIP, "result publication lost"); }
            if (q != hipErrorNotReady) return fail(SVI_ERR_HIP, "stream failed while waiting for the trial results: %s", hipGetErrorString(q));
            // A stream wait on a memory value has no time-out of its own.  If a stage of the Schur reduction were never published
            // (a defect), the main stream would sit in its wait for ever: after 30 s the host publishes the values itself - the queue
            // drains (with a wrong reduced system, which is discarded with the error) - and the handle stops using the second stream.
            if (ba->overlap_ok && ba->stage_seq > 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                for (int k = 0; k < kMaxStages; ++k) __atomic_store_n(ba->sig_stage[k], ~0ull >> 1, __ATOMIC_RELEASE);
                __atomic_store_n(ba->sig_lin, ~0ull >> 1, __ATOMIC_RELEASE);
                ba->overlap_ok = false;
                (void)hipStreamSynchronize(ba->stream);
                (void)hipStreamSynchronize(ba->stream_schur);
                return fail(SVI_ERR_INTERNAL, "a stage of the Schur reduction was never published (stream wait released by the host after 30 s)");
            }
        }
        __builtin_ia32_pause();
    }
    return SVI_OK;
}

int read_scalars(svi_ba* ba, int n)
{
    if (ba->timer.on) { // profiling: a full stream synchronisation, the phase events are collected behind it
        SVI_HIP(hipMemcpyAsync(ba->h_scal, ba->d.scal, sizeof(double) * n, hipMemcpyDeviceToHost, ba->stream));
        SVI_HIP(hipMemcpyAsync(ba->h_status, ba->d.chol_status, sizeof(int), hipMemcpyDeviceToHost, ba->stream));
        SVI_HIP(hipMemsetAsync(ba->d.chol_status, 0, sizeof(int), ba->stream)); // handed over: clean for the next trial
        SVI_HIP(hipStreamSynchronize(ba->stream));
        ba->timer.collect();
        return SVI_OK;
    }
    // One LM decision per trial hangs on these few numbers: a kernel stores them into pinned host memory and the host
    // spins on the sequence number - no copy-engine launches, no interrupt wake-up (tens of microseconds each way).
    // The publishing kernel also clears the status word for the next trial.
    const int seq = ++ba->pub_seq;
    ba_publish(ba->d, n, ba->h_scal, ba->h_status, seq, ba->stream);
    SVI_HIP(hipGetLastError());
    return wait_published(ba, seq);
}

// sums of the trial (chi2, step scale) and their way to the host; with one rank and no profiling the reduction
// kernel publishes them itself
// aux_state: the state whose pose-only edges the reduction evaluates itself (-1: their sums are already in scal[6..7]);
// a linearisation whose closing sums were deferred (linearize) has them taken here as well
// The part of a linearisation that needs nothing from the host: Jacobian sweep (K2 + K3) and the pose-only edges of `state`.
void enqueue_sweep(svi_ba* ba, int state)
{
    BaDev& d = ba->d;
    hipStream_t s = ba->stream;
    PhaseTimer& t = ba->timer;
    ba->sweep_timer.begin(0, s);
    t.begin(SVI_PH_LINEARIZE_LM, s);   ba_linearize_lm(d, state, s);   t.end(s);
    t.begin(SVI_PH_LINEARIZE_POSE, s); ba_linearize_pose(d, state, s); t.end(s);
    ba->sweep_timer.end(s);
    t.begin(SVI_PH_POSE_EDGES, s);
    ba_linearize_aux(d, state, ba->opt.rank, s);
    // (the caller closes SVI_PH_POSE_EDGES behind the pose sums)
}

// spec_state >= 0: while the host waits for the trial's numbers the device already runs the sweep of that state - the
// linearisation of the NEXT iteration if the trial is accepted (it nearly always is).  The host round trip (read, LM rule,
// launch: 12-25 us) then hides behind 58 us of kernels instead of leaving the device idle.  A rejected trial finds the
// buffers of its linearisation overwritten and linearises its state again (optimize_block).
int reduce_and_read_trial(svi_ba* ba, int n, int aux_state, int spec_state = -1)
{
    const int with_lin = ba->lin_post_deferred ? 1 : 0;
    ba->lin_post_deferred = false;
    ba->timer.begin(SVI_PH_CHI2, ba->stream);
    if (ba->opt.n_ranks == 1 && !ba->timer.on) {
        const int seq = ++ba->pub_seq;
        ba_reduce_trial_scalars(ba->d, aux_state, with_lin, n, ba->h_scal, ba->h_status, seq, ba->stream);
        if (spec_state >= 0) { enqueue_sweep(ba, spec_state); ba->spec_lin_state = spec_state; }
        SVI_HIP(hipGetLastError());
        return wait_published(ba, seq);
    }
    ba_reduce_trial_scalars(ba->d, aux_state, with_lin, 0, nullptr, nullptr, 0, ba->stream);
    ba->timer.end(ba->stream);
    SVI_HIP(hipGetLastError());
    SVI_TRY(allreduce(ba, ba->d.scal, 4)); // chi2 robust / plain, landmark and pose parts of the step scale
    return read_scalars(ba, n);
}

// computeActiveErrors + buildSystem at state `cur`; leaves chi2 (robust, plain) and max|H_jj| in h_scal[0,1,5]
int linearize(svi_ba* ba, bool read = true)
{
    BaDev& d = ba->d;
    hipStream_t s = ba->stream;
    PhaseTimer& t = ba->timer;
    if (ba->spec_lin_state != ba->cur) enqueue_sweep(ba, ba->cur); // (else: enqueued behind the previous trial's reduction)
    ba->spec_lin_state = -1;
    // lambda of the trial that follows is known unless this is the first linearisation of a block (lambda_0 needs max H_jj):
    // the landmark blocks are then inverted in the same launch
    ba->hinv_valid = !read && d.Ll > 0;
    ba->hinv_lambda = ba->lambda;
    ba_pose_finalize(d, ba->red_slot, ba->opt.rank, ba->opt.n_ranks, ba->hinv_valid ? 1 : 0, ba->lambda, s);
    t.end(s);
    SVI_HIP(hipGetLastError());
    // Several ranks: the pose sums (Hpp | bp | chi2 | max diag) only have to be exchanged when the host needs
    // max |H_jj| for lambda_0, i.e. in front of the first trial of a block.  Otherwise every rank adds its own partial
    // Hpp / bp to its share of the reduced system and the all-reduce of that system sums them (with this rank's chi2 in
    // the two doubles in front of g): one collective per iteration less.
    ba->lin_local = ba->opt.n_ranks > 1 && !read && d.n_sub > 0;
    if (ba->lin_local) return SVI_OK;
    SVI_TRY(allreduce(ba, d.lin_buf, (size_t)d.lin_count));
    // one rank, results not needed before the trial: the closing sums (chi2 of the linearisation point, max |H_jj|) are
    // taken by the kernel that closes the trial - only the host reads them, and it reads them there
    if (!read && ba->opt.n_ranks == 1) { ba->lin_post_deferred = true; return SVI_OK; }
    ba_lin_post(d, ba->opt.n_ranks, s);
    SVI_HIP(hipGetLastError());
    return read ? read_scalars(ba, 8) : SVI_OK; // unread: the numbers wait in scal[8..10] for the next read
}

// one trial: solve (H + lambda I) dx = b through the Schur complement, apply, evaluate.
// results: h_scal[0] robust chi2, [1] plain chi2, [2]+[3] step scale; *failed
// k_assemble with the pose terms matching what linearize() left in Hpp / bp: totals (rank 0 adds them) or this rank's share
// stage < 0: every sub-tile.  With several stages the tiles and g start a trial as zeros (zero_reduced_system) and everything
// that reaches them - the assembled blocks of their stage, the updates of the factorisation's levels - is added.
// the pose terms an assembly adds must match what linearize() left in Hpp / bp: totals (rank 0 adds them) or this rank's share;
// set before the kernel that assembles is launched (k_schur itself, or k_assemble)
void set_assembly_flags(svi_ba* ba)
{
    BaDev& d = ba->d;
    d.add_pose_terms = (ba->lin_local || ba->opt.rank == 0) ? 1 : 0;
    d.lin_from_red = ba->lin_local ? 1 : 0;
}

void assemble(svi_ba* ba, int stage = -1)
{
    BaDev& d = ba->d;
    if (d.asm_in_schur == 1) return; // (k_schur has written the tiles itself)
    const int acc = d.n_stages > 1 ? 1 : 0;
    if (stage < 0) ba_assemble(d, 0, d.n_sub, acc, ba->stream);
    else ba_assemble(d, ba->sub_stage_ptr[stage], ba->sub_stage_ptr[stage + 1], acc, ba->stream);
}

int zero_reduced_system(svi_ba* ba)
{
    BaDev& d = ba->d;
    if (d.n_stages > 1 && d.asm_in_schur != 1 && d.NT > 0)
        SVI_HIP(hipMemsetAsync(d.g, 0, sizeof(double) * ((size_t)d.NT * d.TS + (size_t)d.n_tiles * d.TS * d.TS), ba->stream));
    return SVI_OK;
}

// the Schur reduction of a trial on its own stream, the factorisation level by level behind the stages it finishes
bool use_overlap(const svi_ba* ba)
{
    const bool off = getenv("SVI_NO_OVERLAP") != nullptr; // (read per trial: the tests switch it)
    return ba->overlap_ok && !off && ba->d.n_stages > 1 && ba->opt.n_ranks == 1 && !ba->timer.on;
}

int trial(svi_ba* ba, double lambda, bool* failed, bool speculate = false)
{
    BaDev& d = ba->d;
    hipStream_t s = ba->stream;
    PhaseTimer& t = ba->timer;
    // (the status word is clean: whoever read it last cleared it)
    // the trial poses ride in the last launch of the solve (an extra workgroup that waits for its dx entries): a launch of their
    // own was 5 us + a boundary between the backward substitution and the landmark back-substitution (not while phases are timed)
    set_assembly_flags(ba);
    const int scale_mode = ba->opt.n_ranks <= 1 ? 0 : (ba->lin_local ? 2 : 1);
    PoseTail tail{};
    tail.src = d.pose[ba->cur]; tail.dst = d.pose[ba->cur ^ 1]; tail.pose_red = d.pose_red; tail.bp = d.bp; tail.scal = d.scal;
    tail.red_base = d.red_base; tail.Pn = d.Pn; tail.lin_from_red = d.lin_from_red;
    tail.wl = (scale_mode == 0 || ba->opt.rank == 0) ? lambda : 0.0; tail.wb = (scale_mode == 1 && ba->opt.rank != 0) ? 0.0 : 1.0;
    int tail_done = 0;
    static const bool no_tail = getenv("SVI_NO_POSE_TAIL") != nullptr; // (A/B timing)
    static const bool no_spec = getenv("SVI_NO_SPECULATION") != nullptr;
    const PoseTail* tailp = (t.on || no_tail) ? nullptr : &tail;
    SVI_TRY(zero_reduced_system(ba));
    // staged reduction with the tiles assembled by k_schur: the factorisation's updates go to d.upd[buf] (zero at this point)
    const bool split = d.n_stages > 1 && d.asm_in_schur == 1 && d.upd[0] != nullptr && d.NT > 0;
    const size_t upd_bytes = sizeof(double) * ((size_t)d.NT * d.TS + (size_t)d.n_tiles * d.TS * d.TS);
    const int buf = (int)(ba->staged_trials & 1);
    double *gu = nullptr, *Su = nullptr;
    if (split) {
        ++ba->staged_trials;
        if (!ba->upd_clean[buf]) SVI_HIP(hipMemsetAsync(d.upd[buf], 0, upd_bytes, s));
        ba->upd_clean[buf] = false;
        gu = d.upd[buf]; Su = d.upd[buf] + (size_t)d.NT * d.TS;
    }
    t.begin(SVI_PH_SCHUR, s);
    if (!(ba->hinv_valid && ba->hinv_lambda == lambda)) ba_invert_landmarks(d, lambda, s);
    ba->hinv_valid = false; // (a second trial of the iteration comes with another lambda)
    if (use_overlap(ba) && d.NT > 0) {
        // main stream: "linearisation ready" -> Schur stream: k_schur, stage by stage -> main stream: per stage, (assemble its
        // sub-tiles and) factorise the dependency levels whose columns it holds.  Only the first stage's wait is on the critical
        // path: from then on the factorisation (17 launches, ~15 us each at config 4) is what the reduction has to keep ahead of.
        const unsigned long long seq = ++ba->stage_seq;
        StageSignals sg{};
        for (int k = 0; k < d.n_stages; ++k) sg.sig[k] = ba->sig_stage[k];
        sg.seq = seq;
        SVI_HIP(hipStreamWriteValue64(s, ba->sig_lin, seq, 0));
        SVI_HIP(hipStreamWaitValue64(ba->stream_schur, ba->sig_lin, seq, hipStreamWaitValueGte, ~0ull));
        ba_schur(d, &sg, seq, ba->schur_reserve_per_se, ba->schur_launch_wgs, ba->stream_schur);
        if (split) { // the other update buffer, for the next trial: zeroed here, where nothing waits for it
            SVI_HIP(hipMemsetAsync(d.upd[buf ^ 1], 0, upd_bytes, ba->stream_schur));
            ba->upd_clean[buf ^ 1] = true;
        }
        SVI_HIP(hipGetLastError());
        const int n_steps = ba->plan.n_steps;
        int st = 0;
        for (int stage = 0; stage < d.n_stages; ++stage) {
            SVI_HIP(hipStreamWaitValue64(s, ba->sig_stage[stage], seq, hipStreamWaitValueGte, ~0ull));
            assemble(ba, stage);
            int st1 = st;
            while (st1 < n_steps && ba->level_stage[st1] == stage) ++st1;
            const bool last = stage == d.n_stages - 1;
            if (last) st1 = n_steps;
            if (chol_factor_range(ba->plan, d.S, d.Lt, d.Linv, d.g, d.dx, lambda, 6 * d.Pf, d.chol_status, s, tailp, &tail_done, st, st1, last ? 1 : 0, Su, gu) != 0)
                return fail(SVI_ERR_HIP, "Cholesky kernels could not be configured (LDS request refused)");
            st = st1;
        }
        SVI_HIP(hipGetLastError());
    } else {
        ba_schur(d, nullptr, ++ba->stage_seq, 0, 0, s);
        t.end(s);
        t.begin(SVI_PH_ASSEMBLE, s); assemble(ba); t.end(s);
        SVI_HIP(hipGetLastError());
        SVI_TRY(allreduce(ba, ba->lin_local ? d.red_base : d.g, (size_t)d.red_count + (ba->lin_local ? 2 : 0)));
        t.begin(SVI_PH_CHOLESKY, s);
        if (d.NT > 0 && chol_factor_range(ba->plan, d.S, d.Lt, d.Linv, d.g, d.dx, lambda, 6 * d.Pf, d.chol_status, s, tailp, &tail_done, 0, ba->plan.n_steps, 1, Su, gu) != 0)
            return fail(SVI_ERR_HIP, "Cholesky kernels could not be configured (LDS request refused)");
        t.end(s);
    }
    t.begin(SVI_PH_BACKSUB_UPDATE, s);
    if (!tail_done) ba_update_poses(d, ba->cur, lambda, scale_mode, ba->opt.rank, s);
    ba_backsub_chi2(d, ba->cur, lambda, s);
    t.end(s);
    // (evaluates the pose-only edges of the trial state itself)
    const bool spec = speculate && ba->opt.n_ranks == 1 && !t.on && !no_spec;
    SVI_TRY(reduce_and_read_trial(ba, 12, ba->cur ^ 1, spec ? (ba->cur ^ 1) : -1));
    // -3: a hand-over of the one-launch backward substitution never arrived (ba_chol.hip).  That is a defect of the library, not
    // a property of the matrix: it must not enter the LM rule as "not positive definite" (lambda would grow and the trajectory
    // silently leave the reference's) - the call fails instead
    if (ba->h_status[0] == -3) {
        ba->stats.backsolve_timeouts++;
        ba->spec_lin_state = -1;
        return fail(SVI_ERR_INTERNAL, "backward substitution hand-over timed out (internal error; the estimates are unchanged)");
    }
    // with several ranks the landmark blocks (and so the status word) are local: a failure anywhere arrives as a
    // non-finite chi2 through the all-reduce, so every rank takes the same branch of the LM rule
    *failed = ba->h_status[0] != 0 || (ba->opt.n_ranks > 1 && !std::isfinite(ba->h_scal[0]));
    return SVI_OK;
}

// keep the host copy of the estimates in step with the device (the reference reads them back in
// _applyOptimizationTo*, and the admission rule of later measurements uses them)
int download_state(svi_ba* ba)
{
    BaDev& d = ba->d;
    std::vector<double> hp((size_t)12 * d.Pn), hl((size_t)3 * std::max(d.Ll, 1));
    if (d.Pn) SVI_HIP(hipMemcpyAsync(hp.data(), d.pose[ba->cur], sizeof(double) * 12 * d.Pn, hipMemcpyDeviceToHost, ba->stream));
    const int Ltot = (int)ba->lms.size();
    if (ba->opt.n_ranks > 1) {
        SVI_HIP(hipMemsetAsync(ba->lm_all, 0, sizeof(double) * 3 * Ltot, ba->stream));
        if (d.Ll) SVI_HIP(hipMemcpyAsync(ba->lm_all + (size_t)3 * ba->L0, d.lm[ba->cur], sizeof(double) * 3 * d.Ll, hipMemcpyDeviceToDevice, ba->stream));
        SVI_TRY(allreduce(ba, ba->lm_all, (size_t)3 * Ltot));
        hl.resize((size_t)3 * std::max(Ltot, 1));
        if (Ltot) SVI_HIP(hipMemcpyAsync(hl.data(), ba->lm_all, sizeof(double) * 3 * Ltot, hipMemcpyDeviceToHost, ba->stream));
        SVI_HIP(hipStreamSynchronize(ba->stream));
        for (int s = 0; s < Ltot; ++s) memcpy(ba->lms[ba->lm_order[s]].p, &hl[(size_t)3 * s], 24);
    } else {
        if (d.Ll) SVI_HIP(hipMemcpyAsync(hl.data(), d.lm[ba->cur], sizeof(double) * 3 * d.Ll, hipMemcpyDeviceToHost, ba->stream));
        SVI_HIP(hipStreamSynchronize(ba->stream));
        for (int l = 0; l < d.Ll; ++l) memcpy(ba->lms[ba->lm_order[ba->L0 + l]].p, &hl[(size_t)3 * l], 24);
    }
    for (int s = 0; s < d.Pn; ++s) memcpy(ba->poses[ba->pose_order[s]].T, &hp[(size_t)12 * s], 96);
    ba->timer.collect();
    return SVI_OK;
}

} // namespace

// Brings the host copy of poses and landmarks in step with the device if an optimize() has run since the last read.
// With several ranks this contains the all-gather of the landmark shards: the first reading call after an optimize()
// is collective.
int ensure_host(svi_ba* ba)
{
    if (!ba->host_stale) return SVI_OK;
    SVI_HIP(svi::enter_device(ba->opt.device));
    SVI_TRY(download_state(ba));
    ba->host_stale = false;
    return SVI_OK;
}

namespace {

int optimize_block(svi_ba* ba, int iterations, int* performed)
{
    if (!ba->initialized) return fail(SVI_ERR_STATE, "svi_ba_optimize before svi_ba_initialize");
    SVI_HIP(svi::enter_device(ba->opt.device));
    const svi_ba_options& o = ba->opt;
    int done = 0;
    ba->spec_lin_state = -1; // (the state may have been edited since the last block)
    for (int it = 0; it < iterations; ++it) {
        // Only the first iteration of a block needs the linearisation results on the host before the trial can be
        // launched (lambda_0 = tau * max H_jj); afterwards lambda is known and the host reads chi2 of the linearisation
        // together with the results of the first trial - one host round trip per iteration instead of two.
        const bool read_now = it == 0 || ba->timer.on;
        SVI_TRY(linearize(ba, read_now));
        double chi = 0.0;
        bool have_lin = read_now;
        if (read_now) {
            chi = ba->h_scal[0];
            ba->last_robust = chi; ba->last_plain = ba->h_scal[1]; ba->have_chi = true;
            if (it == 0) { ba->lambda = o.lm_tau * ba->h_scal[5]; ba->ni = 2.0; } // computeLambdaInit, per optimize() call
        }
        double rho = 0.0;
        int q = 0;
        bool stop_inf = false;
        do {
            bool failed = false;
            // a rejected trial whose speculative successor sweep has overwritten this iteration's linearisation: once more
            if (ba->spec_lin_state >= 0 && ba->spec_lin_state != ba->cur) {
                ba->sweep_timer.drop_last(); // (that sweep linearised a state that was never accepted: not a linearisation of the LM loop)
                ba->spec_lin_state = -1;
                SVI_TRY(linearize(ba, false));
            }
            SVI_TRY(trial(ba, ba->lambda, &failed, it + 1 < iterations));
            if (!have_lin) {
                chi = ba->h_scal[8];
                ba->last_robust = chi; ba->last_plain = ba->h_scal[9]; ba->have_chi = true;
                have_lin = true;
            }
            double temp = ba->h_scal[0];
            if (failed) { temp = DBL_MAX; ba->stats.chol_failures++; }
            else { ba->last_robust = ba->h_scal[0]; ba->last_plain = ba->h_scal[1]; }
            const double scale = ba->h_scal[2] + ba->h_scal[3] + 1e-3;
            rho = (chi - temp) / scale;
            if (failed) rho = -1.0; // the step of a failed factorisation is never accepted
            if (rho > 0 && std::isfinite(temp)) {
                double alpha = 1.0 - std::pow(2.0 * rho - 1.0, 3);
                alpha = std::min(alpha, o.lm_good_step_upper);
                ba->lambda *= std::max(o.lm_good_step_lower, alpha);
                ba->ni = 2.0;
                chi = temp;
                ba->cur ^= 1; // discardTop: the trial state becomes the estimate
            } else {
                ba->lambda *= ba->ni;
                ba->ni *= 2.0; // pop: estimate buffers untouched
                if (!std::isfinite(ba->lambda)) { stop_inf = true; break; }
            }
            ++q;
        } while (rho < 0 && q < o.lm_max_trials);
        ba->stats.lm_iterations++;
        ba->stats.lm_trials += (uint64_t)q;
        ++done;
        if (q == o.lm_max_trials || rho == 0 || stop_inf) break; // SolverResult::Terminate
    }
    if (performed) *performed = done;
    // a sweep speculated behind the last trial of a block that then terminated is never used (the next block starts from -1)
    if (ba->spec_lin_state >= 0) { ba->sweep_timer.drop_last(); ba->spec_lin_state = -1; }
    // the estimates stay on the device; the host copy is refreshed by the first call that reads it (ensure_host)
    ba->host_stale = true;
    ba->timer.collect();
    // the last trial's scalars have been read: the sweeps in front of them have completed (a record that has not stays pending)
    ba->sweep_timer.collect();
    return SVI_OK;
}

int add_proj(svi_ba* ba, int type, int64_t pose_id, int64_t lm_id, const double* z, const double* info, int robust)
{
    SVI_TRY(ensure_host(ba));
    auto ip = ba->pose_ix.find(pose_id);
    auto il = ba->lm_ix.find(lm_id);
    if (ip == ba->pose_ix.end()) return fail(SVI_ERR_NOT_FOUND, "pose id %lld not in graph", (long long)pose_id);
    if (il == ba->lm_ix.end()) return fail(SVI_ERR_NOT_FOUND, "landmark id %lld not in graph", (long long)lm_id);
    SVI_TRY(edges_push(ba, type, robust, ip->second, il->second, z, info));
    ba->initialized = false;
    return SVI_OK;
}

} // namespace

// ---- the edge store -------------------------------------------------------------------------------------------------------
int edges_push(svi_ba* ba, int type, int robust, int pose, int lm, const double* z, const double* info)
{
    EdgeStore& e = ba->proj;
    const size_t i = e.size();
    if (i == e.cap) { // the pinned array doubles; copies that may still read the old one are drained first
        const size_t cap = std::max<size_t>(2 * e.cap, 4096);
        double* nv = nullptr;
        SVI_HIP(svi::enter_device(ba->opt.device));
        SVI_HIP(hipHostMalloc(reinterpret_cast<void**>(&nv), cap * 72));
        if (e.vals) {
            SVI_HIP(hipStreamSynchronize(ba->stream));
            memcpy(nv, e.vals, i * 72);
            (void)hipHostFree(e.vals);
        }
        e.vals = nv; e.cap = cap;
    }
    e.pose.push_back(pose); e.lm.push_back(lm);
    e.flags.push_back((uint8_t)((type & 3) | (robust ? kFlagRobust : 0)));
    memcpy(e.vals + 9 * i, z, 24);
    memcpy(e.vals + 9 * i + 3, info, 48);
    if (info[1] != 0.0 || info[2] != 0.0 || info[4] != 0.0) e.n_offdiag++;
    ba->graph_version++;
    return SVI_OK;
}

void edges_truncate(svi_ba* ba, size_t n)
{
    EdgeStore& e = ba->proj;
    if (n >= e.size()) return;
    for (size_t i = n; i < e.size(); ++i) { const double* f = e.info(i); if (f[1] != 0.0 || f[2] != 0.0 || f[4] != 0.0) e.n_offdiag--; }
    e.pose.resize(n); e.lm.resize(n); e.flags.resize(n);
    ba->raw_uploaded = std::min(ba->raw_uploaded, n);
    ba->graph_version++;
}

// appends what has been added since the last flush to the device-side log (asynchronously, from pinned memory)
int edges_flush(svi_ba* ba)
{
    EdgeStore& e = ba->proj;
    const size_t E = e.size();
    if (ba->raw_uploaded >= E) return SVI_OK;
    SVI_HIP(svi::enter_device(ba->opt.device));
    if (E > ba->raw_cap) { // grow the device log: new buffers, the uploaded part moves device to device
        const size_t cap = std::max<size_t>({2 * ba->raw_cap, E, (size_t)4096});
        DevBuf nl, nf;
        SVI_TRY(nl.reserve(cap * 72));
        SVI_TRY(nf.reserve(cap));
        if (ba->raw_uploaded) {
            SVI_HIP(hipMemcpyAsync(nl.p, ba->raw_log.p, ba->raw_uploaded * 72, hipMemcpyDeviceToDevice, ba->stream));
            SVI_HIP(hipMemcpyAsync(nf.p, ba->raw_flags.p, ba->raw_uploaded, hipMemcpyDeviceToDevice, ba->stream));
            SVI_HIP(hipStreamSynchronize(ba->stream));
        }
        ba->raw_log.release(); ba->raw_flags.release();
        ba->raw_log = nl; ba->raw_flags = nf;
        ba->raw_cap = cap;
    }
    const size_t first = ba->raw_uploaded, cnt = E - first;
    SVI_HIP(hipMemcpyAsync(ba->raw_log.as<double>() + 9 * first, e.vals + 9 * first, cnt * 72, hipMemcpyHostToDevice, ba->stream));
    SVI_HIP(hipMemcpyAsync(ba->raw_flags.as<uint8_t>() + first, e.flags.data() + first, cnt, hipMemcpyHostToDevice, ba->stream));
    ba->raw_uploaded = E;
    return SVI_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

void svi_ba_options_default(svi_ba_options* o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->fx = o->fy = 1.0;
    o->cauchy_delta = 1.0;
    o->lm_tau = 1e-5; o->lm_good_step_lower = 1.0 / 3.0; o->lm_good_step_upper = 2.0 / 3.0; o->lm_max_trials = 10;
    o->max_depth_xyz_l2 = 10.0; o->max_depth_uvdepth_l2 = 50.0; o->max_depth_uvdisp_l2 = 10000.0; o->sane_position_l2 = 1e12;
    o->n_ranks = 1;
    o->chol_tile = 48;
    o->chol_order = 0;
}

int svi_ba_create(const svi_ba_options* o, svi_ba** out)
{
    if (!o || !out) return fail(SVI_ERR_INVALID, "svi_ba_create: null argument");
    *out = nullptr;
    if (o->n_ranks < 1 || o->rank < 0 || o->rank >= o->n_ranks) return fail(SVI_ERR_INVALID, "bad rank %d / n_ranks %d", o->rank, o->n_ranks);
    if (!(o->cauchy_delta > 0) || o->lm_max_trials < 1) return fail(SVI_ERR_INVALID, "bad LM / kernel options");
    if (int rc = use_device(o->device)) return rc;
    svi_ba* ba = new svi_ba();
    ba->opt = *o;
    if (ba->opt.chol_tile == 0) ba->opt.chol_tile = 48;
    if (o->stream) ba->stream = static_cast<hipStream_t>(o->stream);
    else {
        // (the highest priority there is: the factorisation's short launches run on this stream beside the Schur stream's one long kernel)
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess ? hipStreamCreateWithPriority(&ba->stream, hipStreamNonBlocking, hi)
                                                                                 : hipStreamCreateWithFlags(&ba->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ba; return fail(SVI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        ba->own_stream = true;
    }
    ba->timer.on = o->profile != 0;
    ba->sweep_timer.on = o->sweep_events != 0 && o->profile == 0;
    {   // the second stream and the memory values that tie it to the first (see ba_host.h); without them trials run on one stream
        int can = 0;
        bool ok = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ba->opt.device) == hipSuccess && can != 0;
        ok = ok && hipStreamCreateWithFlags(&ba->stream_schur, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipExtMallocWithFlags(reinterpret_cast<void**>(&ba->sig_lin), 8, hipMallocSignalMemory) == hipSuccess;
        for (int k = 0; k < kMaxStages && ok; ++k) ok = hipExtMallocWithFlags(reinterpret_cast<void**>(&ba->sig_stage[k]), 8, hipMallocSignalMemory) == hipSuccess;
        if (ok) {
            *ba->sig_lin = 0;
            for (int k = 0; k < kMaxStages; ++k) *ba->sig_stage[k] = 0;
        } else (void)hipGetLastError();
        ba->overlap_ok = ok;
    }
    *out = ba;
    return SVI_OK;
}

int svi_ba_destroy(svi_ba* ba)
{
    if (!ba) return SVI_OK;
    (void)hipSetDevice(ba->opt.device);
    (void)hipStreamSynchronize(ba->stream);
    if (ba->stream_schur) { (void)hipStreamSynchronize(ba->stream_schur); (void)hipStreamDestroy(ba->stream_schur); ba->stream_schur = nullptr; }
    if (ba->sig_lin) { (void)hipFree(ba->sig_lin); ba->sig_lin = nullptr; }
    for (int k = 0; k < kMaxStages; ++k) if (ba->sig_stage[k]) { (void)hipFree(ba->sig_stage[k]); ba->sig_stage[k] = nullptr; }
    free_device(ba);
    if (ba->proj.vals) { (void)hipHostFree(ba->proj.vals); ba->proj.vals = nullptr; }
    ba->timer.release();
    ba->sweep_timer.release();
    if (ba->own_stream) (void)hipStreamDestroy(ba->stream);
    delete ba;
    return SVI_OK;
}

int svi_ba_add_pose(svi_ba* ba, int64_t id, const double T[12], int fixed)
{
    if (!ba || !T) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    if (ba->pose_ix.count(id) || ba->lm_ix.count(id)) return fail(SVI_ERR_INVALID, "vertex id %lld already in graph", (long long)id);
    HPose p{};
    p.id = id; memcpy(p.T, T, 96); p.fixed = fixed ? 1 : 0;
    ba->pose_ix[id] = (int)ba->poses.size();
    ba->poses.push_back(p);
    ba->initialized = false;
    return SVI_OK;
}

int svi_ba_add_landmark(svi_ba* ba, int64_t id, const double p[3], int fixed)
{
    if (!ba || !p) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    if (ba->pose_ix.count(id) || ba->lm_ix.count(id)) return fail(SVI_ERR_INVALID, "vertex id %lld already in graph", (long long)id);
    HLm l{};
    l.id = id; memcpy(l.p, p, 24); l.fixed = fixed ? 1 : 0;
    ba->lm_ix[id] = (int)ba->lms.size();
    ba->lms.push_back(l);
    ba->initialized = false;
    return SVI_OK;
}

int svi_ba_add_edge_xyz(svi_ba* ba, int64_t pose_id, int64_t lm_id, const double z[3], const double info[6], int robust)
{
    if (!ba || !z || !info) return fail(SVI_ERR_INVALID, "null argument");
    return add_proj(ba, kTypeXYZ, pose_id, lm_id, z, info, robust);
}
int svi_ba_add_edge_depth(svi_ba* ba, int64_t pose_id, int64_t lm_id, const double z[3], const double info[6], int robust)
{
    if (!ba || !z || !info) return fail(SVI_ERR_INVALID, "null argument");
    return add_proj(ba, kTypeDepth, pose_id, lm_id, z, info, robust);
}
int svi_ba_add_edge_disparity(svi_ba* ba, int64_t pose_id, int64_t lm_id, const double z[3], const double info[6], int robust)
{
    if (!ba || !z || !info) return fail(SVI_ERR_INVALID, "null argument");
    return add_proj(ba, kTypeDisparity, pose_id, lm_id, z, info, robust);
}

int svi_ba_add_edges_bulk(svi_ba* ba, int64_t n, const int32_t* type, const int64_t* pose_id, const int64_t* lm_id,
                          const double* z, const double* info, const int32_t* robust)
{
    if (!ba || n < 0) return fail(SVI_ERR_INVALID, "bad argument");
    SVI_TRY(ensure_host(ba));
    if (n == 0) return SVI_OK;
    if (!type || !pose_id || !lm_id || !z || !info) return fail(SVI_ERR_INVALID, "null argument");
    const size_t before = ba->proj.size();
    ba->proj.pose.reserve(before + (size_t)n); ba->proj.lm.reserve(before + (size_t)n); ba->proj.flags.reserve(before + (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (type[i] < 0 || type[i] > 2) { edges_truncate(ba, before); return fail(SVI_ERR_INVALID, "edge %lld: unknown type %d", (long long)i, type[i]); }
        const int rc = add_proj(ba, type[i], pose_id[i], lm_id[i], z + 3 * i, info + 6 * i, robust ? robust[i] : 1);
        if (rc != SVI_OK) { edges_truncate(ba, before); return rc; }
    }
    return edges_flush(ba);
}

int svi_ba_add_edge_se3(svi_ba* ba, int64_t id_i, int64_t id_j, const double Z[12], const double info[21], int robust)
{
    if (!ba || !Z || !info) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    auto i = ba->pose_ix.find(id_i), j = ba->pose_ix.find(id_j);
    if (i == ba->pose_ix.end() || j == ba->pose_ix.end()) return fail(SVI_ERR_NOT_FOUND, "pose id not in graph");
    if (i->second == j->second) return fail(SVI_ERR_INVALID, "EdgeSE3 between a pose and itself");
    HSe3 e{};
    e.i = i->second; e.j = j->second; e.robust = robust ? 1 : 0;
    memcpy(e.Z, Z, 96); memcpy(e.info, info, 21 * 8);
    ba->se3.push_back(e);
    ba->initialized = false;
    return SVI_OK;
}

int svi_ba_add_edge_accel(svi_ba* ba, int64_t pose_id, const double a[3], const double off[12], const double info[6])
{
    if (!ba || !a || !info) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    auto i = ba->pose_ix.find(pose_id);
    if (i == ba->pose_ix.end()) return fail(SVI_ERR_NOT_FOUND, "pose id %lld not in graph", (long long)pose_id);
    HAcc e{};
    e.pose = i->second;
    memcpy(e.a, a, 24);
    memcpy(e.off, off ? off : kIdentity12, 96);
    memcpy(e.info, info, 48);
    ba->acc.push_back(e);
    ba->initialized = false;
    return SVI_OK;
}

int svi_ba_add_edge_lm_lm(svi_ba* ba, int64_t id_i, int64_t id_j, const double z[3], const double info[6], int robust)
{
    if (!ba || !z || !info) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    auto i = ba->lm_ix.find(id_i), j = ba->lm_ix.find(id_j);
    if (i == ba->lm_ix.end() || j == ba->lm_ix.end()) return fail(SVI_ERR_NOT_FOUND, "landmark id not in graph");
    if (i->second == j->second) return fail(SVI_ERR_INVALID, "EdgePointXYZ between a landmark and itself");
    HLL e{};
    e.i = i->second; e.j = j->second; e.robust = robust ? 1 : 0;
    memcpy(e.z, z, 24); memcpy(e.info, info, 48);
    ba->lmlm.push_back(e);
    ba->initialized = false;
    return SVI_OK;
}

// Cg2oOptimizer::_setAndgetPose (:1229-1290) + gravity edge (:480, :982-997)
int svi_ba_add_keyframe(svi_ba* ba, int64_t id, int64_t from_id, const double T[12], const double shift[3], const double accel[3])
{
    if (!ba || !T) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    auto f = ba->pose_ix.find(from_id);
    if (f == ba->pose_ix.end()) return fail(SVI_ERR_NOT_FOUND, "previous keyframe id %lld not in graph", (long long)from_id);
    double X[12];
    memcpy(X, T, 96);
    if (shift) { X[9] += shift[0]; X[10] += shift[1]; X[11] += shift[2]; } // :1233
    SVI_TRY(svi_ba_add_pose(ba, id, X, 0));
    const HPose& pf = ba->poses[f->second];
    // measurement = Xfrom^-1 * Xcur (:1250)
    double Z[12];
    mat3T_mul(pf.T, X, Z);
    const double d[3] = {X[9] - pf.T[9], X[10] - pf.T[10], X[11] - pf.T[11]};
    for (int c = 0; c < 3; ++c) Z[9 + c] = pf.T[c] * d[0] + pf.T[3 + c] * d[1] + pf.T[6 + c] * d[2];
    const double s = 1.0 / (1.0 + (Z[9] * Z[9] + Z[10] * Z[10] + Z[11] * Z[11])); // :1259
    double info[21] = {0};
    info[0] = info[6] = info[11] = 100000.0 * s; // m_matInformationPose (Cg2oOptimizer.cpp:72) scaled on the translation block (:1262-1263)
    info[15] = info[18] = info[20] = 100000.0;
    SVI_TRY(svi_ba_add_edge_se3(ba, from_id, id, Z, info, 0));
    const double a0[3] = {0, 0, 0}, I3[6] = {1, 0, 0, 1, 0, 1};
    return svi_ba_add_edge_accel(ba, id, accel ? accel : a0, ba->imu_off, I3);
}

// the offset parameter every gravity edge of svi_ba_add_keyframe refers to (Cg2oOptimizer.cpp:213, :988)
int svi_ba_set_imu_offset(svi_ba* ba, const double off[12])
{
    if (!ba || !off) return fail(SVI_ERR_INVALID, "null argument");
    memcpy(ba->imu_off, off, 96);
    return SVI_OK;
}

// Cg2oOptimizer::_setLandmarkMeasurementsWORLD (:1383-1466) with the factories (:999-1073)
int svi_ba_add_measurements(svi_ba* ba, int64_t pose_id, int64_t n, const int64_t* lm_id, const float* uvL, const float* uvR,
                            const double* xyz, int64_t stored[3])
{
    if (!ba || n < 0) return fail(SVI_ERR_INVALID, "bad argument");
    SVI_TRY(ensure_host(ba));
    if (stored) stored[0] = stored[1] = stored[2] = 0;
    if (n == 0) return SVI_OK;
    if (!lm_id || !uvL || !uvR || !xyz) return fail(SVI_ERR_INVALID, "null argument");
    auto ip = ba->pose_ix.find(pose_id);
    if (ip == ba->pose_ix.end()) return fail(SVI_ERR_NOT_FOUND, "pose id %lld not in graph", (long long)pose_id);
    const svi_ba_options& o = ba->opt;
    int64_t cnt[3] = {0, 0, 0};
    for (int64_t m = 0; m < n; ++m) {
        auto il = ba->lm_ix.find(lm_id[m]);
        if (il == ba->lm_ix.end()) continue; // not in graph: silently skipped (:1393-1396)
        const HPose& P = ba->poses[ip->second];
        const HLm& L = ba->lms[il->second];
        const double* pm = xyz + 3 * m;
        double pe[3];
        to_camera(P.T, P.T + 9, L.p, pe);
        const double l2abs = pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2];
        const double l2rel = (pe[0] * pe[0] + pe[1] * pe[1] + pe[2] * pe[2]) / l2abs;
        if (!(0.75 < l2rel && 1.25 > l2rel)) continue; // :1409
        const double w = 1.0 / pm[2];                  // :1412
        if (o.max_depth_xyz_l2 > l2abs) {              // :1415
            const double info[6] = {w * 1000, 0, 0, w * 1000, 0, w * 1000};
            SVI_TRY(add_proj(ba, kTypeXYZ, pose_id, lm_id[m], pm, info, 1));
            cnt[0]++;
        } else if (o.max_depth_uvdepth_l2 > l2abs) {   // :1426
            const double z[3] = {(double)uvL[2 * m], (double)uvL[2 * m + 1], pm[2]};
            const double info[6] = {w, 0, 0, w, 0, w * 100};
            SVI_TRY(add_proj(ba, kTypeDepth, pose_id, lm_id[m], z, info, 1));
            cnt[1]++;
        } else if (o.max_depth_uvdisp_l2 > l2abs) {    // :1437
            const double disp = (double)(uvL[2 * m] - uvR[2 * m]); // float difference, promoted (:1440)
            if (1.0 < disp) {                          // :1443
                const double z[3] = {(double)uvL[2 * m], (double)uvL[2 * m + 1], disp / (o.fx * o.baseline_m)}; // :1058
                const double info[6] = {w, 0, 0, w, 0, w * 1000};
                SVI_TRY(add_proj(ba, kTypeDisparity, pose_id, lm_id[m], z, info, 1));
                cnt[2]++;
            }
        }
    }
    if (stored) { stored[0] = cnt[0]; stored[1] = cnt[1]; stored[2] = cnt[2]; }
    return edges_flush(ba); // the new edges' values travel to the device-side log while the caller goes on building the graph
}

int svi_ba_initialize(svi_ba* ba)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    SVI_TRY(ensure_host(ba));
    if (int rc = use_device(ba->opt.device)) return rc;
    SVI_HIP(hipStreamSynchronize(ba->stream)); // nothing may still read the buffers that are about to be refilled
    if (ba->stream_schur) SVI_HIP(hipStreamSynchronize(ba->stream_schur));
    ba->cur = 0;
    ba->have_chi = false;
    ba->hinv_valid = false;          // (hand-overs between a linearisation and its trial: none is pending across an initialize)
    ba->lin_post_deferred = false;
    ba->spec_lin_state = -1;         // (nor a speculated sweep: the estimates may have been edited)
    // every edit of the graph's structure clears `initialized`: if it is still set, the device structures are those of this very
    // graph and only the estimates have to go back (g2o rebuilds everything per call; the result is the same)
    if (ba->initialized) return reupload_state(ba);
    invalidate_device(ba);
    const int rc = build_structure(ba);
    if (rc != SVI_OK) { invalidate_device(ba); return rc; }
    ba->initialized = true;
    return SVI_OK;
}

int svi_ba_optimize(svi_ba* ba, int iterations, int* performed)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    if (performed) *performed = 0;
    if (iterations < 0) return fail(SVI_ERR_INVALID, "negative iteration count");
    return optimize_block(ba, iterations, performed);
}

int svi_ba_optimize_until(svi_ba* ba, double ratio, int first, int block, uint64_t* nominal, uint64_t* executed)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    if (first < 1 || block < 1) return fail(SVI_ERR_INVALID, "first and block must be >= 1");
    uint64_t nom = 0, exe = 0;
    int r = 0;
    SVI_TRY(optimize_block(ba, first, &r)); // :960
    nom += (uint64_t)first; exe += (uint64_t)r;
    double prev = 1.1 * ba->last_plain;         // :966
    while (ratio > ba->last_plain / prev) {     // :969
        prev = ba->last_plain;                  // :972
        SVI_TRY(optimize_block(ba, block, &r)); // :975
        nom += (uint64_t)block; exe += (uint64_t)r;
    }
    if (nominal) *nominal = nom;
    if (executed) *executed = exe;
    return SVI_OK;
}

int svi_ba_chi2(svi_ba* ba, double* plain, double* robust)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    if (!ba->initialized) return fail(SVI_ERR_STATE, "svi_ba_chi2 before svi_ba_initialize");
    if (!ba->have_chi) { // nothing evaluated yet: evaluate the current estimate
        SVI_HIP(svi::enter_device(ba->opt.device));
        ba_chi2_only(ba->d, ba->cur, ba->stream);
        SVI_TRY(reduce_and_read_trial(ba, 8, ba->cur)); // (the pose-only edges are evaluated by the reduction)
        ba->last_robust = ba->h_scal[0]; ba->last_plain = ba->h_scal[1]; ba->have_chi = true;
    }
    if (plain) *plain = ba->last_plain;
    if (robust) *robust = ba->last_robust;
    return SVI_OK;
}

int svi_ba_sync_host(svi_ba* ba)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    return ensure_host(ba);
}

int svi_ba_lambda(svi_ba* ba, double* lambda)
{
    if (!ba || !lambda) return fail(SVI_ERR_INVALID, "null argument");
    *lambda = ba->lambda;
    return SVI_OK;
}

int svi_ba_get_pose(svi_ba* ba, int64_t id, double T[12])
{
    if (!ba || !T) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    auto i = ba->pose_ix.find(id);
    if (i == ba->pose_ix.end()) return fail(SVI_ERR_NOT_FOUND, "pose id %lld not in graph", (long long)id);
    memcpy(T, ba->poses[i->second].T, 96);
    return SVI_OK;
}

int svi_ba_get_landmark(svi_ba* ba, int64_t id, double p[3])
{
    if (!ba || !p) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    auto i = ba->lm_ix.find(id);
    if (i == ba->lm_ix.end()) return fail(SVI_ERR_NOT_FOUND, "landmark id %lld not in graph", (long long)id);
    memcpy(p, ba->lms[i->second].p, 24);
    return SVI_OK;
}

int svi_ba_num_poses(svi_ba* ba, int64_t* n) { if (!ba || !n) return fail(SVI_ERR_INVALID, "null argument"); *n = (int64_t)ba->poses.size(); return SVI_OK; }
int svi_ba_num_landmarks(svi_ba* ba, int64_t* n) { if (!ba || !n) return fail(SVI_ERR_INVALID, "null argument"); *n = (int64_t)ba->lms.size(); return SVI_OK; }
int svi_ba_num_edges(svi_ba* ba, int64_t* n)
{
    if (!ba || !n) return fail(SVI_ERR_INVALID, "null argument");
    *n = (int64_t)(ba->proj.size() + ba->se3.size() + ba->acc.size() + ba->lmlm.size());
    return SVI_OK;
}

int svi_ba_get_poses(svi_ba* ba, int64_t* ids, double* T)
{
    if (!ba || !T) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    std::vector<int> ord(ba->poses.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return ba->poses[a].id < ba->poses[b].id; });
    for (size_t k = 0; k < ord.size(); ++k) { if (ids) ids[k] = ba->poses[ord[k]].id; memcpy(T + 12 * k, ba->poses[ord[k]].T, 96); }
    return SVI_OK;
}

int svi_ba_get_landmarks(svi_ba* ba, int64_t* ids, double* p)
{
    if (!ba || !p) return fail(SVI_ERR_INVALID, "null argument");
    SVI_TRY(ensure_host(ba));
    std::vector<int> ord(ba->lms.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return ba->lms[a].id < ba->lms[b].id; });
    for (size_t k = 0; k < ord.size(); ++k) { if (ids) ids[k] = ba->lms[ord[k]].id; memcpy(p + 3 * k, ba->lms[ord[k]].p, 24); }
    return SVI_OK;
}

// _applyOptimizationToLandmarks (:1486-1504): drop landmarks whose squared norm is not below 1e12
int svi_ba_prune_diverged(svi_ba* ba, int64_t* removed)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    SVI_TRY(ensure_host(ba));
    const int nl = (int)ba->lms.size();
    std::vector<int> remap(nl, -1);
    int64_t gone = 0;
    int k = 0;
    for (int i = 0; i < nl; ++i) {
        const double* p = ba->lms[i].p;
        if (ba->opt.sane_position_l2 > p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) remap[i] = k++;
        else ++gone;
    }
    if (gone) {
        std::vector<HLm> keep;
        keep.reserve(k);
        ba->lm_ix.clear();
        for (int i = 0; i < nl; ++i) if (remap[i] >= 0) { ba->lm_ix[ba->lms[i].id] = (int)keep.size(); keep.push_back(ba->lms[i]); }
        ba->lms.swap(keep);
        {   // compact the edge store in place (the device-side log is in insertion order: refilled by the next flush)
            EdgeStore& e = ba->proj;
            SVI_HIP(hipStreamSynchronize(ba->stream)); // an upload may still read the pinned values
            size_t w = 0;
            e.n_offdiag = 0;
            for (size_t i = 0; i < e.size(); ++i) {
                if (remap[e.lm[i]] < 0) continue;
                e.pose[w] = e.pose[i]; e.lm[w] = remap[e.lm[i]]; e.flags[w] = e.flags[i];
                if (w != i) memmove(e.vals + 9 * w, e.vals + 9 * i, 72);
                const double* f = e.info(w);
                if (f[1] != 0.0 || f[2] != 0.0 || f[4] != 0.0) e.n_offdiag++;
                ++w;
            }
            e.pose.resize(w); e.lm.resize(w); e.flags.resize(w);
            ba->raw_uploaded = 0;
        }
        ba->graph_version++;
        std::vector<HLL> le;
        for (HLL e : ba->lmlm) if (remap[e.i] >= 0 && remap[e.j] >= 0) { e.i = remap[e.i]; e.j = remap[e.j]; le.push_back(e); }
        ba->lmlm.swap(le);
        ba->initialized = false;
    }
    if (removed) *removed = gone;
    return SVI_OK;
}

// _applyOptimizationToLandmarks + _applyOptimizationToKeyFrames (Cg2oOptimizer.cpp:1468-1540)
int svi_ba_apply_optimization(svi_ba* ba, const double shift[3], int64_t* lm_ids, double* lm_xyz, uint8_t* lm_kept, int64_t* kf_ids,
                              double* kf_T, int64_t* erased)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    SVI_TRY(ensure_host(ba));
    const double s[3] = {shift ? shift[0] : 0.0, shift ? shift[1] : 0.0, shift ? shift[2] : 0.0};
    std::vector<int> ord(ba->lms.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return ba->lms[a].id < ba->lms[b].id; });
    for (size_t k = 0; k < ord.size(); ++k) {
        const HLm& l = ba->lms[ord[k]];
        const bool sane = ba->opt.sane_position_l2 > l.p[0] * l.p[0] + l.p[1] * l.p[1] + l.p[2] * l.p[2];
        if (lm_ids) lm_ids[k] = l.id;
        if (lm_kept) lm_kept[k] = sane ? 1 : 0;
        if (lm_xyz) for (int c = 0; c < 3; ++c) lm_xyz[3 * k + c] = sane ? l.p[c] - s[c] : 0.0;
    }
    std::vector<int> po(ba->poses.size());
    std::iota(po.begin(), po.end(), 0);
    std::sort(po.begin(), po.end(), [&](int a, int b) { return ba->poses[a].id < ba->poses[b].id; });
    for (size_t k = 0; k < po.size(); ++k) {
        const HPose& q = ba->poses[po[k]];
        if (kf_ids) kf_ids[k] = q.id;
        if (kf_T) {
            memcpy(kf_T + 12 * k, q.T, 9 * sizeof(double));
            for (int c = 0; c < 3; ++c) kf_T[12 * k + 9 + c] = q.T[9 + c] - s[c];
        }
    }
    return svi_ba_prune_diverged(ba, erased);
}

int svi_ba_set_allreduce(svi_ba* ba, svi_allreduce_fn fn, void* user)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    ba->ar = fn; ba->ar_user = user;
    return SVI_OK;
}

int svi_ba_get_phase_times(svi_ba* ba, double ms[SVI_PH_COUNT], int64_t calls[SVI_PH_COUNT])
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    for (int i = 0; i < SVI_PH_COUNT; ++i) { if (ms) ms[i] = ba->timer.ms[i]; if (calls) calls[i] = ba->timer.calls[i]; }
    return SVI_OK;
}

int svi_ba_reset_phase_times(svi_ba* ba)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    for (int i = 0; i < SVI_PH_COUNT; ++i) { ba->timer.ms[i] = 0.0; ba->timer.calls[i] = 0; }
    ba->sweep_timer.ms[0] = 0.0; ba->sweep_timer.calls[0] = 0;
    return SVI_OK;
}

int svi_ba_get_sweep_time(svi_ba* ba, double* ms_total, int64_t* calls)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    const bool prof = ba->timer.on;
    if (ms_total) *ms_total = prof ? ba->timer.ms[SVI_PH_LINEARIZE_LM] + ba->timer.ms[SVI_PH_LINEARIZE_POSE] : ba->sweep_timer.ms[0];
    if (calls) *calls = prof ? ba->timer.calls[SVI_PH_LINEARIZE_LM] : ba->sweep_timer.calls[0];
    return SVI_OK;
}

int svi_ba_get_stats(svi_ba* ba, svi_ba_stats* s)
{
    if (!ba || !s) return fail(SVI_ERR_INVALID, "null argument");
    *s = ba->stats;
    s->n_poses = (int64_t)ba->poses.size();
    s->n_landmarks = (int64_t)ba->lms.size();
    s->n_edges_proj = (int64_t)ba->proj.size();
    s->n_edges_se3 = (int64_t)ba->se3.size(); s->n_edges_accel = (int64_t)ba->acc.size(); s->n_edges_lmlm = (int64_t)ba->lmlm.size();
    return SVI_OK;
}

int svi_ba_debug_edge_jacobians(svi_ba* ba, double* err, double* J_pose, double* J_lm)
{
    if (!ba || !err || !J_pose || !J_lm) return fail(SVI_ERR_INVALID, "null argument");
    if (!ba->initialized) return fail(SVI_ERR_STATE, "debug tap before svi_ba_initialize");
    SVI_HIP(svi::enter_device(ba->opt.device));
    const size_t E = ba->proj.size();
    double *de = nullptr, *dp = nullptr, *dl = nullptr;
    SVI_HIP(hipMalloc(reinterpret_cast<void**>(&de), sizeof(double) * 30 * std::max<size_t>(E, 1)));
    dp = de + 3 * E; dl = dp + 18 * E;
    hipError_t e1 = hipMemsetAsync(de, 0, sizeof(double) * 30 * std::max<size_t>(E, 1), ba->stream);
    ba_debug_jacobians(ba->d, ba->cur, ba->e_orig, de, dp, dl, ba->stream);
    hipError_t e2 = hipMemcpyAsync(err, de, sizeof(double) * 3 * E, hipMemcpyDeviceToHost, ba->stream);
    hipError_t e3 = hipMemcpyAsync(J_pose, dp, sizeof(double) * 18 * E, hipMemcpyDeviceToHost, ba->stream);
    hipError_t e4 = hipMemcpyAsync(J_lm, dl, sizeof(double) * 9 * E, hipMemcpyDeviceToHost, ba->stream);
    hipError_t e5 = hipStreamSynchronize(ba->stream);
    (void)hipFree(de);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess)
        return fail(SVI_ERR_HIP, "debug_edge_jacobians: HIP failure");
    return SVI_OK;
}

int svi_ba_debug_aux_jacobians(svi_ba* ba, double* se3_err, double* se3_Ji, double* se3_Jj, double* acc_err, double* acc_J)
{
    if (!ba || !se3_err || !se3_Ji || !se3_Jj || !acc_err || !acc_J) return fail(SVI_ERR_INVALID, "null argument");
    if (!ba->initialized) return fail(SVI_ERR_STATE, "debug tap before svi_ba_initialize");
    if (ba->opt.rank != 0) return fail(SVI_ERR_STATE, "pose-only edges live on rank 0");
    SVI_HIP(svi::enter_device(ba->opt.device));
    const size_t ns = ba->se3.size(), na = ba->acc.size(), total = 78 * ns + 21 * na;
    double* dev = nullptr;
    SVI_HIP(hipMalloc(reinterpret_cast<void**>(&dev), sizeof(double) * std::max<size_t>(total, 1)));
    double *d_se = dev, *d_si = d_se + 6 * ns, *d_sj = d_si + 36 * ns, *d_ae = d_sj + 36 * ns, *d_aj = d_ae + 3 * na;
    ba_debug_aux_jacobians(ba->d, ba->cur, d_se, d_si, d_sj, d_ae, d_aj, ba->stream);
    std::vector<double> h(std::max<size_t>(total, 1));
    hipError_t e1 = hipMemcpyAsync(h.data(), dev, sizeof(double) * total, hipMemcpyDeviceToHost, ba->stream);
    hipError_t e2 = hipStreamSynchronize(ba->stream);
    (void)hipFree(dev);
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(SVI_ERR_HIP, "debug_aux_jacobians: HIP failure");
    memcpy(se3_err, h.data(), sizeof(double) * 6 * ns);
    memcpy(se3_Ji, h.data() + 6 * ns, sizeof(double) * 36 * ns);
    memcpy(se3_Jj, h.data() + 42 * ns, sizeof(double) * 36 * ns);
    memcpy(acc_err, h.data() + 78 * ns, sizeof(double) * 3 * na);
    memcpy(acc_J, h.data() + 78 * ns + 3 * na, sizeof(double) * 18 * na);
    return SVI_OK;
}

int svi_debug_set_backsolve_spin_limit(int polls)
{
    if (polls < 1) return fail(SVI_ERR_INVALID, "spin limit must be >= 1");
    g_backsolve_spin_limit.store(polls, std::memory_order_relaxed);
    return SVI_OK;
}

int svi_debug_chol_probe(int device, int tile, int reps, int stop_after, double* ms)
{
    if (!ms || reps < 1 || (tile != 48 && tile != 96)) return fail(SVI_ERR_INVALID, "bad probe argument");
    if (int rc = use_device(device)) return rc;
    double out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (chol_potrf_probe(tile, reps, stop_after, out)) return fail(SVI_ERR_HIP, "probe failed");
    // stop_after 6..9: shader cycles of the pivot sweep in ms[0], 100 MHz ticks in ms[1], per-section cycle sums in
    // ms[3..5] (the caller passes room for 8 doubles); otherwise the mean kernel time in ms[0]
    ms[0] = out[0];
    if (stop_after >= 6 && stop_after <= 9) for (int q = 1; q < 6; ++q) ms[q] = out[q];
    return SVI_OK;
}

// mean duration of the Jacobian sweep (K2 + K3): `reps` back-to-back sweeps on the handle's stream between two HIP events
// which: 0 = the whole sweep (K2 then K3, back to back), 1 = K2 alone, 2 = K3 alone
static int time_sweep_impl(svi_ba* ba, int reps, int which, double* ms_avg)
{
    if (!ba || !ms_avg || reps < 1) return fail(SVI_ERR_INVALID, "bad argument");
    if (!ba->initialized) return fail(SVI_ERR_STATE, "debug tap before svi_ba_initialize");
    SVI_HIP(svi::enter_device(ba->opt.device));
    hipEvent_t a, b;
    SVI_HIP(hipEventCreate(&a));
    SVI_HIP(hipEventCreate(&b));
    auto once = [&]() {
        if (which != 2) ba_linearize_lm(ba->d, ba->cur, ba->stream);
        if (which != 1) ba_linearize_pose(ba->d, ba->cur, ba->stream);
    };
    for (int i = 0; i < 3; ++i) once();
    SVI_HIP(hipEventRecord(a, ba->stream));
    for (int i = 0; i < reps; ++i) once();
    SVI_HIP(hipEventRecord(b, ba->stream));
    SVI_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    SVI_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_avg = (double)ms / reps;
    return SVI_OK;
}
int svi_ba_debug_time_sweep(svi_ba* ba, int reps, double* ms_avg) { return time_sweep_impl(ba, reps, 0, ms_avg); }

int svi_ba_debug_time_sweep_cold(svi_ba* ba, int reps, size_t evict_bytes, double* ms_avg)
{
    if (!ba || !ms_avg || reps < 1 || evict_bytes == 0) return fail(SVI_ERR_INVALID, "bad argument");
    if (!ba->initialized) return fail(SVI_ERR_STATE, "debug tap before svi_ba_initialize");
    SVI_HIP(svi::enter_device(ba->opt.device));
    void* scratch = nullptr;
    SVI_HIP(hipMalloc(&scratch, evict_bytes));
    std::vector<hipEvent_t> ev((size_t)2 * reps);
    for (auto& e : ev) SVI_HIP(hipEventCreate(&e));
    for (int i = 0; i < reps; ++i) {
        SVI_HIP(hipMemsetAsync(scratch, i & 0xFF, evict_bytes, ba->stream)); // pushes the sweep's operands out of L2 and the Infinity Cache
        SVI_HIP(hipEventRecord(ev[2 * i], ba->stream));
        ba_linearize_lm(ba->d, ba->cur, ba->stream);
        ba_linearize_pose(ba->d, ba->cur, ba->stream);
        SVI_HIP(hipEventRecord(ev[2 * i + 1], ba->stream));
    }
    SVI_HIP(hipStreamSynchronize(ba->stream));
    double total = 0.0;
    for (int i = 0; i < reps; ++i) { float t = 0.f; SVI_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1])); total += t; }
    for (auto& e : ev) (void)hipEventDestroy(e);
    (void)hipFree(scratch);
    *ms_avg = total / reps;
    return SVI_OK;
}
int svi_ba_debug_time_sweep_part(svi_ba* ba, int reps, int which, double* ms_avg)
{
    if (which != 1 && which != 2) return fail(SVI_ERR_INVALID, "which must be 1 (K2) or 2 (K3)");
    return time_sweep_impl(ba, reps, which, ms_avg);
}

// linearise at the current estimate, reduce with damping lambda, return dense S (with lambda on its
// diagonal) and g
int svi_ba_debug_reduced_system(svi_ba* ba, double lambda, double* S, double* g, int64_t cap, int64_t* n_out)
{
    if (!ba || !S || !g || !n_out) return fail(SVI_ERR_INVALID, "null argument");
    if (!ba->initialized) return fail(SVI_ERR_STATE, "debug tap before svi_ba_initialize");
    SVI_HIP(svi::enter_device(ba->opt.device));
    BaDev& d = ba->d;
    const int64_t n = 6 * (int64_t)d.Pf;
    *n_out = n;
    if (cap < n) return fail(SVI_ERR_INVALID, "capacity %lld < n %lld", (long long)cap, (long long)n);
    ba->spec_lin_state = -1; // (a block that ended on a rejected speculated trial leaves the buffers of ANOTHER state's sweep behind)
    SVI_TRY(linearize(ba));
    SVI_HIP(hipMemsetAsync(d.chol_status, 0, sizeof(int), ba->stream));
    set_assembly_flags(ba);
    SVI_TRY(zero_reduced_system(ba));
    ba_invert_landmarks(d, lambda, ba->stream);
    ba_schur(d, nullptr, ++ba->stage_seq, 0, 0, ba->stream);
    assemble(ba);
    SVI_HIP(hipGetLastError());
    SVI_TRY(allreduce(ba, d.g, (size_t)d.red_count));
    const int TS = d.TS, NT = d.NT;
    std::vector<double> tiles((size_t)d.n_tiles * TS * TS), gv((size_t)NT * TS);
    std::vector<int> tmap((size_t)NT * NT);
    if (d.n_tiles) SVI_HIP(hipMemcpyAsync(tiles.data(), d.S, sizeof(double) * tiles.size(), hipMemcpyDeviceToHost, ba->stream));
    if (NT) SVI_HIP(hipMemcpyAsync(gv.data(), d.g, sizeof(double) * gv.size(), hipMemcpyDeviceToHost, ba->stream));
    if (NT) SVI_HIP(hipMemcpyAsync(tmap.data(), d.tile_map, sizeof(int) * tmap.size(), hipMemcpyDeviceToHost, ba->stream));
    SVI_HIP(hipMemsetAsync(d.chol_status, 0, sizeof(int), ba->stream)); // the tap does not read it: clean for the next trial
    SVI_HIP(hipStreamSynchronize(ba->stream));
    // the tap reports the system in NATURAL free-pose order (ascending id), whatever elimination order is in use
    std::vector<int64_t> nat(n);
    for (int64_t a = 0; a < (int64_t)d.Pf; ++a)
        for (int c = 0; c < 6; ++c) nat[6 * (int64_t)ba->red_perm[a] + c] = 6 * a + c;
    for (int64_t i = 0; i < n * n; ++i) S[i] = 0.0;
    for (int ti = 0; ti < NT; ++ti)
        for (int tj = 0; tj <= ti; ++tj) {
            const int t = tmap[(size_t)ti * NT + tj];
            if (t < 0) continue;
            for (int r = 0; r < TS; ++r)
                for (int c = 0; c < TS; ++c) {
                    const int64_t R = (int64_t)ti * TS + r, Cc = (int64_t)tj * TS + c;
                    if (R >= n || Cc >= n || Cc > R) continue;
                    const double v = tiles[(size_t)t * TS * TS + (size_t)r * TS + c];
                    S[nat[R] * n + nat[Cc]] = v; S[nat[Cc] * n + nat[R]] = v;
                }
        }
    for (int64_t i = 0; i < n; ++i) { S[nat[i] * n + nat[i]] += lambda; g[nat[i]] = gv[i]; }
    return SVI_OK;
}

} // extern "C"
