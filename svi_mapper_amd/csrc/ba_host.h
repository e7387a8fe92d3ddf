// ba_host.h — host-side state of one svi_ba handle: the graph as the caller built it, the frozen
// device structures and the LM bookkeeping (g2o OptimizationAlgorithmLevenberg semantics).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <unordered_map>
#include <memory>
#include <vector>

#include "ba_device.h"
#include "common.h"

namespace svi {

struct HPose { int64_t id; double T[12]; int fixed; };
struct HLm   { int64_t id; double p[3]; int fixed; };
// Projection edges as they were added (insertion order), structure of arrays: the keys stay in ordinary vectors (the
// structure analysis sorts them), the values (z 3, upper triangle of the information 6: 72 bytes) sit in PINNED host memory
// and are appended to a device-side log as they arrive - an initialize() never touches them again (ba_structure.cpp).
struct EdgeStore {
    std::vector<int> pose, lm;          // indices into svi_ba::poses / lms
    std::vector<uint8_t> flags;         // (type & 3) | kFlagRobust
    double* vals = nullptr;             // pinned, 9 doubles per edge
    size_t  cap = 0;                    // edges the pinned array has room for
    size_t  n_offdiag = 0;              // edges whose information has off-diagonal entries
    size_t size() const { return pose.size(); }
    const double* z(size_t i) const { return vals + 9 * i; }
    const double* info(size_t i) const { return vals + 9 * i + 3; }
    int type(size_t i) const { return flags[i] & 3; }
    bool robust(size_t i) const { return (flags[i] & 4u) != 0; }
};
struct HSe3  { int i, j, robust; double Z[12]; double info[21]; };
struct HAcc  { int pose; double a[3]; double off[12]; double info[6]; };
struct HLL   { int i, j, robust; double z[3]; double info[6]; };

struct PhaseTimer {
    struct Rec { int phase; hipEvent_t a, b; };
    std::vector<Rec> pool;
    size_t used = 0;
    double  ms[SVI_PH_COUNT] = {0};
    int64_t calls[SVI_PH_COUNT] = {0};
    bool on = false;
    void begin(int phase, hipStream_t s);
    void end(hipStream_t s);
    void collect();   // completed records are added up, pending ones kept
    void drop_last(); // the record just closed does not count (a discarded speculative sweep)
    void release();
};

} // namespace svi

struct svi_ba {
    svi_ba_options opt{};
    // ---- graph (host) ----
    std::vector<svi::HPose> poses;
    std::vector<svi::HLm>   lms;
    svi::EdgeStore proj;
    std::vector<svi::HSe3>  se3;
    std::vector<svi::HAcc>  acc;
    std::vector<svi::HLL>   lmlm;
    std::unordered_map<int64_t, int> pose_ix, lm_ix;
    // g2o::ParameterSE3Offset eOFFSET_IMUtoLEFT (Cg2oOptimizer.cpp:209-213): the offset of the gravity edges of add_keyframe
    double imu_off[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};

    // ---- device ----
    hipStream_t stream = nullptr;
    // The Schur reduction runs on a stream of its own while the factorisation of the early dependency levels already works on
    // the stages it has finished (ba_host.cpp trial()).  The two streams are tied by VALUES IN MEMORY the command processor waits
    // on (hipStreamWaitValue64 / hipStreamWriteValue64 on signal memory): a wave of k_schur that leaves a stage is counted, the
    // last one publishes the trial's sequence number for that stage - no kernel ever spins on another stream's kernel.
    hipStream_t stream_schur = nullptr;
    unsigned long long* sig_lin = nullptr;                 // main -> Schur stream: the linearisation (and H_ll^-1) of this trial is ready
    unsigned long long* sig_stage[svi::kMaxStages] = {};   // Schur stream -> main: stage s of this trial is reduced
    unsigned long long stage_seq = 0;                      // sequence number of the last staged trial
    bool upd_clean[2] = {true, true};                      // d.upd[q] is all zero (or will be when the next trial's factorisation reads it)
    uint64_t staged_trials = 0;                            // staged trials so far: trial t uses d.upd[t & 1]
    int schur_launch_wgs = 0, schur_reserve_per_se = 0;    // staged k_schur: workgroups launched (two per CU), CUs per shader engine it leaves empty
    bool overlap_ok = false;                               // the device supports stream waits on memory and the streams / signals exist
    std::vector<int> sub_stage_ptr, level_stage;           // host mirrors of the stage ranges (sub-tiles, dependency levels)
    int spec_lin_state = -1;   // state whose Jacobian sweep + pose-only edges are already enqueued (speculation on "accepted"), or -1
    bool own_stream = false;
    bool host_stale = false; // an optimize() has run since the host copy of the estimates was refreshed (ensure_host)
    bool lin_local = false; // several ranks: the last linearisation kept its pose sums local (see linearize())
    bool initialized = false;
    svi::BaDev d{}, d_copy{};   // d_copy: what was sent to BaDev::self
    svi::CholPlan plan{};
    int cur = 0;
    std::vector<svi::DevBuf> pool;  // the device buffers of initialize(), in allocation order; kept (and grown) across calls
    svi::DevBuf raw_log, raw_flags; // projection-edge values (72 B) and flags in INSERTION order on the device, appended to
    size_t raw_cap = 0;             // edges the device log has room for
    size_t raw_uploaded = 0;        // edges of `proj` already in the log
    uint64_t graph_version = 0;     // bumped by every edit of the graph's STRUCTURE (vertices, edges, fixed flags)
    uint64_t built_version = ~0ull; // graph_version the device structures were built for
    double* h_scal = nullptr;       // pinned readback (16 doubles)
    int*    h_status = nullptr;     // pinned: [0] factorisation status, [1] sequence number of the last published results
    int     pub_seq = 0;
    int*    red_slot = nullptr;     // device [Pf]
    int*    e_orig = nullptr;       // device [E] lm-major edge -> insertion index
    double* lm_all = nullptr;       // device [3 * Ltot] gather buffer (multi-rank)
    // host mirrors of the structure
    std::vector<int> pose_order;    // slot -> index into poses
    std::vector<int> lm_order;      // global landmark slot -> index into lms
    std::shared_ptr<void> build_ctx; // ba_structure.cpp's working set, kept so that its vectors keep their memory
    bool hinv_valid = false;        // the landmark blocks are already inverted for hinv_lambda (done with the pose sums)
    double hinv_lambda = 0.0;
    bool lin_post_deferred = false; // the closing sums of the last linearisation are taken by the trial's reduction
    std::vector<int> h_step_ptr, h_tgt_ptr, h_trsm_ptr;
    std::vector<svi::ChainInline> chain_inl; // per level: the chain / back-substitution records as kernel arguments (ba_chol.hip)
    std::vector<svi::SolveInline> solve_inl;
    std::vector<int> red_perm;      // natural reduced pose index -> elimination (reduced) index
    int L0 = 0, L1 = 0;             // this rank's landmark slots [L0, L1)
    int64_t E_total = 0;
    svi_ba_stats stats{};

    // ---- LM ----
    double lambda = 0.0, ni = 2.0;
    double last_plain = 0.0, last_robust = 0.0;
    bool   have_chi = false;

    // ---- hooks / instrumentation ----
    svi_allreduce_fn ar = nullptr;
    void* ar_user = nullptr;
    svi::PhaseTimer timer;
    svi::PhaseTimer sweep_timer; // phase 0 = K2 + K3 inside the LM loop (options.sweep_events)
};

// host copy of the estimates in step with the device (ba_host.cpp); collective with several ranks when it has to act
int ensure_host(svi_ba* ba);
// edge store (ba_host.cpp): append one edge / drop everything from `n` on / bring the device-side log up to date
int edges_push(svi_ba* ba, int type, int robust, int pose, int lm, const double* z, const double* info);
void edges_truncate(svi_ba* ba, size_t n);
int edges_flush(svi_ba* ba);
