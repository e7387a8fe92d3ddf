// ba_host.h — host-side state of one svi_ba handle: the graph as the caller built it, the frozen
// device structures and the LM bookkeeping (g2o OptimizationAlgorithmLevenberg semantics).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <unordered_map>
#include <vector>

#include "ba_device.h"
#include "common.h"

namespace svi {

struct HPose { int64_t id; double T[12]; int fixed; };
struct HLm   { int64_t id; double p[3]; int fixed; };
struct HProj { int type, robust, pose, lm; double z[3]; double info[6]; };
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
    void collect(); // after a stream sync
    void release();
};

} // namespace svi

struct svi_ba {
    svi_ba_options opt{};
    // ---- graph (host) ----
    std::vector<svi::HPose> poses;
    std::vector<svi::HLm>   lms;
    std::vector<svi::HProj> proj;
    std::vector<svi::HSe3>  se3;
    std::vector<svi::HAcc>  acc;
    std::vector<svi::HLL>   lmlm;
    std::unordered_map<int64_t, int> pose_ix, lm_ix;
    // g2o::ParameterSE3Offset eOFFSET_IMUtoLEFT (Cg2oOptimizer.cpp:209-213): the offset of the gravity edges of add_keyframe
    double imu_off[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};

    // ---- device ----
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool host_stale = false; // an optimize() has run since the host copy of the estimates was refreshed (ensure_host)
    bool lin_local = false; // several ranks: the last linearisation kept its pose sums local (see linearize())
    bool initialized = false;
    svi::BaDev d{};
    svi::CholPlan plan{};
    int cur = 0;
    std::vector<void*> allocs;      // everything hipMalloc'ed by initialize()
    double* h_scal = nullptr;       // pinned readback (16 doubles)
    int*    h_status = nullptr;     // pinned: [0] factorisation status, [1] sequence number of the last published results
    int     pub_seq = 0;
    int*    red_slot = nullptr;     // device [Pf]
    int*    e_orig = nullptr;       // device [E] lm-major edge -> insertion index
    double* lm_all = nullptr;       // device [3 * Ltot] gather buffer (multi-rank)
    // host mirrors of the structure
    std::vector<int> pose_order;    // slot -> index into poses
    std::vector<int> lm_order;      // global landmark slot -> index into lms
    std::vector<int> h_step_ptr, h_tgt_ptr, h_trsm_ptr;
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
