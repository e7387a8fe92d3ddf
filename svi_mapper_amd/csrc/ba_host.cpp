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
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
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
void ba_pose_finalize(const BaDev& d, const int* red_slot, int rank, int n_ranks, void* st);
void ba_publish(const BaDev& d, int n, double* h_scal, int* h_status, int seq, void* st);
void ba_lin_post(const BaDev& d, int n_ranks, void* st);
void ba_invert_landmarks(const BaDev& d, double lambda, void* st);
void ba_schur(const BaDev& d, void* st);
void ba_assemble(const BaDev& d, void* st);
void ba_update_poses(const BaDev& d, int cur, double lambda, int scale_mode, int rank, void* st);
void ba_backsub_chi2(const BaDev& d, int cur, double lambda, void* st);
void ba_chi2_only(const BaDev& d, int which, void* st);
void ba_reduce_trial_scalars(const BaDev& d, int n_pub, double* h_scal, int* h_status, int seq, void* st);
void ba_debug_jacobians(const BaDev& d, int cur, const int* e_orig, double* err, double* Jp, double* Jl, void* st);
void ba_debug_aux_jacobians(const BaDev& d, int cur, double* se3_err, double* se3_Ji, double* se3_Jj, double* acc_err, double* acc_J, void* st);
void ba_configure_kernels(int TS);
int chol_potrf_probe(int tile, int reps, int stop_after, double* ms);
int chol_factor_solve(const CholPlan& p, double* S, double* Lt, double* Linv, double* g, double* x, double lambda, int n,
                      int* status, void* st);

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
    for (size_t i = 0; i < used; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, pool[i].a, pool[i].b) == hipSuccess) { ms[pool[i].phase] += t; calls[pool[i].phase]++; }
    }
    used = 0;
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

void free_device(svi_ba* ba)
{
    for (void* p : ba->allocs) (void)hipFree(p);
    ba->allocs.clear();
    if (ba->h_scal) { (void)hipHostFree(ba->h_scal); ba->h_scal = nullptr; }
    if (ba->h_status) { (void)hipHostFree(ba->h_status); ba->h_status = nullptr; }
    ba->initialized = false;
    ba->d = BaDev{};
    ba->plan = CholPlan{};
}

template <class T> int dev_upload(svi_ba* ba, const std::vector<T>& h, const T** out, size_t min_elems = 1)
{
    const size_t n = std::max(h.size(), min_elems);
    void* p = nullptr;
    SVI_HIP(hipMalloc(&p, n * sizeof(T)));
    ba->allocs.push_back(p);
    if (!h.empty()) SVI_HIP(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ba->stream));
    *out = static_cast<const T*>(p);
    return SVI_OK;
}
template <class T> int dev_alloc(svi_ba* ba, size_t n, T** out, bool zero = true)
{
    void* p = nullptr;
    n = std::max<size_t>(n, 1);
    SVI_HIP(hipMalloc(&p, n * sizeof(T)));
    ba->allocs.push_back(p);
    if (zero) SVI_HIP(hipMemsetAsync(p, 0, n * sizeof(T), ba->stream));
    *out = static_cast<T*>(p);
    return SVI_OK;
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
// structure analysis (g2o initializeOptimization + buildStructure equivalent)
// ---------------------------------------------------------------------------------------------
#define SVI_TIMING_MARK(k) do { if (dbg_t) { auto now_ = std::chrono::steady_clock::now(); fprintf(stderr, "build_structure: section %d starts at %.1f ms\n", k, std::chrono::duration<double, std::milli>(now_ - t_begin_).count()); } } while (0)
int build_structure(svi_ba* ba)
{
    const bool dbg_t = getenv("SVI_DEBUG_PLAN") != nullptr;
    const auto t_begin_ = std::chrono::steady_clock::now();
    BaDev& d = ba->d;
    const svi_ba_options& o = ba->opt;
    d.fx = o.fx; d.fy = o.fy; d.cx = o.cx; d.cy = o.cy; d.cauchy_delta = o.cauchy_delta;
    const int Pn = (int)ba->poses.size();
    const int Ltot = (int)ba->lms.size();

    SVI_TIMING_MARK(0);
    // ---- vertex order: ascending id (g2o index mapping) ----
    ba->pose_order.resize(Pn);
    std::iota(ba->pose_order.begin(), ba->pose_order.end(), 0);
    std::sort(ba->pose_order.begin(), ba->pose_order.end(), [&](int a, int b) { return ba->poses[a].id < ba->poses[b].id; });
    std::vector<int> pose_slot(Pn), pose_red(Pn), red_slot;
    for (int s = 0; s < Pn; ++s) pose_slot[ba->pose_order[s]] = s;
    for (int s = 0; s < Pn; ++s) {
        if (ba->poses[ba->pose_order[s]].fixed) pose_red[s] = -1;
        else { pose_red[s] = (int)red_slot.size(); red_slot.push_back(s); }
    }
    const int Pf = (int)red_slot.size();
    SVI_TIMING_MARK(1);
    // ---- elimination order of the reduced camera system (nested dissection of the key-frame sequence) ----
    // The reduced system of a trajectory is block-banded: in natural order its Cholesky is ONE chain of tile
    // columns.  Cutting the sequence at separators as wide as the co-visibility span gives independent chains
    // that are factorised side by side (ba_chol.hip processes all columns of one dependency level per launch).
    // Every piece is a whole number of tiles except the top separator, which comes last and absorbs the
    // remainder, so the identity padding stays at the end of the reduced index range.
    ba->red_perm.resize(Pf);
    std::iota(ba->red_perm.begin(), ba->red_perm.end(), 0);
    {
        const int TSo = o.chol_tile > 0 ? o.chol_tile : 96;
        const int PBo = TSo / 6;
        const int NTo = PBo > 0 ? (Pf + PBo - 1) / PBo : 0;
        if (o.chol_order == 0 && PBo > 0 && NTo >= 6) {
            // free poses of every landmark (natural reduced indices), pose-pose edges
            std::vector<std::vector<int>> lm_red(Ltot);
            for (const HProj& e : ba->proj) {
                const int r = pose_red[pose_slot[e.pose]];
                if (r >= 0 && !ba->lms[e.lm].fixed) lm_red[e.lm].push_back(r);
            }
            int span = 0; // largest |r_i - r_j| over coupled free poses
            for (auto& v : lm_red) {
                if (v.empty()) continue;
                const auto mm = std::minmax_element(v.begin(), v.end());
                span = std::max(span, *mm.second - *mm.first);
            }
            std::vector<std::pair<int, int>> pp;
            for (const HSe3& e : ba->se3) {
                const int ri = pose_red[pose_slot[e.i]], rj = pose_red[pose_slot[e.j]];
                if (ri >= 0 && rj >= 0) { pp.push_back({ri, rj}); span = std::max(span, std::abs(ri - rj)); }
            }
            const int rem = Pf % PBo;
            // elimination order for separators of `w` tiles; false if the sequence is too short for it
            auto make_perm = [&](int w, std::vector<int>& perm) {
                const int sep = w * PBo;
                const int sep_top = rem == 0 ? sep : rem + PBo * ((std::max(sep - rem, 0) + PBo - 1) / PBo); // absorbs the remainder
                if (Pf < sep_top + 4 * PBo) return false;
                std::vector<std::pair<int, int>> pieces; // natural ranges in elimination order
                std::function<void(int, int, int)> rec = [&](int a, int b, int wd) {
                    const int len = b - a;
                    if (len < wd + 2 * PBo) { if (len > 0) pieces.push_back({a, b}); return; }
                    const int left = PBo * (((len - wd) / PBo) / 2);
                    rec(a, a + left, sep);
                    rec(a + left + wd, b, sep);
                    pieces.push_back({a + left, a + left + wd});
                };
                rec(0, Pf, sep_top);
                perm.assign(Pf, 0);
                int pos = 0;
                for (auto& pc : pieces) for (int r = pc.first; r < pc.second; ++r) perm[r] = pos++;
                return true;
            };
            // which free poses share a landmark (lower triangle, natural reduced indices): computed once, every
            // candidate order only re-maps it to tiles
            const bool use_cpl = Pf <= 4096; // 16 MB at most; longer sequences walk the landmarks per candidate
            std::vector<uint8_t> cpl(use_cpl ? (size_t)Pf * Pf : 0, 0);
            if (use_cpl)
                for (auto& lr : lm_red)
                    for (size_t x = 0; x < lr.size(); ++x)
                        for (size_t y = 0; y < lr.size(); ++y)
                            if (lr[y] <= lr[x]) cpl[(size_t)lr[x] * Pf + lr[y]] = 1;
            // dependency levels (= launches on the critical path) and filled tiles of an order
            auto analyse = [&](const std::vector<int>& perm, int& depth, int& tiles) {
                std::vector<uint8_t> z((size_t)NTo * NTo, 0);
                for (int t = 0; t < NTo; ++t) z[(size_t)t * NTo + t] = 1;
                if (use_cpl) {
                    for (int ri = 0; ri < Pf; ++ri) { // pose-level coupling mapped to tiles
                        const int tx = perm[ri] / PBo;
                        const uint8_t* row = &cpl[(size_t)ri * Pf];
                        for (int rj = 0; rj <= ri; ++rj)
                            if (row[rj]) { const int ty = perm[rj] / PBo; z[(size_t)std::max(tx, ty) * NTo + std::min(tx, ty)] = 1; }
                    }
                } else {
                    std::vector<int> v;
                    for (auto& lr : lm_red) {
                        v.clear();
                        for (int r : lr) v.push_back(perm[r] / PBo);
                        std::sort(v.begin(), v.end());
                        v.erase(std::unique(v.begin(), v.end()), v.end());
                        for (size_t x = 0; x < v.size(); ++x) for (size_t y = 0; y <= x; ++y) z[(size_t)v[x] * NTo + v[y]] = 1;
                    }
                }
                for (auto& e : pp) { const int x = perm[e.first] / PBo, y = perm[e.second] / PBo; z[(size_t)std::max(x, y) * NTo + std::min(x, y)] = 1; }
                std::vector<int> rows;
                for (int k = 0; k < NTo; ++k) {
                    rows.clear();
                    for (int i = k + 1; i < NTo; ++i) if (z[(size_t)i * NTo + k]) rows.push_back(i);
                    for (size_t x = 0; x < rows.size(); ++x) for (size_t y = 0; y <= x; ++y) z[(size_t)rows[x] * NTo + rows[y]] = 1;
                }
                std::vector<int> lev(NTo, 0);
                depth = 0; tiles = 0;
                for (int c = 0; c < NTo; ++c) {
                    for (int q = 0; q < c; ++q) if (z[(size_t)c * NTo + q]) { lev[c] = std::max(lev[c], lev[q] + 1); }
                    for (int q = 0; q <= c; ++q) tiles += z[(size_t)c * NTo + q];
                    depth = std::max(depth, lev[c] + 1);
                }
            };
            // candidates: natural order and separators of 1 .. ceil(span / tile) tiles (a separator narrower than the
            // longest track still gives a valid order - the few tracks that cross it only add dependencies); keep the
            // order with the fewest levels, then the fewest tiles
            std::vector<int> best = ba->red_perm, cand;
            int best_depth = 0, best_tiles = 0;
            analyse(best, best_depth, best_tiles);
            const int wmax = std::max(1, (span + PBo - 1) / PBo);
            // (the symbolic analysis is cubic in the tile count: very long sequences only try the span-wide separator)
            for (int w = (NTo > 256 ? std::min(wmax, 8) : 1); w <= wmax && w <= 8; ++w) {
                if (!make_perm(w, cand)) break;
                int dp = 0, tl = 0;
                analyse(cand, dp, tl);
                if (dp < best_depth || (dp == best_depth && tl < best_tiles)) { best = cand; best_depth = dp; best_tiles = tl; }
            }
            ba->red_perm = best;
            for (int sl = 0; sl < Pn; ++sl) if (pose_red[sl] >= 0) pose_red[sl] = ba->red_perm[pose_red[sl]];
            for (int sl = 0; sl < Pn; ++sl) if (pose_red[sl] >= 0) red_slot[pose_red[sl]] = sl;
        }
    }
    ba->lm_order.resize(Ltot);
    std::iota(ba->lm_order.begin(), ba->lm_order.end(), 0);
    std::sort(ba->lm_order.begin(), ba->lm_order.end(), [&](int a, int b) { return ba->lms[a].id < ba->lms[b].id; });
    std::vector<int> lm_slot(Ltot);
    for (int s = 0; s < Ltot; ++s) lm_slot[ba->lm_order[s]] = s;

    SVI_TIMING_MARK(2);
    // ---- landmark sharding: contiguous slot ranges balanced by projection-edge count ----
    std::vector<int64_t> deg(Ltot + 1, 0);
    for (const HProj& e : ba->proj) deg[lm_slot[e.lm] + 1]++;
    for (int s = 0; s < Ltot; ++s) deg[s + 1] += deg[s];
    const int64_t Etot = (int64_t)ba->proj.size();
    ba->E_total = Etot;
    auto bound = [&](int r) -> int {
        if (r <= 0) return 0;
        if (r >= o.n_ranks) return Ltot;
        const int64_t want = Etot * r / o.n_ranks;
        int s = (int)(std::lower_bound(deg.begin(), deg.end(), want) - deg.begin());
        return std::min(std::max(s, 0), Ltot);
    };
    ba->L0 = bound(o.rank);
    ba->L1 = bound(o.rank + 1);
    const int L0 = ba->L0, Ll = ba->L1 - ba->L0;

    SVI_TIMING_MARK(3);
    // ---- local projection edges, lm-major: (landmark, reduced pose [fixed first], slot) ----
    std::vector<int> loc;
    loc.reserve(ba->proj.size());
    for (int i = 0; i < (int)ba->proj.size(); ++i) {
        const int s = lm_slot[ba->proj[i].lm];
        if (s >= L0 && s < ba->L1) loc.push_back(i);
    }
    const int E = (int)loc.size();
    {
        // order: landmark slot, then reduced pose index (fixed poses first), then pose slot, then insertion order -
        // two stable counting sorts (by pose rank, then by landmark) instead of a comparison sort over 800 k records
        std::vector<int> prank(Pn), pidx(Pn);
        for (int s = 0; s < Pn; ++s) pidx[s] = s;
        std::sort(pidx.begin(), pidx.end(), [&](int a, int b) { return pose_red[a] != pose_red[b] ? pose_red[a] < pose_red[b] : a < b; });
        for (int r = 0; r < Pn; ++r) prank[pidx[r]] = r;
        auto counting_sort = [](std::vector<int>& v, int n_keys, auto&& key) {
            std::vector<int> cnt(n_keys + 1, 0), out(v.size());
            for (int x : v) cnt[key(x) + 1]++;
            for (int k = 0; k < n_keys; ++k) cnt[k + 1] += cnt[k];
            for (int x : v) out[cnt[key(x)]++] = x;
            v.swap(out);
        };
        counting_sort(loc, Pn, [&](int i) { return prank[pose_slot[ba->proj[i].pose]]; });
        counting_sort(loc, ba->L1 - L0, [&](int i) { return lm_slot[ba->proj[i].lm] - L0; });
    }
    bool diag_info = true;
    for (const HProj& e : ba->proj) if (e.info[1] != 0.0 || e.info[2] != 0.0 || e.info[4] != 0.0) { diag_info = false; break; }
    const int planes = diag_info ? 3 : 6;
    static const int kDiagIdx[3] = {0, 3, 5};
    std::vector<int> e_pose(E), e_lm(E), e_orig(E), lm_ptr(Ll + 1, 0);
    std::vector<uint8_t> e_flags(E);
    const int np2 = (3 + planes + 1) / 2; // double2 planes of the packed [z | information | pad] record
    std::vector<double> e_zi((size_t)2 * np2 * std::max(E, 1), 0.0);
    auto zi_at = [&](std::vector<double>& a, int v, int k) -> double& { return a[2 * ((size_t)(v / 2) * E + k) + (v & 1)]; };
    for (int k = 0; k < E; ++k) {
        const HProj& e = ba->proj[loc[k]];
        e_pose[k] = pose_slot[e.pose];
        e_lm[k] = lm_slot[e.lm] - L0;
        e_orig[k] = loc[k];
        e_flags[k] = (uint8_t)((e.type & 3) | (e.robust ? kFlagRobust : 0));
        for (int c = 0; c < 3; ++c) zi_at(e_zi, c, k) = e.z[c];
        for (int c = 0; c < planes; ++c) zi_at(e_zi, 3 + c, k) = diag_info ? e.info[kDiagIdx[c]] : e.info[c];
        lm_ptr[e_lm[k] + 1]++;
    }
    for (int l = 0; l < Ll; ++l) {
        if (lm_ptr[l + 1] > kLmBlockEdges)
            return fail(SVI_ERR_UNSUPPORTED, "landmark %lld has %d projection edges (limit %d)",
                        (long long)ba->lms[ba->lm_order[L0 + l]].id, lm_ptr[l + 1], kLmBlockEdges);
        lm_ptr[l + 1] += lm_ptr[l];
    }
    std::vector<int> lb_lm(1, 0);
    for (int l = 0; l < Ll;) {
        int l2 = l, edges = 0;
        while (l2 < Ll && l2 - l < kLmBlockEdges && edges + (lm_ptr[l2 + 1] - lm_ptr[l2]) <= kLmBlockEdges) { edges += lm_ptr[l2 + 1] - lm_ptr[l2]; ++l2; }
        lb_lm.push_back(l2);
        l = l2;
    }
    const int n_lm_blocks = (int)lb_lm.size() - 1;
    std::vector<uint8_t> lm_fixed(Ll);
    for (int l = 0; l < Ll; ++l) lm_fixed[l] = (uint8_t)(ba->lms[ba->lm_order[L0 + l]].fixed ? 1 : 0);

    SVI_TIMING_MARK(4);
    // ---- pose-major copy: free poses only, (slot, landmark) ----
    std::vector<int> pm;
    pm.reserve(E);
    for (int k = 0; k < E; ++k) if (pose_red[e_pose[k]] >= 0) pm.push_back(k);
    { // stable counting sort by pose slot
        std::vector<int> cnt(Pn + 1, 0), out(pm.size());
        for (int x : pm) cnt[e_pose[x] + 1]++;
        for (int k = 0; k < Pn; ++k) cnt[k + 1] += cnt[k];
        for (int x : pm) out[cnt[e_pose[x]]++] = x;
        pm.swap(out);
    }
    const int Epm = (int)pm.size();
    std::vector<int> pm_lm(Epm), chunk_pose, chunk_begin, pose_chunk_ptr(Pn + 1, 0);
    std::vector<uint8_t> pm_flags(Epm);
    // the pose-major copy uses the same plane stride E as the lm-major arrays (the kernels share load_edge)
    std::vector<double> pm_zi((size_t)2 * np2 * std::max(E, 1), 0.0);
    for (int k = 0; k < Epm; ++k) {
        const int src = pm[k];
        pm_lm[k] = e_lm[src];
        pm_flags[k] = e_flags[src];
        for (int v = 0; v < 3 + planes; ++v) zi_at(pm_zi, v, k) = zi_at(e_zi, v, src);
    }
    {
        int k = 0;
        for (int s = 0; s < Pn; ++s) {
            pose_chunk_ptr[s] = (int)chunk_pose.size();
            int k2 = k;
            while (k2 < Epm && e_pose[pm[k2]] == s) ++k2;
            for (int b = k; b < k2; b += kPoseChunk) { chunk_pose.push_back(s); chunk_begin.push_back(b); }
            k = k2;
        }
        pose_chunk_ptr[Pn] = (int)chunk_pose.size();
        chunk_begin.push_back(Epm);
    }
    const int n_chunks = (int)chunk_pose.size();
    // chunks are contiguous: the next chunk (same or next pose) starts exactly where this one ends,
    // so chunk_begin[c+1] is the end of chunk c

    SVI_TIMING_MARK(5);
    // ---- pose-only / landmark-only edges ----
    std::vector<int> se3_i, se3_j, acc_pose, ll_free;
    std::vector<double> se3_Z, se3_info, acc_a, acc_info, ll_ref, ll_z, ll_info;
    std::vector<uint8_t> se3_robust, ll_robust;
    std::vector<std::vector<int>> pose_aux(Pn);
    if (o.rank == 0) {
        for (const HSe3& e : ba->se3) {
            const int k = (int)se3_i.size();
            se3_i.push_back(pose_slot[e.i]); se3_j.push_back(pose_slot[e.j]);
            se3_Z.insert(se3_Z.end(), e.Z, e.Z + 12);
            se3_info.insert(se3_info.end(), e.info, e.info + 21);
            se3_robust.push_back((uint8_t)(e.robust ? 1 : 0));
            pose_aux[pose_slot[e.i]].push_back((k << 2) | 0);
            pose_aux[pose_slot[e.j]].push_back((k << 2) | 1);
        }
        for (const HAcc& e : ba->acc) {
            const int k = (int)acc_pose.size();
            acc_pose.push_back(pose_slot[e.pose]);
            for (int r = 0; r < 3; ++r) acc_a.push_back(e.off[3 * r] * e.a[0] + e.off[3 * r + 1] * e.a[1] + e.off[3 * r + 2] * e.a[2]);
            acc_info.insert(acc_info.end(), e.info, e.info + 6);
            pose_aux[pose_slot[e.pose]].push_back((k << 2) | 2);
        }
    }
    std::vector<int> pose_aux_ptr(Pn + 1, 0), pose_aux_ref;
    for (int s = 0; s < Pn; ++s) {
        pose_aux_ptr[s] = (int)pose_aux_ref.size();
        pose_aux_ref.insert(pose_aux_ref.end(), pose_aux[s].begin(), pose_aux[s].end());
    }
    pose_aux_ptr[Pn] = (int)pose_aux_ref.size();
    {
        struct LL { int free_l; double ref[3], z[3], info[6]; uint8_t robust; };
        std::vector<LL> v;
        for (const HLL& e : ba->lmlm) {
            const HLm &li = ba->lms[e.i], &lj = ba->lms[e.j];
            if (!li.fixed && !lj.fixed)
                return fail(SVI_ERR_UNSUPPORTED, "landmark-landmark edge %lld-%lld with two free ends (the reference fixes one, Cg2oOptimizer.cpp:445)",
                            (long long)li.id, (long long)lj.id);
            if (li.fixed && lj.fixed) continue;
            LL x{};
            const bool free_is_j = li.fixed != 0;
            const HLm& fr = free_is_j ? lj : li;
            const HLm& fx = free_is_j ? li : lj;
            const int s = lm_slot[free_is_j ? e.j : e.i];
            if (s < L0 || s >= ba->L1) continue;
            x.free_l = s - L0;
            (void)fr;
            for (int c = 0; c < 3; ++c) { x.ref[c] = fx.p[c]; x.z[c] = free_is_j ? e.z[c] : -e.z[c]; }
            memcpy(x.info, e.info, sizeof(x.info));
            x.robust = (uint8_t)(e.robust ? 1 : 0);
            v.push_back(x);
        }
        std::stable_sort(v.begin(), v.end(), [](const LL& a, const LL& b) { return a.free_l < b.free_l; });
        for (const LL& x : v) {
            ll_free.push_back(x.free_l);
            ll_ref.insert(ll_ref.end(), x.ref, x.ref + 3);
            ll_z.insert(ll_z.end(), x.z, x.z + 3);
            ll_info.insert(ll_info.end(), x.info, x.info + 6);
            ll_robust.push_back(x.robust);
        }
    }
    std::vector<int> lm_ll_ptr(Ll + 1, 0);
    for (int f : ll_free) lm_ll_ptr[f + 1]++;
    for (int l = 0; l < Ll; ++l) lm_ll_ptr[l + 1] += lm_ll_ptr[l];

    SVI_TIMING_MARK(6);
    // ---- reduced system tiling (identical on every rank: derived from the GLOBAL graph) ----
    int TS = o.chol_tile > 0 ? o.chol_tile : 96;
    if (TS % 48 != 0 || TS > kMaxTile) return fail(SVI_ERR_INVALID, "chol_tile must be 48 or 96");
    const int PB = TS / 6;
    const int n = 6 * Pf;
    const int NT = (n + TS - 1) / TS;
    std::vector<uint8_t> nz((size_t)NT * NT, 0);
    for (int t = 0; t < NT; ++t) nz[(size_t)t * NT + t] = 1;
    {
        // chunks touched by every landmark of the global graph
        std::vector<std::vector<int>> lm_chunks(Ltot);
        for (const HProj& e : ba->proj) {
            const int r = pose_red[pose_slot[e.pose]];
            if (r >= 0 && !ba->lms[e.lm].fixed) lm_chunks[e.lm].push_back(r / PB);
        }
        for (auto& v : lm_chunks) {
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            for (size_t a = 0; a < v.size(); ++a)
                for (size_t b = 0; b <= a; ++b) nz[(size_t)v[a] * NT + v[b]] = 1;
        }
        for (const HSe3& e : ba->se3) {
            const int ri = pose_red[pose_slot[e.i]], rj = pose_red[pose_slot[e.j]];
            if (ri >= 0 && rj >= 0) { const int a = std::max(ri, rj) / PB, b = std::min(ri, rj) / PB; nz[(size_t)a * NT + b] = 1; }
        }
    }
    const std::vector<uint8_t> nz_orig = nz; // tiles that receive Schur / pose-edge contributions (before fill-in)
    // symbolic fill, right-looking over tile columns
    std::vector<int> h_col_ptr(NT + 1, 0);
    std::vector<std::pair<int, int>> col_rows;            // (k, i)
    std::vector<int> upd_i, upd_j, upd_k;
    for (int k = 0; k < NT; ++k) {
        std::vector<int> rows;
        for (int i = k + 1; i < NT; ++i) if (nz[(size_t)i * NT + k]) rows.push_back(i);
        h_col_ptr[k] = (int)col_rows.size();
        for (int i : rows) col_rows.push_back({k, i});
        for (size_t a = 0; a < rows.size(); ++a)
            for (size_t b = 0; b <= a; ++b) {
                nz[(size_t)rows[a] * NT + rows[b]] = 1;
                upd_i.push_back(rows[a]); upd_j.push_back(rows[b]); upd_k.push_back(k);
            }
    }
    h_col_ptr[NT] = (int)col_rows.size();
    // tile ids: the tiles with contributions first, pure fill-in tiles after them - only the former (and g) have to
    // cross the all-reduce, the latter are zero on every rank until the factorisation fills them
    std::vector<int> tile_map((size_t)NT * NT, -1), tile_ti, tile_tj;
    for (int pass = 0; pass < 2; ++pass)
        for (int j = 0; j < NT; ++j)
            for (int i = j; i < NT; ++i)
                if (nz[(size_t)i * NT + j] && (nz_orig[(size_t)i * NT + j] != 0) == (pass == 0)) {
                    tile_map[(size_t)i * NT + j] = (int)tile_ti.size(); tile_ti.push_back(i); tile_tj.push_back(j);
                }
    const int n_tiles = (int)tile_ti.size();
    int n_tiles_orig = 0;
    for (size_t q = 0; q < nz_orig.size(); ++q) n_tiles_orig += nz_orig[q] ? 1 : 0;
    std::vector<int> trsm_tile, trsm_row, diag_tile(NT);
    for (auto& kr : col_rows) { trsm_tile.push_back(tile_map[(size_t)kr.second * NT + kr.first]); trsm_row.push_back(kr.second); }
    for (int k = 0; k < NT; ++k) diag_tile[k] = tile_map[(size_t)k * NT + k];

    // dependency levels: column c waits for every column p < c with a tile (c,p); all columns of one level are
    // factorised by one launch (ba_chol.hip).  The update of a diagonal tile by a column of the level just below
    // is applied by the workgroup that factorises it ("pre" list); every other update is grouped by TARGET tile
    // and runs in the launch that follows its source column's level, one workgroup set per target with the
    // sources in ascending order (no two workgroups ever write the same tile: deterministic without atomics).
    std::vector<int> level(NT, 0);
    for (int c = 0; c < NT; ++c)
        for (int q = 0; q < c; ++q) if (tile_map[(size_t)c * NT + q] >= 0) level[c] = std::max(level[c], level[q] + 1);
    const int n_steps = NT ? *std::max_element(level.begin(), level.end()) + 1 : 0;
    std::vector<int> h_step_ptr(n_steps + 1, 0), step_col;
    for (int st = 0; st < n_steps; ++st) {
        h_step_ptr[st] = (int)step_col.size();
        for (int c = 0; c < NT; ++c) if (level[c] == st) step_col.push_back(c);
    }
    h_step_ptr[n_steps] = (int)step_col.size();
    std::vector<int> pre_ptr(NT + 1, 0), pre_tile, pre_col;
    for (int c = 0; c < NT; ++c) {
        pre_ptr[c] = (int)pre_tile.size();
        for (int q = 0; q < c; ++q)
            if (tile_map[(size_t)c * NT + q] >= 0 && level[q] == level[c] - 1) { pre_tile.push_back(tile_map[(size_t)c * NT + q]); pre_col.push_back(q); }
    }
    pre_ptr[NT] = (int)pre_tile.size();
    // target-grouped updates per launch step
    std::vector<int> h_tgt_ptr(n_steps + 1, 0), tgt_tile, tgt_row, tgt_pair_ptr(1, 0), pair_a, pair_b, pair_src;
    double chol_flops = 0.0;
    {
        std::vector<std::vector<size_t>> by_step(n_steps);
        for (size_t u = 0; u < upd_i.size(); ++u) {
            const int i = upd_i[u], j = upd_j[u], q = upd_k[u];
            if (i == j && level[q] == level[i] - 1) continue; // pre-update, done by the factorising workgroup
            by_step[level[q] + 1].push_back(u);
        }
        for (int st = 0; st < n_steps; ++st) {
            h_tgt_ptr[st] = (int)tgt_tile.size();
            auto& v = by_step[st];
            std::stable_sort(v.begin(), v.end(), [&](size_t x, size_t y) {
                const int tx = tile_map[(size_t)upd_i[x] * NT + upd_j[x]], ty = tile_map[(size_t)upd_i[y] * NT + upd_j[y]];
                return tx != ty ? tx < ty : upd_k[x] < upd_k[y];
            });
            int cur = -1;
            for (size_t w = 0; w < v.size(); ++w) {
                const size_t u = v[w];
                const int tt = tile_map[(size_t)upd_i[u] * NT + upd_j[u]];
                if (tt != cur) {
                    tgt_tile.push_back(tt);
                    tgt_row.push_back(upd_i[u] == upd_j[u] ? upd_i[u] : -1);
                    tgt_pair_ptr.push_back(tgt_pair_ptr.back());
                    cur = tt;
                }
                pair_a.push_back(tile_map[(size_t)upd_i[u] * NT + upd_k[u]]);
                pair_b.push_back(tile_map[(size_t)upd_j[u] * NT + upd_k[u]]);
                pair_src.push_back(upd_k[u]);
                tgt_pair_ptr.back()++;
            }
        }
        h_tgt_ptr[n_steps] = (int)tgt_tile.size();
        const double t3 = (double)TS * TS * TS;
        chol_flops = t3 / 3.0 * NT + t3 * (double)col_rows.size() + 2.0 * t3 * (double)upd_i.size();
    }
    // trsm items per step
    std::vector<int> h_trsm_ptr(n_steps + 1, 0), st_tile, st_col;
    for (int st = 0; st < n_steps; ++st) {
        h_trsm_ptr[st] = (int)st_tile.size();
        for (int q = h_step_ptr[st]; q < h_step_ptr[st + 1]; ++q) {
            const int c = step_col[q];
            for (int w = h_col_ptr[c]; w < h_col_ptr[c + 1]; ++w) { st_tile.push_back(trsm_tile[w]); st_col.push_back(c); }
        }
    }
    h_trsm_ptr[n_steps] = (int)st_tile.size();
    if (getenv("SVI_DEBUG_PLAN")) {
        for (int st = 0; st < n_steps; ++st) {
            int mxpre = 0, mxpair = 0;
            for (int q = h_step_ptr[st]; q < h_step_ptr[st + 1]; ++q) mxpre = std::max(mxpre, pre_ptr[step_col[q] + 1] - pre_ptr[step_col[q]]);
            for (int t = h_tgt_ptr[st]; t < h_tgt_ptr[st + 1]; ++t) mxpair = std::max(mxpair, tgt_pair_ptr[t + 1] - tgt_pair_ptr[t]);
            fprintf(stderr, "level %d: %d columns, max pre sources %d, %d update targets, max pairs per target %d, %d trsm tiles\n", st,
                    h_step_ptr[st + 1] - h_step_ptr[st], mxpre, h_tgt_ptr[st + 1] - h_tgt_ptr[st], mxpair, h_trsm_ptr[st + 1] - h_trsm_ptr[st]);
        }
    }

    SVI_TIMING_MARK(7);
    // ---- Schur decomposition: always on 48 x 48 sub-tiles (8 poses x 8 poses), whatever TS is ----
    // item  = (landmark, row chunk cX, column chunk cY): the poses of the landmark in either chunk as
    //         8-bit masks plus the first lm-major edge of each segment
    // job   = a run of items of ONE sub-tile, processed by one wavefront (lane = 6x6 block of the sub-tile)
    constexpr int PBS = 8, SUB = 48;
    const int Q = TS / SUB;          // sub-tiles per tile edge
    const int NSUB = NT * Q;         // sub-tile rows of the padded system
    // duplicate (pose, landmark) edges would alias one 6x6 block inside an item; the reference never
    // creates them (one measurement per landmark per keyframe), reject instead of mis-summing
    for (int l = 0; l < Ll; ++l)
        for (int a = lm_ptr[l] + 1; a < lm_ptr[l + 1]; ++a)
            if (e_pose[a] == e_pose[a - 1])
                return fail(SVI_ERR_UNSUPPORTED, "two projection edges between pose %lld and landmark %lld",
                            (long long)ba->poses[ba->pose_order[e_pose[a]]].id, (long long)ba->lms[ba->lm_order[L0 + l]].id);
    // stored sub-tiles: every lower sub-tile inside a stored tile (they all have to be (re)written per trial)
    std::vector<int> sub_map((size_t)NSUB * NSUB, -1), sub_cx, sub_cy, sub_tile;
    for (int t = 0; t < n_tiles; ++t)
        for (int sx = 0; sx < Q; ++sx)
            for (int sy = 0; sy < Q; ++sy) {
                const int cx = tile_ti[t] * Q + sx, cy = tile_tj[t] * Q + sy;
                if (cy > cx) continue;
                sub_map[(size_t)cx * NSUB + cy] = (int)sub_cx.size();
                sub_cx.push_back(cx); sub_cy.push_back(cy); sub_tile.push_back(t);
            }
    const int n_sub = (int)sub_cx.size();
    // Schur work list.  A stored 48 x 48 sub-tile is four CELLS of 4 x 4 poses (24 x 24); an item is one landmark in one
    // cell: its edges to the cell's row poses and to its column poses (masks over the four poses of each).  Cells of
    // four poses instead of eight raise the share of (pose, pose) lanes that have work from 36 % to 61 % at KITTI-like
    // co-visibility.  A quarter job is a run of <= L items of one cell, a wavefront job four quarter jobs of similar
    // length (one per group of 16 lanes), so that the four quarters of a wave finish together.
    constexpr int PQ = 4;
    struct Item { int cell, lm, a0, b0, masks; };
    std::vector<Item> items;
    items.reserve((size_t)E * 2);
    struct Seg { int chunk, begin, mask, count; };
    std::vector<Seg> seg;
    int64_t total_pairs = 0;
    for (int l = 0; l < Ll; ++l) {
        if (lm_fixed[l]) continue;
        int a = lm_ptr[l];
        const int end = lm_ptr[l + 1];
        while (a < end && pose_red[e_pose[a]] < 0) ++a; // edges to fixed poses come first
        seg.clear();
        while (a < end) {
            const int c = pose_red[e_pose[a]] / PQ;
            Seg sg{c, a, 0, 0};
            while (a < end && pose_red[e_pose[a]] / PQ == c) { sg.mask |= 1 << (pose_red[e_pose[a]] % PQ); ++sg.count; ++a; }
            seg.push_back(sg);
        }
        for (size_t x = 0; x < seg.size(); ++x)
            for (size_t y = 0; y <= x; ++y) {
                const int qx = seg[x].chunk, qy = seg[y].chunk; // qx >= qy: edges of a landmark ascend in reduced index
                if (qy > qx) return fail(SVI_ERR_STATE, "internal: landmark edges not in reduced pose order");
                const int sub = sub_map[(size_t)(qx / 2) * NSUB + qy / 2];
                if (sub < 0) return fail(SVI_ERR_STATE, "internal: Schur sub-tile outside the tile structure");
                items.push_back({4 * sub + 2 * (qx % 2) + (qy % 2), l, seg[x].begin, seg[y].begin, seg[x].mask | (seg[y].mask << 8)});
                total_pairs += (x == y) ? (int64_t)seg[x].count * (seg[x].count + 1) / 2 : (int64_t)seg[x].count * seg[y].count;
            }
    }
    { // stable counting sort by cell
        std::vector<int> cnt(4 * n_sub + 1, 0);
        std::vector<Item> out(items.size());
        for (const Item& it : items) cnt[it.cell + 1]++;
        for (int c = 0; c < 4 * n_sub; ++c) cnt[c + 1] += cnt[c];
        for (const Item& it : items) out[cnt[it.cell]++] = it;
        items.swap(out);
    }
    const int n_items = (int)items.size();
    std::vector<int> it_pack((size_t)4 * std::max(n_items, 1));
    for (int i = 0; i < n_items; ++i) {
        it_pack[4 * i] = items[i].lm; it_pack[4 * i + 1] = items[i].a0; it_pack[4 * i + 2] = items[i].b0; it_pack[4 * i + 3] = items[i].masks;
    }
    // quarter jobs: as many as fit on the chip at once - the kernel holds two waves per SIMD (214 VGPRs, 59 KB of LDS
    // per workgroup), one more would wait for a whole round.  Measured at config 4 with the current kernels (quarter
    // jobs: Schur + assemble us): 4096: 207 + 22, 6144: 185 + 27, 8192: 176 + 33, 10240: 209 + 38.
    const int n_cells = 4 * n_sub;
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ba->opt.device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount; }
    const int64_t qj_cap = (int64_t)4 * (n_cu * 4 * 2);
    std::vector<int> cell_count;
    for (int i = 0; i < n_items;) { int j = i; while (j < n_items && items[j].cell == items[i].cell) ++j; cell_count.push_back(j - i); i = j; }
    auto pieces = [&](int len) { int64_t n = 0; for (int c : cell_count) n += (c + len - 1) / len; return n; };
    int L = 16;
    while (L < 1024 && pieces(L) > qj_cap) ++L;
    struct QJob { int begin, end, cell; };
    std::vector<QJob> qjobs;
    for (int i = 0; i < n_items;) {
        const int cell = items[i].cell;
        int j = i;
        while (j < n_items && items[j].cell == cell && j - i < L) ++j;
        qjobs.push_back({i, j, cell});
        i = j;
    }
    // waves take four quarter jobs of similar length; the slabs of a cell are summed in the order of its pieces
    std::vector<int> qorder(qjobs.size());
    for (size_t i = 0; i < qorder.size(); ++i) qorder[i] = (int)i;
    std::stable_sort(qorder.begin(), qorder.end(), [&](int a, int b) { return qjobs[a].end - qjobs[a].begin > qjobs[b].end - qjobs[b].begin; });
    const int n_jobs = ((int)qjobs.size() + 3) / 4;
    std::vector<int> qj_begin((size_t)4 * std::max(n_jobs, 1), 0), qj_end((size_t)4 * std::max(n_jobs, 1), 0), qj_diag((size_t)4 * std::max(n_jobs, 1), 0);
    std::vector<int> job_len(std::max(n_jobs, 1), 0), slot_of(qjobs.size(), -1);
    for (size_t k = 0; k < qorder.size(); ++k) {
        const QJob& q = qjobs[qorder[k]];
        qj_begin[k] = q.begin; qj_end[k] = q.end;
        const int sub = q.cell / 4, u = (q.cell / 2) % 2, v = q.cell % 2;
        qj_diag[k] = (sub_cx[sub] == sub_cy[sub] && u == v) ? 1 : 0;
        job_len[k / 4] = std::max(job_len[k / 4], q.end - q.begin);
        slot_of[qorder[k]] = (int)k;
    }
    std::vector<int> cell_qj_ptr(n_cells + 1, 0), cell_qj;
    {
        size_t k = 0;
        for (int c = 0; c < n_cells; ++c) {
            cell_qj_ptr[c] = (int)cell_qj.size();
            while (k < qjobs.size() && qjobs[k].cell == c) { cell_qj.push_back(slot_of[k]); ++k; } // qjobs ascend in cell
        }
        cell_qj_ptr[n_cells] = (int)cell_qj.size();
    }
    std::vector<std::vector<int>> taux(n_sub);
    for (int k = 0; k < (int)se3_i.size(); ++k) {
        const int ri = pose_red[se3_i[k]], rj = pose_red[se3_j[k]];
        if (ri < 0 || rj < 0 || ri == rj) continue;
        const int tr = ri > rj ? 0 : 1; // row pose = the one with the larger reduced index
        const int hi = std::max(ri, rj), lo = std::min(ri, rj);
        const int sub = sub_map[(size_t)(hi / PBS) * NSUB + lo / PBS];
        if (sub < 0) return fail(SVI_ERR_STATE, "internal: odometry block outside the tile structure");
        taux[sub].push_back((k << 1) | tr);
    }
    std::vector<int> sub_aux_ptr(n_sub + 1, 0), sub_aux_ref;
    for (int t = 0; t < n_sub; ++t) {
        sub_aux_ptr[t] = (int)sub_aux_ref.size();
        sub_aux_ref.insert(sub_aux_ref.end(), taux[t].begin(), taux[t].end());
    }
    sub_aux_ptr[n_sub] = (int)sub_aux_ref.size();

    SVI_TIMING_MARK(8);
    // ---- upload ----
    d.Pn = Pn; d.Pf = Pf; d.Ll = Ll; d.E = E;
    d.n_lm_blocks = n_lm_blocks; d.n_chunks = n_chunks;
    d.n_se3 = (int)se3_i.size(); d.n_accel = (int)acc_pose.size(); d.n_lmlm = (int)ll_free.size();
    d.info_planes = planes;
    std::vector<double> hp((size_t)12 * Pn), hl((size_t)3 * Ll);
    for (int s = 0; s < Pn; ++s) memcpy(&hp[(size_t)12 * s], ba->poses[ba->pose_order[s]].T, 96);
    for (int l = 0; l < Ll; ++l) memcpy(&hl[(size_t)3 * l], ba->lms[ba->lm_order[L0 + l]].p, 24);
    for (int b = 0; b < 2; ++b) {
        const double* p = nullptr;
        SVI_TRY(dev_upload(ba, hp, &p)); d.pose[b] = const_cast<double*>(p);
        SVI_TRY(dev_upload(ba, hl, &p)); d.lm[b] = const_cast<double*>(p);
    }
    SVI_TRY(dev_upload(ba, pose_red, &d.pose_red));
    SVI_TRY(dev_upload(ba, lm_fixed, &d.lm_fixed));
    SVI_TRY(dev_upload(ba, e_pose, &d.e_pose));
    SVI_TRY(dev_upload(ba, e_lm, &d.e_lm));
    SVI_TRY(dev_upload(ba, e_flags, &d.e_flags));
    SVI_TRY(dev_upload(ba, e_zi, &d.e_zi));
    SVI_TRY(dev_upload(ba, lm_ptr, &d.lm_ptr));
    SVI_TRY(dev_upload(ba, lb_lm, &d.lb_lm));
    SVI_TRY(dev_upload(ba, pm_lm, &d.pm_lm));
    SVI_TRY(dev_upload(ba, pm_flags, &d.pm_flags));
    SVI_TRY(dev_upload(ba, pm_zi, &d.pm_zi));
    SVI_TRY(dev_upload(ba, chunk_pose, &d.chunk_pose));
    SVI_TRY(dev_upload(ba, chunk_begin, &d.chunk_begin));
    SVI_TRY(dev_upload(ba, pose_chunk_ptr, &d.pose_chunk_ptr));
    SVI_TRY(dev_upload(ba, se3_i, &d.se3_i));
    SVI_TRY(dev_upload(ba, se3_j, &d.se3_j));
    SVI_TRY(dev_upload(ba, se3_Z, &d.se3_Z));
    SVI_TRY(dev_upload(ba, se3_info, &d.se3_info));
    SVI_TRY(dev_upload(ba, se3_robust, &d.se3_robust));
    SVI_TRY(dev_upload(ba, acc_pose, &d.acc_pose));
    SVI_TRY(dev_upload(ba, acc_a, &d.acc_a));
    SVI_TRY(dev_upload(ba, acc_info, &d.acc_info));
    SVI_TRY(dev_upload(ba, ll_free, &d.ll_free));
    SVI_TRY(dev_upload(ba, ll_ref, &d.ll_ref));
    SVI_TRY(dev_upload(ba, ll_z, &d.ll_z));
    SVI_TRY(dev_upload(ba, ll_info, &d.ll_info));
    SVI_TRY(dev_upload(ba, ll_robust, &d.ll_robust));
    SVI_TRY(dev_upload(ba, lm_ll_ptr, &d.lm_ll_ptr));
    SVI_TRY(dev_upload(ba, pose_aux_ptr, &d.pose_aux_ptr));
    SVI_TRY(dev_upload(ba, pose_aux_ref, &d.pose_aux_ref));
    {
        const int* p = nullptr;
        SVI_TRY(dev_upload(ba, red_slot, &p)); ba->red_slot = const_cast<int*>(p);
        SVI_TRY(dev_upload(ba, e_orig, &p)); ba->e_orig = const_cast<int*>(p);
    }
    SVI_TRY(dev_alloc(ba, (size_t)12 * E, &d.NZ));
    SVI_TRY(dev_alloc(ba, (size_t)6 * Ll, &d.Hll));
    SVI_TRY(dev_alloc(ba, (size_t)3 * Ll, &d.bl));
    SVI_TRY(dev_alloc(ba, (size_t)6 * Ll, &d.Hinv));
    SVI_TRY(dev_alloc(ba, (size_t)12 * std::max(Ll, 1), &d.HinvB));
    SVI_TRY(dev_alloc(ba, (size_t)27 * n_chunks, &d.chunk_out));
    SVI_TRY(dev_alloc(ba, (size_t)120 * d.n_se3, &d.se3_out));
    SVI_TRY(dev_alloc(ba, (size_t)42 * d.n_accel, &d.acc_out));
    d.lin_count = 27 * Pf + 2 + o.n_ranks;
    SVI_TRY(dev_alloc(ba, (size_t)d.lin_count, &d.lin_buf));
    d.Hpp = d.lin_buf; d.bp = d.lin_buf + (size_t)21 * Pf; d.lin_scal = d.lin_buf + (size_t)27 * Pf;
    SVI_TRY(dev_alloc(ba, (size_t)16 * std::max(n_lm_blocks, 1), &d.block_part));
    d.TS = TS; d.NT = NT; d.n_tiles = n_tiles;
    SVI_TRY(dev_upload(ba, tile_map, &d.tile_map));
    // [ g | tiles with contributions | fill-in tiles ]: the all-reduce payload is the prefix g + contributing tiles
    d.red_count = NT * TS + n_tiles_orig * TS * TS;
    SVI_TRY(dev_alloc(ba, 2 + (size_t)NT * TS + (size_t)n_tiles * TS * TS, &d.red_base)); // two doubles in front: see linearize()
    d.g = d.red_base + 2;
    d.S = d.g + (size_t)NT * TS;
    SVI_TRY(dev_alloc(ba, (size_t)n_tiles * TS * TS, &d.Lt));
    SVI_TRY(dev_alloc(ba, (size_t)NT * TS * TS, &d.Linv));
    SVI_TRY(dev_alloc(ba, (size_t)NT * TS, &d.dx));
    SVI_TRY(dev_alloc(ba, 1, &d.chol_status));
    SVI_HIP(hipMemsetAsync(d.chol_status, 0, sizeof(int), ba->stream));
    d.n_items = n_items; d.n_jobs = n_jobs; d.n_sub = n_sub;
    SVI_TRY(dev_upload(ba, it_pack, &d.it_pack));
    SVI_TRY(dev_upload(ba, qj_begin, &d.qj_begin));
    SVI_TRY(dev_upload(ba, qj_end, &d.qj_end));
    SVI_TRY(dev_upload(ba, qj_diag, &d.qj_diag));
    SVI_TRY(dev_upload(ba, job_len, &d.job_len));
    SVI_TRY(dev_alloc(ba, (size_t)std::max(n_jobs, 1) * 36 * 64, &d.slab, false));
    SVI_TRY(dev_alloc(ba, (size_t)std::max(n_jobs, 1) * 4 * 6 * 4, &d.gslab, false));
    SVI_TRY(dev_upload(ba, cell_qj_ptr, &d.cell_qj_ptr));
    SVI_TRY(dev_upload(ba, cell_qj, &d.cell_qj));
    SVI_TRY(dev_upload(ba, sub_cx, &d.sub_cx));
    SVI_TRY(dev_upload(ba, sub_cy, &d.sub_cy));
    SVI_TRY(dev_upload(ba, sub_tile, &d.sub_tile));
    SVI_TRY(dev_upload(ba, sub_aux_ptr, &d.sub_aux_ptr));
    SVI_TRY(dev_upload(ba, sub_aux_ref, &d.sub_aux_ref));
    d.add_pose_terms = d.add_aux_blocks = (o.rank == 0) ? 1 : 0;
    d.lin_from_red = 0;
    SVI_TRY(dev_alloc(ba, 16, &d.scal));
    d.aux_blocks = std::max(1, (std::max(d.n_se3, d.n_accel) + 63) / 64);
    SVI_TRY(dev_alloc(ba, (size_t)2 * d.aux_blocks, &d.aux_part));
    SVI_TRY(dev_alloc(ba, 1, &d.aux_count));
    SVI_HIP(hipMemsetAsync(d.aux_count, 0, sizeof(int), ba->stream));
    if (o.n_ranks > 1) SVI_TRY(dev_alloc(ba, (size_t)3 * Ltot, &ba->lm_all));

    CholPlan& p = ba->plan;
    p.TS = TS; p.NT = NT; p.n_steps = n_steps;
    ba->h_step_ptr = h_step_ptr; ba->h_tgt_ptr = h_tgt_ptr; ba->h_trsm_ptr = h_trsm_ptr;
    p.h_step_ptr = ba->h_step_ptr.data(); p.h_tgt_ptr = ba->h_tgt_ptr.data(); p.h_trsm_ptr = ba->h_trsm_ptr.data();
    SVI_TRY(dev_upload(ba, h_col_ptr, &p.col_ptr));
    SVI_TRY(dev_upload(ba, trsm_tile, &p.trsm_tile));
    SVI_TRY(dev_upload(ba, trsm_row, &p.trsm_row));
    SVI_TRY(dev_upload(ba, step_col, &p.step_col));
    {
        // everything a chain / back-substitution workgroup needs to know about its column in ONE record (two 16-byte
        // loads side by side instead of a chain of three dependent index loads at the start of every launch)
        std::vector<int> step_desc((size_t)8 * std::max<size_t>(step_col.size(), 1), 0);
        for (size_t q = 0; q < step_col.size(); ++q) {
            const int c = step_col[q];
            int* r = &step_desc[8 * q];
            r[0] = c; r[1] = diag_tile[c]; r[2] = pre_ptr[c]; r[3] = pre_ptr[c + 1] - pre_ptr[c];
            r[4] = h_col_ptr[c]; r[5] = h_col_ptr[c + 1] - h_col_ptr[c];
        }
        SVI_TRY(dev_upload(ba, step_desc, &p.step_desc));
    }
    SVI_TRY(dev_upload(ba, diag_tile, &p.diag_tile));
    SVI_TRY(dev_upload(ba, pre_ptr, &p.pre_ptr));
    SVI_TRY(dev_upload(ba, pre_tile, &p.pre_tile));
    SVI_TRY(dev_upload(ba, pre_col, &p.pre_col));
    SVI_TRY(dev_upload(ba, tgt_tile, &p.tgt_tile));
    SVI_TRY(dev_upload(ba, tgt_row, &p.tgt_row));
    SVI_TRY(dev_upload(ba, tgt_pair_ptr, &p.tgt_pair_ptr));
    SVI_TRY(dev_upload(ba, pair_a, &p.pair_a));
    SVI_TRY(dev_upload(ba, pair_b, &p.pair_b));
    SVI_TRY(dev_upload(ba, pair_src, &p.pair_src));
    SVI_TRY(dev_upload(ba, st_tile, &p.st_tile));
    SVI_TRY(dev_upload(ba, st_col, &p.st_col));

    SVI_HIP(hipHostMalloc(reinterpret_cast<void**>(&ba->h_scal), 16 * sizeof(double)));
    SVI_HIP(hipHostMalloc(reinterpret_cast<void**>(&ba->h_status), sizeof(int) * 4));
    ba->h_status[0] = ba->h_status[1] = 0;
    ba->pub_seq = 0;
    ba_configure_kernels(TS);
    SVI_HIP(hipStreamSynchronize(ba->stream));

    svi_ba_stats& st = ba->stats;
    const uint64_t it0 = st.lm_iterations, tr0 = st.lm_trials, cf0 = st.chol_failures;
    st = svi_ba_stats{};
    st.lm_iterations = it0; st.lm_trials = tr0; st.chol_failures = cf0;
    st.n_poses = Pn; st.n_poses_free = Pf; st.n_landmarks = Ltot; st.n_landmarks_local = Ll;
    st.n_edges_proj = Etot; st.n_edges_proj_local = E;
    st.n_edges_se3 = (int64_t)ba->se3.size(); st.n_edges_accel = (int64_t)ba->acc.size(); st.n_edges_lmlm = (int64_t)ba->lmlm.size();
    st.n_schur_tiles = n_jobs; st.n_window_blocks = total_pairs;
    st.chol_n = n; st.chol_tile = TS; st.chol_tiles_nnz = n_tiles; st.chol_steps = n_steps;
    st.reduce_doubles = d.red_count;
    st.chol_flops = chol_flops;
    SVI_TIMING_MARK(9);
    return SVI_OK;
}

// ---------------------------------------------------------------------------------------------
// LM pieces
// ---------------------------------------------------------------------------------------------
// waits until the device has published sequence number `seq` (h_scal / h_status[0] are valid then)
int wait_published(svi_ba* ba, int seq)
{
    volatile int* flag = ba->h_status + 1;
    for (long spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spin & 0xFFF) == 0xFFF) { // every few thousand polls: has the stream died or drained without publishing?
            const hipError_t q = hipStreamQuery(ba->stream);
            if (q == hipSuccess) { if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break; SVI_HIP(hipStreamSynchronize(ba->stream)); if (*flag == seq) break; return fail(SVI_ERR_HIP, "result publication lost"); }
            if (q != hipErrorNotReady) return fail(SVI_ERR_HIP, "stream failed while waiting for the trial results: %s", hipGetErrorString(q));
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
int reduce_and_read_trial(svi_ba* ba, int n)
{
    if (ba->opt.n_ranks == 1 && !ba->timer.on) {
        const int seq = ++ba->pub_seq;
        ba_reduce_trial_scalars(ba->d, n, ba->h_scal, ba->h_status, seq, ba->stream);
        SVI_HIP(hipGetLastError());
        return wait_published(ba, seq);
    }
    ba_reduce_trial_scalars(ba->d, 0, nullptr, nullptr, 0, ba->stream);
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
    ba->sweep_timer.begin(0, s);
    t.begin(SVI_PH_LINEARIZE_LM, s);   ba_linearize_lm(d, ba->cur, s);   t.end(s);
    t.begin(SVI_PH_LINEARIZE_POSE, s); ba_linearize_pose(d, ba->cur, s); t.end(s);
    ba->sweep_timer.end(s);
    t.begin(SVI_PH_POSE_EDGES, s);
    ba_linearize_aux(d, ba->cur, ba->opt.rank, s);
    ba_pose_finalize(d, ba->red_slot, ba->opt.rank, ba->opt.n_ranks, s);
    t.end(s);
    SVI_HIP(hipGetLastError());
    // Several ranks: the pose sums (Hpp | bp | chi2 | max diag) only have to be exchanged when the host needs
    // max |H_jj| for lambda_0, i.e. in front of the first trial of a block.  Otherwise every rank adds its own partial
    // Hpp / bp to its share of the reduced system and the all-reduce of that system sums them (with this rank's chi2 in
    // the two doubles in front of g): one collective per iteration less.
    ba->lin_local = ba->opt.n_ranks > 1 && !read && d.n_sub > 0;
    if (ba->lin_local) return SVI_OK;
    SVI_TRY(allreduce(ba, d.lin_buf, (size_t)d.lin_count));
    ba_lin_post(d, ba->opt.n_ranks, s);
    SVI_HIP(hipGetLastError());
    return read ? read_scalars(ba, 8) : SVI_OK; // unread: the numbers wait in scal[8..10] for the next read
}

// one trial: solve (H + lambda I) dx = b through the Schur complement, apply, evaluate.
// results: h_scal[0] robust chi2, [1] plain chi2, [2]+[3] step scale; *failed
// k_assemble with the pose terms matching what linearize() left in Hpp / bp: totals (rank 0 adds them) or this rank's share
void assemble(svi_ba* ba)
{
    BaDev& d = ba->d;
    d.add_pose_terms = (ba->lin_local || ba->opt.rank == 0) ? 1 : 0;
    d.lin_from_red = ba->lin_local ? 1 : 0;
    ba_assemble(d, ba->stream);
}

int trial(svi_ba* ba, double lambda, bool* failed)
{
    BaDev& d = ba->d;
    hipStream_t s = ba->stream;
    PhaseTimer& t = ba->timer;
    // (the status word is clean: whoever read it last cleared it)
    t.begin(SVI_PH_SCHUR, s);
    ba_invert_landmarks(d, lambda, s);
    ba_schur(d, s);
    t.end(s);
    t.begin(SVI_PH_ASSEMBLE, s); assemble(ba); t.end(s);
    SVI_HIP(hipGetLastError());
    SVI_TRY(allreduce(ba, ba->lin_local ? d.red_base : d.g, (size_t)d.red_count + (ba->lin_local ? 2 : 0)));
    t.begin(SVI_PH_CHOLESKY, s);
    if (d.NT > 0 && chol_factor_solve(ba->plan, d.S, d.Lt, d.Linv, d.g, d.dx, lambda, 6 * d.Pf, d.chol_status, s) != 0)
        return fail(SVI_ERR_HIP, "Cholesky kernels could not be configured (LDS request refused)");
    t.end(s);
    t.begin(SVI_PH_BACKSUB_UPDATE, s);
    ba_update_poses(d, ba->cur, lambda, ba->opt.n_ranks <= 1 ? 0 : (ba->lin_local ? 2 : 1), ba->opt.rank, s);
    ba_backsub_chi2(d, ba->cur, lambda, s);
    t.end(s);
    t.begin(SVI_PH_CHI2, s);
    ba_chi2_aux(d, ba->cur ^ 1, ba->opt.rank, s);
    t.end(s);
    SVI_TRY(reduce_and_read_trial(ba, 12));
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
    SVI_HIP(hipSetDevice(ba->opt.device));
    SVI_TRY(download_state(ba));
    ba->host_stale = false;
    return SVI_OK;
}

namespace {

int optimize_block(svi_ba* ba, int iterations, int* performed)
{
    if (!ba->initialized) return fail(SVI_ERR_STATE, "svi_ba_optimize before svi_ba_initialize");
    SVI_HIP(hipSetDevice(ba->opt.device));
    const svi_ba_options& o = ba->opt;
    int done = 0;
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
            SVI_TRY(trial(ba, ba->lambda, &failed));
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
    // the estimates stay on the device; the host copy is refreshed by the first call that reads it (ensure_host)
    ba->host_stale = true;
    ba->timer.collect();
    // the last trial's scalars have been read: every sweep of this block has completed, its events can be queried
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
    HProj e{};
    e.type = type; e.robust = robust ? 1 : 0; e.pose = ip->second; e.lm = il->second;
    memcpy(e.z, z, 24); memcpy(e.info, info, 48);
    ba->proj.push_back(e);
    ba->initialized = false;
    return SVI_OK;
}

} // namespace

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
    o->chol_tile = 96;
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
    if (ba->opt.chol_tile == 0) ba->opt.chol_tile = 96;
    if (o->stream) ba->stream = static_cast<hipStream_t>(o->stream);
    else {
        hipError_t e = hipStreamCreateWithFlags(&ba->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ba; return fail(SVI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        ba->own_stream = true;
    }
    ba->timer.on = o->profile != 0;
    ba->sweep_timer.on = o->sweep_events != 0 && o->profile == 0;
    *out = ba;
    return SVI_OK;
}

int svi_ba_destroy(svi_ba* ba)
{
    if (!ba) return SVI_OK;
    (void)hipSetDevice(ba->opt.device);
    (void)hipStreamSynchronize(ba->stream);
    free_device(ba);
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
    for (int64_t i = 0; i < n; ++i) {
        if (type[i] < 0 || type[i] > 2) { ba->proj.resize(before); return fail(SVI_ERR_INVALID, "edge %lld: unknown type %d", (long long)i, type[i]); }
        const int rc = add_proj(ba, type[i], pose_id[i], lm_id[i], z + 3 * i, info + 6 * i, robust ? robust[i] : 1);
        if (rc != SVI_OK) { ba->proj.resize(before); return rc; }
    }
    return SVI_OK;
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
    return SVI_OK;
}

int svi_ba_initialize(svi_ba* ba)
{
    if (!ba) return fail(SVI_ERR_INVALID, "null handle");
    SVI_TRY(ensure_host(ba));
    if (int rc = use_device(ba->opt.device)) return rc;
    free_device(ba);
    ba->cur = 0;
    ba->have_chi = false;
    const int rc = build_structure(ba);
    if (rc != SVI_OK) { free_device(ba); return rc; }
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
        SVI_HIP(hipSetDevice(ba->opt.device));
        ba_chi2_only(ba->d, ba->cur, ba->stream);
        ba_chi2_aux(ba->d, ba->cur, ba->opt.rank, ba->stream);
        SVI_TRY(reduce_and_read_trial(ba, 8));
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
        std::vector<HProj> pe;
        pe.reserve(ba->proj.size());
        for (HProj e : ba->proj) if (remap[e.lm] >= 0) { e.lm = remap[e.lm]; pe.push_back(e); }
        ba->proj.swap(pe);
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
    SVI_HIP(hipSetDevice(ba->opt.device));
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
    SVI_HIP(hipSetDevice(ba->opt.device));
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
    SVI_HIP(hipSetDevice(ba->opt.device));
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
    SVI_HIP(hipSetDevice(ba->opt.device));
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
    SVI_HIP(hipSetDevice(ba->opt.device));
    BaDev& d = ba->d;
    const int64_t n = 6 * (int64_t)d.Pf;
    *n_out = n;
    if (cap < n) return fail(SVI_ERR_INVALID, "capacity %lld < n %lld", (long long)cap, (long long)n);
    SVI_TRY(linearize(ba));
    SVI_HIP(hipMemsetAsync(d.chol_status, 0, sizeof(int), ba->stream));
    ba_invert_landmarks(d, lambda, ba->stream);
    ba_schur(d, ba->stream);
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
