// ba_structure.cpp — svi_ba_initialize: the structure analysis behind g2o's initializeOptimization + buildStructure
// (Cg2oOptimizer.cpp:957 runs it at the top of EVERY optimize() call, so it is part of what one call costs).
//
// Pieces, in order (each a function below, sharing one Build context):
//   order_vertices      ascending id = g2o's index mapping; reduced (free) poses
//   sort_edges          projection edges by (landmark, pose): one counting sort over compact keys -> CSR per landmark
//   elimination_order   nested dissection of the key-frame sequence, candidates evaluated symbolically at tile level
//   local_edges         this rank's landmark range, lm-major order (landmark, elimination index), pose-major copy
//   aux_edges           odometry / gravity / landmark-closure edges
//   tile_structure      tiles of the reduced system, symbolic fill, dependency levels, grouped updates
//   schur_work_lists    cells, items, quarter jobs, slabs
//   upload              device buffers (kept across calls, re-allocated only when they grow)
//
// What keeps it short of the LM loop's own time (config 4: 800 k edges, 100 k landmarks):
//   * no per-landmark containers: every per-landmark list is a range of the sorted edge array;
//   * the edge VALUES (z, information: 72 bytes per edge) never pass through the host again: they live in a device-side
//     log in insertion order, only the edges added since the last call are uploaded (through pinned staging), and two
//     gather kernels lay them out lm-major and pose-major from the permutations computed here;
//   * device buffers are re-used between calls; the big loops run on a few threads.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <thread>

#include "ba_host.h"
#include "ba_math.h"

namespace svi {
void ba_gather_edges(const double* raw, const uint8_t* raw_flags, const int* src, const int* via, int count, int stride, int planes, double* out_zi,
                     uint8_t* out_flags, const int* lm_in, int* lm_out, void* st);
void ba_configure_kernels(int TS);
}

using namespace svi;

namespace {

#define SVI_TRY(x) do { int rc_ = (x); if (rc_ != SVI_OK) return rc_; } while (0)

// f(begin, end) over contiguous chunks of [0, n) on up to 8 threads (the caller's thread takes the first chunk)
size_t max_threads()
{
    static const size_t n = []() {
        const char* e = getenv("SVI_HOST_THREADS");
        if (e && atoi(e) > 0) return (size_t)atoi(e);
        unsigned hw = std::thread::hardware_concurrency();
        return std::min<size_t>(hw ? hw : 1, 8);
    }();
    return n;
}

template <class F> void parallel_chunks(size_t n, size_t min_chunk, F&& f)
{
    size_t nt = std::min<size_t>(max_threads(), std::max<size_t>(n / std::max<size_t>(min_chunk, 1), 1));
    if (nt <= 1) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 1; t < nt; ++t) {
        const size_t a = std::min(n, t * per), b = std::min(n, (t + 1) * per);
        if (a < b) th.emplace_back([&f, a, b]() { f(a, b); });
    }
    f((size_t)0, std::min(n, per));
    for (auto& x : th) x.join();
}

struct Build {
    svi_ba* ba;
    bool dbg;
    std::chrono::steady_clock::time_point t0;
    int Pn = 0, Pf = 0, Ltot = 0;
    int64_t Etot = 0;
    std::vector<int> pose_slot, pose_red, red_slot, lm_slot;
    // all projection edges sorted by (landmark slot, pose slot, insertion): insertion index and pose slot, CSR per landmark slot
    std::vector<int> g_edge, g_pose, g_ptr;
    std::vector<int> tmp_ls, tmp_ps, tmp_cur; // sort_edges' scratch (members: they keep their memory between calls)
    // which free poses share a landmark: lower triangle over NATURAL reduced indices (only for Pf <= 4096: 16 MB), longest track
    std::vector<uint8_t> cpl;
    std::vector<int> cpl_lo;    // first coupled column of every row of cpl (the matrix is a band: scans start there)
    bool use_cpl = false;
    int span = 0;
    // this rank
    int L0 = 0, Ll = 0, E = 0, Epm = 0;
    std::vector<int> loc, e_pose, e_lm, lm_ptr, lb_lm, lb_rec, pm, pm_src, chunk_pose, chunk_begin, pose_chunk_ptr;
    std::vector<uint8_t> lm_fixed;
    int n_lm_blocks = 0, n_chunks = 0, planes = 3;
    // aux
    std::vector<int> se3_i, se3_j, acc_pose, ll_free, lm_ll_ptr, pose_aux_ptr, pose_aux_ref;
    std::vector<double> se3_Z, se3_info, acc_a, acc_info, ll_ref, ll_z, ll_info;
    std::vector<uint8_t> se3_robust, ll_robust;
    // tiles
    int TS = 96, PB = 16, n = 0, NT = 0, n_tiles = 0, n_tiles_orig = 0, n_steps = 0;
    std::vector<int> tile_map, tile_ti, tile_tj, h_col_ptr, trsm_tile, trsm_row, diag_tile, level, h_step_ptr, step_col, pre_ptr, pre_tile, pre_col,
        h_tgt_ptr, tgt_tile, tgt_row, tgt_pair_ptr, pair_a, pair_b, pair_src, h_trsm_ptr, st_tile, st_col;
    double chol_flops = 0.0;
    // Schur
    int n_sub = 0, n_items = 0, n_jobs = 0; // n_jobs: wavefront jobs PER STAGE (the kernel's grid), n_stages of them back to back
    int n_stages = 1, schur_wgs = 0, schur_reserve_per_se = 0;
    std::vector<int> qj_cell, orphan_ptr, orphan_cell; // cell of every quarter-job slot (-1: empty); per stage the cells no slab reaches
    std::vector<int> level_stage, sub_stage_ptr; // stage of every dependency level; sub-tiles [sub_stage_ptr[s], sub_stage_ptr[s+1]) belong to stage s
    int64_t total_pairs = 0;
    std::vector<int> sub_cx, sub_cy, sub_tile, it_pack, qj_begin, qj_end, qj_diag, job_len, job_merged, cell_qj_ptr, cell_qj, sub_aux_ptr, sub_aux_ref;

    // the context lives in the handle: the vectors keep their memory from one svi_ba_initialize to the next (a fresh 15 MB
    // vector costs up to 3 ms of page faults on a busy box), so every one of them is emptied here (list generated from the
    // member declarations above)
    void clear_all()
    {
        pose_slot.clear(); pose_red.clear(); red_slot.clear(); lm_slot.clear(); g_edge.clear(); g_pose.clear(); g_ptr.clear();
        tmp_ls.clear(); tmp_ps.clear(); tmp_cur.clear(); cpl.clear(); cpl_lo.clear(); loc.clear(); e_pose.clear();
        e_lm.clear(); lm_ptr.clear(); lb_lm.clear(); lb_rec.clear(); pm.clear(); pm_src.clear(); chunk_pose.clear(); chunk_begin.clear();
        pose_chunk_ptr.clear(); lm_fixed.clear(); se3_i.clear(); se3_j.clear(); acc_pose.clear(); ll_free.clear();
        lm_ll_ptr.clear(); pose_aux_ptr.clear(); pose_aux_ref.clear(); se3_Z.clear(); se3_info.clear(); acc_a.clear();
        acc_info.clear(); ll_ref.clear(); ll_z.clear(); ll_info.clear(); se3_robust.clear(); ll_robust.clear();
        tile_map.clear(); tile_ti.clear(); tile_tj.clear(); h_col_ptr.clear(); trsm_tile.clear(); trsm_row.clear();
        diag_tile.clear(); level.clear(); h_step_ptr.clear(); step_col.clear(); pre_ptr.clear(); pre_tile.clear();
        pre_col.clear(); h_tgt_ptr.clear(); tgt_tile.clear(); tgt_row.clear(); tgt_pair_ptr.clear(); pair_a.clear();
        pair_b.clear(); pair_src.clear(); h_trsm_ptr.clear(); st_tile.clear(); st_col.clear(); sub_cx.clear(); sub_cy.clear();
        sub_tile.clear(); it_pack.clear(); qj_begin.clear(); qj_end.clear(); qj_diag.clear(); job_len.clear(); job_merged.clear();
        cell_qj_ptr.clear(); cell_qj.clear(); sub_aux_ptr.clear(); sub_aux_ref.clear(); level_stage.clear(); sub_stage_ptr.clear(); qj_cell.clear(); orphan_ptr.clear(); orphan_cell.clear();
        n_stages = 1; schur_wgs = 0; schur_reserve_per_se = 0;
        Pn = Pf = Ltot = 0; Etot = 0; use_cpl = false; span = 0; L0 = Ll = E = Epm = 0; n_lm_blocks = n_chunks = 0; planes = 3;
        TS = 96; PB = 16; n = NT = n_tiles = n_tiles_orig = n_steps = 0; chol_flops = 0.0; n_sub = n_items = n_jobs = 0; total_pairs = 0;
    }
    void mark(int k) const
    {
        if (dbg) fprintf(stderr, "build_structure: section %d starts at %.2f ms\n", k, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ---- device buffers that survive re-initialisation ----------------------------------------------------------------------
struct Uploader {
    svi_ba* ba;
    size_t next = 0;
    DevBuf& slot()
    {
        if (next == ba->pool.size()) ba->pool.emplace_back();
        return ba->pool[next++];
    }
    template <class T> int up(const std::vector<T>& h, const T** out, size_t min_elems = 1)
    {
        DevBuf& b = slot();
        const size_t cnt = std::max(h.size(), min_elems);
        SVI_TRY(b.reserve(cnt * sizeof(T)));
        if (!h.empty()) SVI_HIP(hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ba->stream));
        *out = b.as<T>();
        return SVI_OK;
    }
    template <class T> int alloc(size_t cnt, T** out, bool zero = true)
    {
        DevBuf& b = slot();
        cnt = std::max<size_t>(cnt, 1);
        SVI_TRY(b.reserve(cnt * sizeof(T)));
        if (zero) SVI_HIP(hipMemsetAsync(b.p, 0, cnt * sizeof(T), ba->stream));
        *out = b.as<T>();
        return SVI_OK;
    }
};

// ---- vertex order: ascending id (g2o index mapping) ---------------------------------------------------------------------
void order_vertices(Build& b)
{
    svi_ba* ba = b.ba;
    b.Pn = (int)ba->poses.size();
    b.Ltot = (int)ba->lms.size();
    ba->pose_order.resize(b.Pn);
    std::iota(ba->pose_order.begin(), ba->pose_order.end(), 0);
    std::sort(ba->pose_order.begin(), ba->pose_order.end(), [&](int x, int y) { return ba->poses[x].id < ba->poses[y].id; });
    b.pose_slot.resize(b.Pn);
    b.pose_red.resize(b.Pn);
    b.red_slot.clear();
    for (int s = 0; s < b.Pn; ++s) b.pose_slot[ba->pose_order[s]] = s;
    for (int s = 0; s < b.Pn; ++s) {
        if (ba->poses[ba->pose_order[s]].fixed) b.pose_red[s] = -1;
        else { b.pose_red[s] = (int)b.red_slot.size(); b.red_slot.push_back(s); }
    }
    b.Pf = (int)b.red_slot.size();
    ba->lm_order.resize(b.Ltot);
    std::iota(ba->lm_order.begin(), ba->lm_order.end(), 0);
    bool sorted = true; // landmarks usually arrive in ascending id (the reference adds them as they are created)
    for (int i = 1; i < b.Ltot && sorted; ++i) sorted = ba->lms[i - 1].id < ba->lms[i].id;
    if (!sorted) std::sort(ba->lm_order.begin(), ba->lm_order.end(), [&](int x, int y) { return ba->lms[x].id < ba->lms[y].id; });
    b.lm_slot.resize(b.Ltot);
    for (int s = 0; s < b.Ltot; ++s) b.lm_slot[ba->lm_order[s]] = s;
}

// ---- all projection edges by (landmark slot, pose slot, insertion order) ------------------------------------------------
void sort_edges(Build& b)
{
    svi_ba* ba = b.ba;
    const size_t E = ba->proj.size();
    b.Etot = (int64_t)E;
    std::vector<int>&ls = b.tmp_ls, &ps = b.tmp_ps;
    ls.resize(E); ps.resize(E);
    parallel_chunks(E, 1 << 16, [&](size_t a, size_t z) {
        for (size_t i = a; i < z; ++i) { ls[i] = b.lm_slot[ba->proj.lm[i]]; ps[i] = b.pose_slot[ba->proj.pose[i]]; }
    });
    b.planes = ba->proj.n_offdiag == 0 ? 3 : 6;
    b.g_ptr.assign((size_t)b.Ltot + 1, 0);
    for (size_t i = 0; i < E; ++i) b.g_ptr[ls[i] + 1]++;
    for (int l = 0; l < b.Ltot; ++l) b.g_ptr[l + 1] += b.g_ptr[l];
    b.g_edge.resize(E);
    b.g_pose.resize(E);
    {
        std::vector<int>& cur = b.tmp_cur;
        cur.assign(b.g_ptr.begin(), b.g_ptr.end() - 1);
        for (size_t i = 0; i < E; ++i) { const int k = cur[ls[i]]++; b.g_edge[k] = (int)i; b.g_pose[k] = ps[i]; }
    }
    // inside a landmark: by pose slot, ties in insertion order (the segments are short and nearly always sorted already)
    parallel_chunks((size_t)b.Ltot, 1 << 13, [&](size_t la, size_t lz) {
        for (size_t l = la; l < lz; ++l)
            for (int i = b.g_ptr[l] + 1; i < b.g_ptr[l + 1]; ++i) {
                const int p = b.g_pose[i], e = b.g_edge[i];
                int j = i - 1;
                while (j >= b.g_ptr[l] && b.g_pose[j] > p) { b.g_pose[j + 1] = b.g_pose[j]; b.g_edge[j + 1] = b.g_edge[j]; --j; }
                b.g_pose[j + 1] = p; b.g_edge[j + 1] = e;
            }
    });
}

// ---- co-visibility of the free poses, from the sorted edge ranges ----------------------------------------------------------
void pose_coupling(Build& b)
{
    svi_ba* ba = b.ba;
    const int Pf = b.Pf;
    b.use_cpl = Pf <= 4096; // longer sequences walk the landmarks where the matrix would be read
    b.cpl.assign(b.use_cpl ? (size_t)Pf * Pf : 0, 0);
    b.span = 0;
    b.cpl_lo.resize(b.use_cpl ? (size_t)Pf : 0);
    for (int r = 0; r < (int)b.cpl_lo.size(); ++r) b.cpl_lo[r] = r;
    std::vector<int> reds; // free poses of one landmark, ascending (pose slots ascend, so do natural reduced indices)
    // A feature track is a run of consecutive key frames, and many landmarks share theirs: a run [first, last] that has been
    // entered once is skipped (bit (last - first) of seen[first]; longer or broken runs are always entered).
    std::vector<uint64_t> seen(b.use_cpl ? (size_t)Pf : 0, 0);
    for (int l = 0; l < b.Ltot; ++l) {
        int k0 = b.g_ptr[l];
        const int k1 = b.g_ptr[l + 1];
        while (k0 < k1 && b.pose_red[b.g_pose[k0]] < 0) ++k0; // (fixed poses have the lowest slots: their edges come first)
        if (k0 == k1) continue;
        const int first = b.pose_red[b.g_pose[k0]], len = b.pose_red[b.g_pose[k1 - 1]] - first;
        const bool run = len + 1 == k1 - k0 && len < 64; // as many edges as poses between the first and the last: consecutive
        if (b.use_cpl && run && (seen[first] >> len & 1) && len <= b.span) continue; // (first and last edge only: the common case)
        if (ba->lms[ba->lm_order[l]].fixed) continue;
        b.span = std::max(b.span, len);
        if (!b.use_cpl) continue;
        if (run) seen[first] |= (uint64_t)1 << len;
        reds.clear();
        for (int k = k0; k < k1; ++k) reds.push_back(b.pose_red[b.g_pose[k]]);
        for (size_t x = 0; x < reds.size(); ++x) {
            uint8_t* row = &b.cpl[(size_t)reds[x] * Pf];
            for (size_t y = 0; y <= x; ++y) row[reds[y]] = 1;
            b.cpl_lo[reds[x]] = std::min(b.cpl_lo[reds[x]], first);
        }
    }
}

// ---- elimination order of the reduced camera system (nested dissection of the key-frame sequence) ----------------------
// The reduced system of a trajectory is block-banded: in natural order its Cholesky is ONE chain of tile columns.  Cutting
// the sequence at separators as wide as the co-visibility span gives independent chains that are factorised side by side
// (ba_chol.hip processes all columns of one dependency level per launch).  Every piece is a whole number of tiles except
// the top separator, which comes last and absorbs the remainder, so the identity padding stays at the end of the range.
void elimination_order(Build& b)
{
    svi_ba* ba = b.ba;
    const svi_ba_options& o = ba->opt;
    const int Pf = b.Pf;
    ba->red_perm.resize(Pf);
    std::iota(ba->red_perm.begin(), ba->red_perm.end(), 0);
    const int TSo = o.chol_tile > 0 ? o.chol_tile : 48;
    const int PBo = TSo / 6;
    const int NTo = PBo > 0 ? (Pf + PBo - 1) / PBo : 0;
    if (!(o.chol_order == 0 && PBo > 0 && NTo >= 6)) return;
    const bool use_cpl = b.use_cpl;
    const std::vector<uint8_t>& cpl = b.cpl;
    int span = b.span;
    std::vector<std::pair<int, int>> pp;
    for (const HSe3& e : ba->se3) {
        const int ri = b.pose_red[b.pose_slot[e.i]], rj = b.pose_red[b.pose_slot[e.j]];
        if (ri >= 0 && rj >= 0) { pp.push_back({ri, rj}); span = std::max(span, std::abs(ri - rj)); }
    }
    const int rem = Pf % PBo;
    // elimination order for separators of `w` tiles; false if the sequence is too short for it
    auto make_perm = [&](int w, std::vector<int>& perm) {
        const int sep = w * PBo;
        const int sep_top = rem == 0 ? sep : rem + PBo * ((std::max(sep - rem, 0) + PBo - 1) / PBo); // absorbs the remainder
        if (Pf < sep_top + 4 * PBo) return false;
        std::vector<std::pair<int, int>> pieces; // natural ranges in elimination order
        std::function<void(int, int, int)> rec = [&](int a, int z, int wd) {
            const int len = z - a;
            if (len < wd + 2 * PBo) { if (len > 0) pieces.push_back({a, z}); return; }
            const int left = PBo * (((len - wd) / PBo) / 2);
            rec(a, a + left, sep);
            rec(a + left + wd, z, sep);
            pieces.push_back({a + left, a + left + wd});
        };
        rec(0, Pf, sep_top);
        perm.assign(Pf, 0);
        int pos = 0;
        for (auto& pc : pieces) for (int r = pc.first; r < pc.second; ++r) perm[r] = pos++;
        return true;
    };
    // the coupling at TILE level in natural order, computed once: a candidate only re-maps tile indices when its pieces are
    // whole tiles - they are not (the top separator absorbs the remainder), so the pose-level matrix is re-mapped per candidate
    auto analyse = [&](const std::vector<int>& perm, int& depth, int& tiles) {
        std::vector<uint8_t> z((size_t)NTo * NTo, 0);
        for (int t = 0; t < NTo; ++t) z[(size_t)t * NTo + t] = 1;
        if (use_cpl) {
            for (int ri = 0; ri < Pf; ++ri) {
                const int tx = perm[ri] / PBo;
                const uint8_t* row = &cpl[(size_t)ri * Pf];
                for (int rj = b.cpl_lo[ri]; rj <= ri; ++rj) // (nothing is coupled in front of cpl_lo: the matrix is a band)
                    if (row[rj]) { const int ty = perm[rj] / PBo; z[(size_t)std::max(tx, ty) * NTo + std::min(tx, ty)] = 1; }
            }
        } else {
            std::vector<int> v;
            for (int l = 0; l < b.Ltot; ++l) {
                if (ba->lms[ba->lm_order[l]].fixed) continue;
                v.clear();
                for (int k = b.g_ptr[l]; k < b.g_ptr[l + 1]; ++k) { const int r = b.pose_red[b.g_pose[k]]; if (r >= 0) v.push_back(perm[r] / PBo); }
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
    // candidates: natural order and separators of 1 .. ceil(span / tile) tiles (a separator narrower than the longest track
    // still gives a valid order - the few tracks that cross it only add dependencies); fewest levels, then fewest tiles.
    // The candidates are independent: one thread each.
    const int wmax = std::max(1, (span + PBo - 1) / PBo);
    const int w_first = NTo > 256 ? std::min(wmax, 8) : 1, w_last = std::min(wmax, 8);
    struct Cand { std::vector<int> perm; int depth = 0, tiles = 0; bool ok = false; };
    std::vector<Cand> cands((size_t)std::max(w_last - w_first + 1, 0) + 1);
    cands[0].perm = ba->red_perm; cands[0].ok = true;
    for (int w = w_first; w <= w_last; ++w) {
        Cand& c = cands[(size_t)(w - w_first) + 1];
        c.ok = make_perm(w, c.perm);
        if (!c.ok) break;
    }
    {
        std::vector<std::thread> th;
        for (size_t i = 1; i < cands.size(); ++i)
            if (cands[i].ok) th.emplace_back([&, i]() { analyse(cands[i].perm, cands[i].depth, cands[i].tiles); });
        analyse(cands[0].perm, cands[0].depth, cands[0].tiles);
        for (auto& x : th) x.join();
    }
    size_t best = 0;
    for (size_t i = 1; i < cands.size(); ++i)
        if (cands[i].ok && (cands[i].depth < cands[best].depth || (cands[i].depth == cands[best].depth && cands[i].tiles < cands[best].tiles))) best = i;
    ba->red_perm = cands[best].perm;
    for (int sl = 0; sl < b.Pn; ++sl) if (b.pose_red[sl] >= 0) b.pose_red[sl] = ba->red_perm[b.pose_red[sl]];
    for (int sl = 0; sl < b.Pn; ++sl) if (b.pose_red[sl] >= 0) b.red_slot[b.pose_red[sl]] = sl;
}

// ---- this rank's edges: lm-major (landmark, elimination index [fixed poses first], pose slot, insertion), pose-major copy ----
int local_edges(Build& b)
{
    svi_ba* ba = b.ba;
    const svi_ba_options& o = ba->opt;
    // landmark sharding: contiguous slot ranges balanced by projection-edge count
    auto bound = [&](int r) -> int {
        if (r <= 0) return 0;
        if (r >= o.n_ranks) return b.Ltot;
        const int64_t want = b.Etot * r / o.n_ranks;
        int s = (int)(std::lower_bound(b.g_ptr.begin(), b.g_ptr.end(), (int)want) - b.g_ptr.begin());
        return std::min(std::max(s, 0), b.Ltot);
    };
    ba->E_total = b.Etot;
    ba->L0 = bound(o.rank);
    ba->L1 = bound(o.rank + 1);
    b.L0 = ba->L0;
    b.Ll = ba->L1 - ba->L0;
    const int e0 = b.g_ptr[b.L0], e1 = b.g_ptr[ba->L1];
    b.E = e1 - e0;
    const int E = b.E, Ll = b.Ll, Pn = b.Pn;
    std::vector<int> prank(Pn), pidx(Pn);
    for (int s = 0; s < Pn; ++s) pidx[s] = s;
    std::sort(pidx.begin(), pidx.end(), [&](int x, int y) { return b.pose_red[x] != b.pose_red[y] ? b.pose_red[x] < b.pose_red[y] : x < y; });
    for (int r = 0; r < Pn; ++r) prank[pidx[r]] = r;
    b.loc.assign(b.g_edge.begin() + e0, b.g_edge.begin() + e1);
    b.e_pose.assign(b.g_pose.begin() + e0, b.g_pose.begin() + e1);
    b.e_lm.resize(E);
    b.lm_ptr.resize((size_t)Ll + 1);
    for (int l = 0; l <= Ll; ++l) b.lm_ptr[l] = b.g_ptr[b.L0 + l] - e0;
    int bad_lm = -1;
    parallel_chunks((size_t)Ll, 1 << 13, [&](size_t la, size_t lz) {
        for (size_t l = la; l < lz; ++l) {
            const int a = b.lm_ptr[l], z = b.lm_ptr[l + 1];
            if (z - a > kLmBlockEdges) { __atomic_store_n(&bad_lm, (int)l, __ATOMIC_RELAXED); continue; }
            for (int i = a; i < z; ++i) b.e_lm[i] = (int)l;
            for (int i = a + 1; i < z; ++i) { // by elimination rank; equal ranks cannot occur for distinct poses, ties keep insertion order
                const int p = b.e_pose[i], e = b.loc[i], key = prank[p];
                int j = i - 1;
                while (j >= a && prank[b.e_pose[j]] > key) { b.e_pose[j + 1] = b.e_pose[j]; b.loc[j + 1] = b.loc[j]; --j; }
                b.e_pose[j + 1] = p; b.loc[j + 1] = e;
            }
        }
    });
    if (bad_lm >= 0)
        return fail(SVI_ERR_UNSUPPORTED, "landmark %lld has %d projection edges (limit %d)", (long long)ba->lms[ba->lm_order[b.L0 + bad_lm]].id,
                    b.lm_ptr[bad_lm + 1] - b.lm_ptr[bad_lm], kLmBlockEdges);
    // duplicate (pose, landmark) edges would alias one 6x6 block inside a Schur item; the reference never creates them (one
    // measurement per landmark per keyframe), reject instead of mis-summing
    for (int l = 0; l < Ll; ++l)
        for (int a = b.lm_ptr[l] + 1; a < b.lm_ptr[l + 1]; ++a)
            if (b.e_pose[a] == b.e_pose[a - 1])
                return fail(SVI_ERR_UNSUPPORTED, "two projection edges between pose %lld and landmark %lld",
                            (long long)ba->poses[ba->pose_order[b.e_pose[a]]].id, (long long)ba->lms[ba->lm_order[b.L0 + l]].id);
    b.lb_lm.assign(1, 0);
    for (int l = 0; l < Ll;) {
        int l2 = l, edges = 0;
        while (l2 < Ll && l2 - l < kLmBlockEdges && edges + (b.lm_ptr[l2 + 1] - b.lm_ptr[l2]) <= kLmBlockEdges) { edges += b.lm_ptr[l2 + 1] - b.lm_ptr[l2]; ++l2; }
        b.lb_lm.push_back(l2);
        l = l2;
    }
    b.n_lm_blocks = (int)b.lb_lm.size() - 1;
    // one 16-byte record per workgroup {first landmark, #landmarks, first edge, end edge}: lb_lm -> lm_ptr was two dependent
    // round trips at the head of every landmark-major workgroup
    b.lb_rec.resize((size_t)4 * std::max(b.n_lm_blocks, 1), 0);
    for (int k = 0; k < b.n_lm_blocks; ++k) {
        const int l0 = b.lb_lm[k], l1 = b.lb_lm[k + 1];
        b.lb_rec[4 * k] = l0; b.lb_rec[4 * k + 1] = l1 - l0; b.lb_rec[4 * k + 2] = b.lm_ptr[l0]; b.lb_rec[4 * k + 3] = b.lm_ptr[l1];
    }
    b.lm_fixed.resize(Ll);
    for (int l = 0; l < Ll; ++l) b.lm_fixed[l] = (uint8_t)(ba->lms[ba->lm_order[b.L0 + l]].fixed ? 1 : 0);
    // pose-major copy: free poses only, (pose slot, lm-major position): a stable counting sort
    std::vector<int> cnt((size_t)Pn + 1, 0);
    for (int k = 0; k < E; ++k) if (b.pose_red[b.e_pose[k]] >= 0) cnt[b.e_pose[k] + 1]++;
    for (int s = 0; s < Pn; ++s) cnt[s + 1] += cnt[s];
    b.Epm = cnt[Pn];
    b.pm.resize(b.Epm);
    {
        std::vector<int> cur(cnt.begin(), cnt.end() - 1);
        for (int k = 0; k < E; ++k) if (b.pose_red[b.e_pose[k]] >= 0) b.pm[cur[b.e_pose[k]]++] = k;
    }
    b.chunk_pose.clear(); b.chunk_begin.clear();
    b.pose_chunk_ptr.assign((size_t)Pn + 1, 0);
    for (int s = 0; s < Pn; ++s) {
        b.pose_chunk_ptr[s] = (int)b.chunk_pose.size();
        for (int k = cnt[s]; k < cnt[s + 1]; k += kPoseChunk) { b.chunk_pose.push_back(s); b.chunk_begin.push_back(k); }
    }
    b.pose_chunk_ptr[Pn] = (int)b.chunk_pose.size();
    b.chunk_begin.push_back(b.Epm); // chunks are contiguous: chunk_begin[c+1] is the end of chunk c
    b.n_chunks = (int)b.chunk_pose.size();
    return SVI_OK;
}

// ---- pose-only / landmark-only edges -------------------------------------------------------------------------------------
int aux_edges(Build& b)
{
    svi_ba* ba = b.ba;
    const svi_ba_options& o = ba->opt;
    const int Pn = b.Pn, Ll = b.Ll, L0 = b.L0;
    std::vector<std::vector<int>> pose_aux(Pn);
    if (o.rank == 0) {
        for (const HSe3& e : ba->se3) {
            const int k = (int)b.se3_i.size();
            b.se3_i.push_back(b.pose_slot[e.i]); b.se3_j.push_back(b.pose_slot[e.j]);
            b.se3_Z.insert(b.se3_Z.end(), e.Z, e.Z + 12);
            b.se3_info.insert(b.se3_info.end(), e.info, e.info + 21);
            b.se3_robust.push_back((uint8_t)(e.robust ? 1 : 0));
            pose_aux[b.pose_slot[e.i]].push_back((k << 2) | 0);
            pose_aux[b.pose_slot[e.j]].push_back((k << 2) | 1);
        }
        for (const HAcc& e : ba->acc) {
            const int k = (int)b.acc_pose.size();
            b.acc_pose.push_back(b.pose_slot[e.pose]);
            for (int r = 0; r < 3; ++r) b.acc_a.push_back(e.off[3 * r] * e.a[0] + e.off[3 * r + 1] * e.a[1] + e.off[3 * r + 2] * e.a[2]);
            b.acc_info.insert(b.acc_info.end(), e.info, e.info + 6);
            pose_aux[b.pose_slot[e.pose]].push_back((k << 2) | 2);
        }
    }
    b.pose_aux_ptr.assign((size_t)Pn + 1, 0);
    for (int s = 0; s < Pn; ++s) {
        b.pose_aux_ptr[s] = (int)b.pose_aux_ref.size();
        b.pose_aux_ref.insert(b.pose_aux_ref.end(), pose_aux[s].begin(), pose_aux[s].end());
    }
    b.pose_aux_ptr[Pn] = (int)b.pose_aux_ref.size();
    b.pose_aux_ref.insert(b.pose_aux_ref.end(), 4, 0); // (pose_finalize_block reads three references ahead of a list's start)
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
        const HLm& fx = free_is_j ? li : lj;
        const int s = b.lm_slot[free_is_j ? e.j : e.i];
        if (s < L0 || s >= L0 + Ll) continue;
        x.free_l = s - L0;
        for (int c = 0; c < 3; ++c) { x.ref[c] = fx.p[c]; x.z[c] = free_is_j ? e.z[c] : -e.z[c]; }
        memcpy(x.info, e.info, sizeof(x.info));
        x.robust = (uint8_t)(e.robust ? 1 : 0);
        v.push_back(x);
    }
    std::stable_sort(v.begin(), v.end(), [](const LL& x, const LL& y) { return x.free_l < y.free_l; });
    for (const LL& x : v) {
        b.ll_free.push_back(x.free_l);
        b.ll_ref.insert(b.ll_ref.end(), x.ref, x.ref + 3);
        b.ll_z.insert(b.ll_z.end(), x.z, x.z + 3);
        b.ll_info.insert(b.ll_info.end(), x.info, x.info + 6);
        b.ll_robust.push_back(x.robust);
    }
    b.lm_ll_ptr.assign((size_t)Ll + 1, 0);
    for (int f : b.ll_free) b.lm_ll_ptr[f + 1]++;
    for (int l = 0; l < Ll; ++l) b.lm_ll_ptr[l + 1] += b.lm_ll_ptr[l];
    return SVI_OK;
}

// ---- reduced system tiling (identical on every rank: derived from the GLOBAL graph) ------------------------------------
int tile_structure(Build& b)
{
    svi_ba* ba = b.ba;
    const svi_ba_options& o = ba->opt;
    b.TS = o.chol_tile > 0 ? o.chol_tile : 48;
    if (b.TS % 48 != 0 || b.TS > kMaxTile) return fail(SVI_ERR_INVALID, "chol_tile must be 48 or 96");
    const int TS = b.TS, PB = TS / 6;
    b.PB = PB;
    b.n = 6 * b.Pf;
    const int NT = (b.n + TS - 1) / TS;
    b.NT = NT;
    std::vector<uint8_t> nz((size_t)NT * NT, 0);
    for (int t = 0; t < NT; ++t) nz[(size_t)t * NT + t] = 1;
    {
        if (b.use_cpl) { // the pose-level coupling (natural reduced indices) mapped through the elimination order
            const int Pf = b.Pf;
            for (int ri = 0; ri < Pf; ++ri) {
                const int tx = ba->red_perm[ri] / PB;
                const uint8_t* row = &b.cpl[(size_t)ri * Pf];
                for (int rj = b.cpl_lo[ri]; rj <= ri; ++rj)
                    if (row[rj]) { const int ty = ba->red_perm[rj] / PB; nz[(size_t)std::max(tx, ty) * NT + std::min(tx, ty)] = 1; }
            }
        } else {
            // tile chunks touched by every landmark of the global graph (a landmark's poses sit in one or two tiles almost always)
            int v[kLmBlockEdges];
            for (int l = 0; l < b.Ltot; ++l) {
                if (ba->lms[ba->lm_order[l]].fixed) continue;
                int m = 0;
                for (int k = b.g_ptr[l]; k < b.g_ptr[l + 1]; ++k) {
                    const int r = b.pose_red[b.g_pose[k]];
                    if (r < 0) continue;
                    const int t = r / PB;
                    bool seen = false;
                    for (int q = 0; q < m; ++q) if (v[q] == t) { seen = true; break; }
                    if (!seen && m < kLmBlockEdges) v[m++] = t;
                }
                for (int x = 0; x < m; ++x)
                    for (int y = 0; y < m; ++y) if (v[y] <= v[x]) nz[(size_t)v[x] * NT + v[y]] = 1;
            }
        }
        for (const HSe3& e : ba->se3) {
            const int ri = b.pose_red[b.pose_slot[e.i]], rj = b.pose_red[b.pose_slot[e.j]];
            if (ri >= 0 && rj >= 0) { const int x = std::max(ri, rj) / PB, y = std::min(ri, rj) / PB; nz[(size_t)x * NT + y] = 1; }
        }
    }
    const std::vector<uint8_t> nz_orig = nz; // tiles that receive Schur / pose-edge contributions (before fill-in)
    // symbolic fill, right-looking over tile columns
    b.h_col_ptr.assign((size_t)NT + 1, 0);
    std::vector<std::pair<int, int>> col_rows; // (k, i)
    std::vector<int> upd_i, upd_j, upd_k;
    for (int k = 0; k < NT; ++k) {
        std::vector<int> rows;
        for (int i = k + 1; i < NT; ++i) if (nz[(size_t)i * NT + k]) rows.push_back(i);
        b.h_col_ptr[k] = (int)col_rows.size();
        for (int i : rows) col_rows.push_back({k, i});
        for (size_t x = 0; x < rows.size(); ++x)
            for (size_t y = 0; y <= x; ++y) {
                nz[(size_t)rows[x] * NT + rows[y]] = 1;
                upd_i.push_back(rows[x]); upd_j.push_back(rows[y]); upd_k.push_back(k);
            }
    }
    b.h_col_ptr[NT] = (int)col_rows.size();
    // tile ids: the tiles with contributions first, pure fill-in tiles after them - only the former (and g) have to cross the
    // all-reduce, the latter are zero on every rank until the factorisation fills them
    b.tile_map.assign((size_t)NT * NT, -1);
    for (int pass = 0; pass < 2; ++pass)
        for (int j = 0; j < NT; ++j)
            for (int i = j; i < NT; ++i)
                if (nz[(size_t)i * NT + j] && (nz_orig[(size_t)i * NT + j] != 0) == (pass == 0)) {
                    b.tile_map[(size_t)i * NT + j] = (int)b.tile_ti.size(); b.tile_ti.push_back(i); b.tile_tj.push_back(j);
                }
    b.n_tiles = (int)b.tile_ti.size();
    b.n_tiles_orig = 0;
    for (size_t q = 0; q < nz_orig.size(); ++q) b.n_tiles_orig += nz_orig[q] ? 1 : 0;
    b.diag_tile.resize(NT);
    for (auto& kr : col_rows) { b.trsm_tile.push_back(b.tile_map[(size_t)kr.second * NT + kr.first]); b.trsm_row.push_back(kr.second); }
    for (int k = 0; k < NT; ++k) b.diag_tile[k] = b.tile_map[(size_t)k * NT + k];
    // dependency levels: column c waits for every column p < c with a tile (c,p); all columns of one level are factorised by
    // one launch (ba_chol.hip).  The update of a diagonal tile by a column of the level just below is applied by the workgroup
    // that factorises it ("pre" list); every other update is grouped by TARGET tile and runs in the launch that follows its
    // source column's level, one workgroup set per target with the sources in ascending order (no two workgroups ever write
    // the same tile: deterministic without atomics).
    b.level.assign(NT, 0);
    for (int c = 0; c < NT; ++c)
        for (int q = 0; q < c; ++q) if (b.tile_map[(size_t)c * NT + q] >= 0) b.level[c] = std::max(b.level[c], b.level[q] + 1);
    b.n_steps = NT ? *std::max_element(b.level.begin(), b.level.end()) + 1 : 0;
    const int n_steps = b.n_steps;
    b.h_step_ptr.assign((size_t)n_steps + 1, 0);
    for (int st = 0; st < n_steps; ++st) {
        b.h_step_ptr[st] = (int)b.step_col.size();
        for (int c = 0; c < NT; ++c) if (b.level[c] == st) b.step_col.push_back(c);
    }
    b.h_step_ptr[n_steps] = (int)b.step_col.size();
    b.pre_ptr.assign((size_t)NT + 1, 0);
    for (int c = 0; c < NT; ++c) {
        b.pre_ptr[c] = (int)b.pre_tile.size();
        for (int q = 0; q < c; ++q)
            if (b.tile_map[(size_t)c * NT + q] >= 0 && b.level[q] == b.level[c] - 1) { b.pre_tile.push_back(b.tile_map[(size_t)c * NT + q]); b.pre_col.push_back(q); }
    }
    b.pre_ptr[NT] = (int)b.pre_tile.size();
    // target-grouped updates per launch step
    b.h_tgt_ptr.assign((size_t)n_steps + 1, 0);
    b.tgt_pair_ptr.assign(1, 0);
    {
        std::vector<std::vector<size_t>> by_step(n_steps);
        for (size_t u = 0; u < upd_i.size(); ++u) {
            const int i = upd_i[u], j = upd_j[u], q = upd_k[u];
            if (i == j && b.level[q] == b.level[i] - 1) continue; // pre-update, done by the factorising workgroup
            by_step[b.level[q] + 1].push_back(u);
        }
        for (int st = 0; st < n_steps; ++st) {
            b.h_tgt_ptr[st] = (int)b.tgt_tile.size();
            auto& v = by_step[st];
            std::stable_sort(v.begin(), v.end(), [&](size_t x, size_t y) {
                const int tx = b.tile_map[(size_t)upd_i[x] * NT + upd_j[x]], ty = b.tile_map[(size_t)upd_i[y] * NT + upd_j[y]];
                return tx != ty ? tx < ty : upd_k[x] < upd_k[y];
            });
            int cur = -1;
            for (size_t w = 0; w < v.size(); ++w) {
                const size_t u = v[w];
                const int tt = b.tile_map[(size_t)upd_i[u] * NT + upd_j[u]];
                if (tt != cur) {
                    b.tgt_tile.push_back(tt);
                    b.tgt_row.push_back(upd_i[u] == upd_j[u] ? upd_i[u] : -1);
                    b.tgt_pair_ptr.push_back(b.tgt_pair_ptr.back());
                    cur = tt;
                }
                b.pair_a.push_back(b.tile_map[(size_t)upd_i[u] * NT + upd_k[u]]);
                b.pair_b.push_back(b.tile_map[(size_t)upd_j[u] * NT + upd_k[u]]);
                b.pair_src.push_back(upd_k[u]);
                b.tgt_pair_ptr.back()++;
            }
        }
        b.h_tgt_ptr[n_steps] = (int)b.tgt_tile.size();
        const double t3 = (double)TS * TS * TS;
        b.chol_flops = t3 / 3.0 * NT + t3 * (double)col_rows.size() + 2.0 * t3 * (double)upd_i.size();
    }
    b.h_trsm_ptr.assign((size_t)n_steps + 1, 0);
    for (int st = 0; st < n_steps; ++st) {
        b.h_trsm_ptr[st] = (int)b.st_tile.size();
        for (int q = b.h_step_ptr[st]; q < b.h_step_ptr[st + 1]; ++q) {
            const int c = b.step_col[q];
            for (int w = b.h_col_ptr[c]; w < b.h_col_ptr[c + 1]; ++w) { b.st_tile.push_back(b.trsm_tile[w]); b.st_col.push_back(c); }
        }
    }
    b.h_trsm_ptr[n_steps] = (int)b.st_tile.size();
    if (b.dbg) {
        for (int st = 0; st < n_steps; ++st) {
            int mxpre = 0, mxpair = 0;
            for (int q = b.h_step_ptr[st]; q < b.h_step_ptr[st + 1]; ++q) mxpre = std::max(mxpre, b.pre_ptr[b.step_col[q] + 1] - b.pre_ptr[b.step_col[q]]);
            for (int t = b.h_tgt_ptr[st]; t < b.h_tgt_ptr[st + 1]; ++t) mxpair = std::max(mxpair, b.tgt_pair_ptr[t + 1] - b.tgt_pair_ptr[t]);
            fprintf(stderr, "level %d: %d columns, max pre sources %d, %d update targets, max pairs per target %d, %d trsm tiles\n", st,
                    b.h_step_ptr[st + 1] - b.h_step_ptr[st], mxpre, b.h_tgt_ptr[st + 1] - b.h_tgt_ptr[st], mxpair, b.h_trsm_ptr[st + 1] - b.h_trsm_ptr[st]);
        }
    }
    return SVI_OK;
}

// ---- Schur decomposition: always on 48 x 48 sub-tiles (8 poses x 8 poses), whatever TS is ------------------------------
// A stored 48 x 48 sub-tile is four CELLS of 4 x 4 poses (24 x 24); an item is one landmark in one cell: its edges to the
// cell's row poses and to its column poses (masks over the four poses of each).  Cells of four poses instead of eight raise
// the share of (pose, pose) lanes that have work from 36 % to 61 % at KITTI-like co-visibility.  A quarter job is a run of
// <= L items of one cell, a wavefront job four quarter jobs of similar length (one per group of 16 lanes), so that the
// four quarters of a wave finish together.
int schur_work_lists(Build& b)
{
    svi_ba* ba = b.ba;
    constexpr int PBS = 8, SUB = 48, PQ = 4;
    const int TS = b.TS, NT = b.NT, Q = TS / SUB, NSUB = NT * Q, Ll = b.Ll;
    // Stages.  The reduced system is wanted by the factorisation level by level (ba_chol.hip: one launch per dependency level), and
    // the first levels - the leaves of the nested dissection - only need the tiles of THEIR columns.  The Schur work is therefore
    // cut into stages by the level of a tile's column: every wave of k_schur walks its piece of stage 0, then of stage 1, ... and
    // reports each stage as it leaves it, so that the factorisation of the early levels runs (on a second stream, behind a
    // stream wait on a value in memory) while the later stages are still being reduced.  Stage boundaries: levels 1, 2, 4, 8
    // (SVI_SCHUR_STAGES overrides, "0" = one stage); one stage where there is nothing to hide (few levels, several ranks - the
    // all-reduce of the reduced system sits between reduction and factorisation there).
    {
        std::vector<int> bounds; // (off by default until the staged path beats the single launch: DESIGN.md section 9)
        if (const char* e = getenv("SVI_SCHUR_STAGES")) {
            bounds.clear();
            for (const char* p = e; *p;) { char* q = nullptr; const long v = strtol(p, &q, 10); if (q == p) break; if (v > 0) bounds.push_back((int)v); p = (*q == ',') ? q + 1 : q; }
        }
        if (bounds.size() > (size_t)kMaxStages - 1) bounds.resize((size_t)kMaxStages - 1);
        const bool staged = ba->opt.n_ranks == 1 && b.n_steps >= 5 && b.E >= 50000 && !bounds.empty();
        b.level_stage.assign((size_t)std::max(b.n_steps, 1), 0);
        b.n_stages = 1;
        if (staged) {
            std::sort(bounds.begin(), bounds.end());
            int stage = 0;
            size_t nb = 0;
            for (int st = 0; st < b.n_steps; ++st) {
                while (nb < bounds.size() && bounds[nb] <= st) { if (bounds[nb] > 0 && (nb == 0 || bounds[nb] != bounds[nb - 1])) ++stage; ++nb; }
                b.level_stage[st] = stage;
            }
            b.n_stages = stage + 1;
        }
    }
    // stored sub-tiles: every lower sub-tile inside a stored tile (they all have to be (re)written per trial), by stage of the
    // tile's column, then in tile order
    std::vector<int> sub_map((size_t)NSUB * NSUB, -1);
    b.sub_stage_ptr.assign((size_t)b.n_stages + 1, 0);
    for (int stage = 0; stage < b.n_stages; ++stage) {
        b.sub_stage_ptr[stage] = (int)b.sub_cx.size();
        for (int t = 0; t < b.n_tiles; ++t) {
            if (b.level_stage[b.level[b.tile_tj[t]]] != stage) continue;
            for (int sx = 0; sx < Q; ++sx)
                for (int sy = 0; sy < Q; ++sy) {
                    const int cx = b.tile_ti[t] * Q + sx, cy = b.tile_tj[t] * Q + sy;
                    if (cy > cx) continue;
                    sub_map[(size_t)cx * NSUB + cy] = (int)b.sub_cx.size();
                    b.sub_cx.push_back(cx); b.sub_cy.push_back(cy); b.sub_tile.push_back(t);
                }
        }
    }
    b.sub_stage_ptr[b.n_stages] = (int)b.sub_cx.size();
    b.n_sub = (int)b.sub_cx.size();
    const int n_cells = 4 * b.n_sub;
    b.mark(61);
    // the segments of a landmark: runs of its free-pose edges inside one group of four reduced poses
    struct Seg { int chunk, begin, mask, count; };
    auto segments = [&](int l, Seg* seg) -> int {
        int a = b.lm_ptr[l];
        const int end = b.lm_ptr[l + 1];
        while (a < end && b.pose_red[b.e_pose[a]] < 0) ++a; // edges to fixed poses come first
        int ns = 0;
        while (a < end) {
            const int c = b.pose_red[b.e_pose[a]] / PQ;
            Seg sg{c, a, 0, 0};
            while (a < end && b.pose_red[b.e_pose[a]] / PQ == c) { sg.mask |= 1 << (b.pose_red[b.e_pose[a]] % PQ); ++sg.count; ++a; }
            seg[ns++] = sg;
        }
        return ns;
    };
    // pass 1: items per cell and thread (a thread = a contiguous range of landmarks);  pass 2: the items straight into their
    // cell's range, threads in landmark order inside a cell (= a stable sort by cell, landmarks ascending)
    const int NTH = (int)std::min<size_t>(max_threads(), (size_t)std::max(Ll / 8192, 1));
    std::vector<std::vector<int>> hist(NTH, std::vector<int>((size_t)n_cells, 0));
    std::vector<int> terr(NTH, 0);
    std::vector<int64_t> tpairs(NTH, 0);
    auto lrange = [&](int t, int& la, int& lz) { const int per = (Ll + NTH - 1) / NTH; la = std::min(Ll, t * per); lz = std::min(Ll, (t + 1) * per); };
    auto run_threads = [&](auto&& body) {
        std::vector<std::thread> th;
        for (int t = 1; t < NTH; ++t) th.emplace_back([&body, t]() { body(t); });
        body(0);
        for (auto& x : th) x.join();
    };
    run_threads([&](int t) {
        int la, lz;
        lrange(t, la, lz);
        Seg seg[kLmBlockEdges];
        int* h = hist[t].data();
        int err = 0;         // (thread-local: the per-thread slots share cache lines, a million updates each would fight over them)
        int64_t npairs = 0;
        for (int l = la; l < lz; ++l) {
            if (b.lm_fixed[l]) continue;
            const int ns = segments(l, seg);
            for (int x = 0; x < ns; ++x)
                for (int y = 0; y <= x; ++y) {
                    const int qx = seg[x].chunk, qy = seg[y].chunk; // qx >= qy: edges of a landmark ascend in reduced index
                    if (qy > qx) { err = 1; continue; }
                    const int sub = sub_map[(size_t)(qx / 2) * NSUB + qy / 2];
                    if (sub < 0) { err = 2; continue; }
                    h[4 * sub + 2 * (qx % 2) + (qy % 2)]++;
                    npairs += (x == y) ? (int64_t)seg[x].count * (seg[x].count + 1) / 2 : (int64_t)seg[x].count * seg[y].count;
                }
        }
        terr[t] = err; tpairs[t] = npairs;
    });
    b.mark(62);
    int64_t pairs = 0;
    for (int t = 0; t < NTH; ++t) {
        if (terr[t] == 1) return fail(SVI_ERR_STATE, "internal: landmark edges not in reduced pose order");
        if (terr[t] == 2) return fail(SVI_ERR_STATE, "internal: Schur sub-tile outside the tile structure");
        pairs += tpairs[t];
    }
    b.total_pairs = pairs;
    std::vector<int> cell_ptr((size_t)n_cells + 1, 0);
    {
        int run = 0;
        for (int c = 0; c < n_cells; ++c) {
            cell_ptr[c] = run;
            for (int t = 0; t < NTH; ++t) { const int k = hist[t][c]; hist[t][c] = run; run += k; } // now: the thread's cursor in the cell
        }
        cell_ptr[n_cells] = run;
    }
    b.n_items = cell_ptr[n_cells];
    b.mark(63);
    b.it_pack.resize((size_t)4 * std::max(b.n_items, 1));
    b.mark(64);
    run_threads([&](int t) {
        int la, lz;
        lrange(t, la, lz);
        Seg seg[kLmBlockEdges];
        int* cur = hist[t].data();
        for (int l = la; l < lz; ++l) {
            if (b.lm_fixed[l]) continue;
            const int ns = segments(l, seg);
            for (int x = 0; x < ns; ++x)
                for (int y = 0; y <= x; ++y) {
                    const int qx = seg[x].chunk, qy = seg[y].chunk;
                    const int sub = sub_map[(size_t)(qx / 2) * NSUB + qy / 2];
                    int* r = &b.it_pack[(size_t)4 * cur[4 * sub + 2 * (qx % 2) + (qy % 2)]++];
                    r[0] = l; r[1] = seg[x].begin; r[2] = seg[y].begin; r[3] = seg[x].mask | (seg[y].mask << 8);
                }
        }
    });
    b.mark(65);
    // quarter jobs: as many as fit on the chip at once - the kernel holds two waves per SIMD (212 VGPRs, 59 KB of LDS per
    // workgroup), one more would wait for a whole round.  Measured at config 4 with the round-2 kernels (quarter jobs: Schur +
    // assemble us): 4096: 207 + 22, 6144: 185 + 27, 8192: 176 + 33, 10240: 209 + 38.
    // With several stages the workgroups stay for the whole launch and the factorisation's workgroups (512 threads, 100 KB of
    // LDS: they fit beside ONE Schur workgroup on a CU, not beside two) must find room while it runs: `reserve` CUs are left
    // with a single Schur workgroup (SVI_SCHUR_RESERVE_CUS).
    int n_cu = 256;
    if (const int cu = device_compute_units(ba->opt.device)) n_cu = cu;
    constexpr int NX = 8;
    // `reserve` CUs per shader engine (32 of them: n_cu / 8 CUs each) are kept EMPTY by the staged launch (k_schur: workgroups that
    // land there leave at once), so the launch is planned for two workgroups on each of the others (SVI_SCHUR_RESERVE_PER_SE)
    int reserve = 0;
    if (b.n_stages > 1) {
        reserve = 1;
        if (const char* e = getenv("SVI_SCHUR_RESERVE_PER_SE")) reserve = std::min(std::max(atoi(e), 0), 4);
    }
    b.schur_reserve_per_se = reserve;
    const int n_se = std::max(n_cu / 8, 1);
    const int wg_cap = std::max(NX, ((2 * (n_cu - reserve * n_se)) / NX) * NX); // workgroups with work (a multiple of the XCD count)
    b.schur_wgs = wg_cap;
    const int64_t qj_cap = (int64_t)16 * wg_cap;
    // XCD-aware placement.  Workgroups b and b + 8 share an XCD (its 4 MiB L2); an edge's operands are wanted by every cell of
    // its pose group's ROW (as row segment) and COLUMN (as column segment), 7-9 items in as many cells at config 4.  Dealt
    // round-robin, those cells end up on all eight XCDs and each of them fetches the edge over the fabric (700 MB per launch
    // against 77 MB of operands).  So the rows of cells are cut into eight contiguous ranges of equal work, one per XCD:
    // the row-segment re-reads all hit that XCD's L2, the column-segment re-reads mostly (band width < range width).
    const int NG = 2 * NSUB; // row groups of four poses
    auto row_group = [&](int c) { return 2 * b.sub_cx[c / 4] + (c / 2) % 2; };
    struct QJob { int begin, end, cell; };
    std::vector<QJob> qjobs;                 // all stages, ascending in cell
    std::vector<int> stage_q0((size_t)b.n_stages + 1, 0), stage_L((size_t)b.n_stages, 16);
    std::vector<std::vector<int>> stage_grp_x((size_t)b.n_stages, std::vector<int>((size_t)NG, 0));
    int nblk = 0;
    for (int stage = 0; stage < b.n_stages; ++stage) {
        const int c0 = 4 * b.sub_stage_ptr[stage], c1 = 4 * b.sub_stage_ptr[stage + 1];
        std::vector<int> grp_pieces((size_t)NG);
        std::vector<int>& grp_x = stage_grp_x[stage];
        int L = 16;
        for (;; ++L) {
            std::fill(grp_pieces.begin(), grp_pieces.end(), 0);
            int64_t total = 0;
            for (int c = c0; c < c1; ++c) { const int k = (cell_ptr[c + 1] - cell_ptr[c] + L - 1) / L; grp_pieces[row_group(c)] += k; total += k; }
            // contiguous ranges of row groups with about total / NX pieces each
            int64_t acc = 0, worst = 0, in_x = 0;
            int x = 0;
            for (int g = 0; g < NG; ++g) {
                if (x < NX - 1 && in_x > 0 && (acc + grp_pieces[g] / 2) * NX > total * (x + 1)) { worst = std::max(worst, in_x); in_x = 0; ++x; }
                grp_x[g] = x; acc += grp_pieces[g]; in_x += grp_pieces[g];
            }
            worst = std::max(worst, in_x);
            if (L >= 1024 || (total <= qj_cap && worst <= qj_cap / NX)) break;
        }
        stage_L[stage] = L;
        stage_q0[stage] = (int)qjobs.size();
        for (int c = c0; c < c1; ++c)
            for (int i = cell_ptr[c]; i < cell_ptr[c + 1]; i += L) qjobs.push_back({i, std::min(i + L, cell_ptr[c + 1]), c});
    }
    stage_q0[b.n_stages] = (int)qjobs.size();
    // inside an XCD waves take four quarter jobs of similar length; the slabs of a cell are summed in the order of its pieces
    std::vector<std::vector<int>> xq((size_t)b.n_stages * NX);
    auto is_diag = [&](int q) { const int c = qjobs[q].cell, sub = c / 4; return b.sub_cx[sub] == b.sub_cy[sub] && (c / 2) % 2 == c % 2; };
    for (int stage = 0; stage < b.n_stages; ++stage) {
        for (int i = stage_q0[stage]; i < stage_q0[stage + 1]; ++i) xq[(size_t)stage * NX + stage_grp_x[stage][row_group(qjobs[i].cell)]].push_back(i);
        for (int x = 0; x < NX; ++x) {
            std::vector<int>& v = xq[(size_t)stage * NX + x];
            // diagonal cells first: their waves carry the right-hand side (and skip the column segment), the others do neither
            std::stable_sort(v.begin(), v.end(), [&](int u, int w) {
                const bool du = is_diag(u), dw = is_diag(w);
                if (du != dw) return du;
                return qjobs[u].end - qjobs[u].begin > qjobs[w].end - qjobs[w].begin;
            });
            nblk = std::max(nblk, ((int)v.size() + 15) / 16);
        }
    }
    // one grid for all stages: workgroup 8 i + x = the i-th group of sixteen quarter jobs of XCD x in EVERY stage (empty ones pad
    // the short lists); job slot of stage s = s * n_jobs + (job inside the stage)
    b.n_jobs = NX * nblk * 4;
    const size_t nq4s = (size_t)4 * std::max(b.n_jobs, 1);            // quarter-job slots per stage
    const size_t nq4 = nq4s * (size_t)b.n_stages;
    const size_t n_jobs_all = (size_t)std::max(b.n_jobs, 1) * (size_t)b.n_stages;
    b.qj_begin.assign(nq4, 0); b.qj_end.assign(nq4, 0); b.qj_diag.assign(nq4, 0);
    b.job_len.assign(n_jobs_all, 0);
    std::vector<int> slot_of(qjobs.size(), -1), slot_cell(nq4, -1);
    for (int stage = 0; stage < b.n_stages; ++stage)
        for (int x = 0; x < NX; ++x) {
            const std::vector<int>& v = xq[(size_t)stage * NX + x];
            for (size_t j = 0; j < v.size(); ++j) {
                const QJob& q = qjobs[v[j]];
                const size_t k = nq4s * (size_t)stage + ((size_t)(NX * (j / 16) + x) * 4 + (j / 4) % 4) * 4 + j % 4;
                b.qj_begin[k] = q.begin; b.qj_end[k] = q.end;
                const int sub = q.cell / 4, u = (q.cell / 2) % 2, vv = q.cell % 2;
                b.qj_diag[k] = (b.sub_cx[sub] == b.sub_cy[sub] && u == vv) ? 1 : 0;
                b.job_len[k / 4] = std::max(b.job_len[k / 4], q.end - q.begin);
                slot_of[v[j]] = (int)k;
                slot_cell[k] = q.cell;
            }
        }
    // A wave whose four quarter jobs belong to ONE cell (the full-length pieces of a hot cell are neighbours in the sorted lists)
    // adds its quarters up itself and leaves one slab instead of four: k_assemble walks a quarter of the list of a hot cell
    // (forty pieces of a diagonal cell was what its slowest workgroups waited for).  The order of the sums stays fixed.
    b.job_merged.assign(n_jobs_all, 0);
    for (size_t job = 0; job < n_jobs_all; ++job) {
        const int c0 = slot_cell[4 * job];
        b.job_merged[job] = (c0 >= 0 && slot_cell[4 * job + 1] == c0 && slot_cell[4 * job + 2] == c0 && slot_cell[4 * job + 3] == c0) ? 1 : 0;
    }
    if (b.dbg) {
        int nm = 0, nonempty = 0;
        for (size_t job = 0; job < n_jobs_all; ++job) { nm += b.job_merged[job]; nonempty += b.job_len[job] > 0; }
        fprintf(stderr, "schur: %d stages x %d jobs (%d with work), %d of them single-cell (merged slabs), %d workgroups (%d CUs per shader engine kept empty)\n", b.n_stages, b.n_jobs,
                nonempty, nm, b.n_jobs / 4, reserve);
        for (int stage = 0; stage < b.n_stages; ++stage)
            fprintf(stderr, "  stage %d: sub-tiles %d, items %d, quarter jobs %d, piece length %d\n", stage, b.sub_stage_ptr[stage + 1] - b.sub_stage_ptr[stage],
                    cell_ptr[4 * b.sub_stage_ptr[stage + 1]] - cell_ptr[4 * b.sub_stage_ptr[stage]], stage_q0[stage + 1] - stage_q0[stage], stage_L[stage]);
    }
    b.cell_qj_ptr.assign((size_t)n_cells + 1, 0);
    {
        size_t k = 0;
        for (int c = 0; c < n_cells; ++c) {
            b.cell_qj_ptr[c] = (int)b.cell_qj.size();
            while (k < qjobs.size() && qjobs[k].cell == c) { // qjobs ascend in cell
                const int slot = slot_of[k];
                if (!(b.job_merged[slot >> 2] && (slot & 3) != 0)) b.cell_qj.push_back(slot); // (a merged wave: its first quarter carries the sum)
                ++k;
            }
        }
        b.cell_qj_ptr[n_cells] = (int)b.cell_qj.size();
        b.cell_qj.push_back(0); // (k_assemble requests a clamped index for cells without jobs: the list is never empty)
    }
    // in-kernel assembly (k_schur): the wave whose slab is the LAST of a cell to arrive sums the cell's slabs into the tile; it finds
    // the cell of its quarter jobs here.  Cells no slab ever reaches (fill-in tiles, the unused upper cell of a diagonal sub-tile,
    // poses without landmarks) are written by the waves of their stage up front, from this list.
    b.qj_cell.assign(slot_cell.begin(), slot_cell.end());
    for (size_t k = 0; k < b.qj_cell.size(); ++k)
        if (b.qj_cell[k] >= 0 && b.job_merged[k >> 2] && (k & 3) != 0) b.qj_cell[k] = -1; // (a merged wave arrives once, for its first quarter)
    b.orphan_ptr.assign((size_t)b.n_stages + 1, 0);
    for (int stage = 0; stage < b.n_stages; ++stage) {
        b.orphan_ptr[stage] = (int)b.orphan_cell.size();
        for (int c = 4 * b.sub_stage_ptr[stage]; c < 4 * b.sub_stage_ptr[stage + 1]; ++c)
            if (b.cell_qj_ptr[c + 1] == b.cell_qj_ptr[c]) b.orphan_cell.push_back(c);
    }
    b.orphan_ptr[b.n_stages] = (int)b.orphan_cell.size();
    std::vector<std::vector<int>> taux(b.n_sub);
    for (int k = 0; k < (int)b.se3_i.size(); ++k) {
        const int ri = b.pose_red[b.se3_i[k]], rj = b.pose_red[b.se3_j[k]];
        if (ri < 0 || rj < 0 || ri == rj) continue;
        const int tr = ri > rj ? 0 : 1; // row pose = the one with the larger reduced index
        const int hi = std::max(ri, rj), lo = std::min(ri, rj);
        const int sub = sub_map[(size_t)(hi / PBS) * NSUB + lo / PBS];
        if (sub < 0) return fail(SVI_ERR_STATE, "internal: odometry block outside the tile structure");
        taux[sub].push_back((k << 1) | tr);
    }
    b.sub_aux_ptr.assign((size_t)b.n_sub + 1, 0);
    for (int t = 0; t < b.n_sub; ++t) {
        b.sub_aux_ptr[t] = (int)b.sub_aux_ref.size();
        b.sub_aux_ref.insert(b.sub_aux_ref.end(), taux[t].begin(), taux[t].end());
    }
    b.sub_aux_ptr[b.n_sub] = (int)b.sub_aux_ref.size();
    return SVI_OK;
}

int upload(Build& b)
{
    svi_ba* ba = b.ba;
    BaDev& d = ba->d;
    const svi_ba_options& o = ba->opt;
    Uploader up{ba};
    const int Pn = b.Pn, Pf = b.Pf, Ll = b.Ll, E = b.E, TS = b.TS, NT = b.NT, n_tiles = b.n_tiles, n_jobs = b.n_jobs, n_sub = b.n_sub;
    d.Pn = Pn; d.Pf = Pf; d.Ll = Ll; d.E = E;
    d.n_lm_blocks = b.n_lm_blocks; d.n_chunks = b.n_chunks;
    d.n_se3 = (int)b.se3_i.size(); d.n_accel = (int)b.acc_pose.size(); d.n_lmlm = (int)b.ll_free.size();
    d.info_planes = b.planes;
    std::vector<double> hp((size_t)12 * Pn), hl((size_t)3 * Ll);
    for (int s = 0; s < Pn; ++s) memcpy(&hp[(size_t)12 * s], ba->poses[ba->pose_order[s]].T, 96);
    for (int l = 0; l < Ll; ++l) memcpy(&hl[(size_t)3 * l], ba->lms[ba->lm_order[b.L0 + l]].p, 24);
    for (int q = 0; q < 2; ++q) {
        const double* p = nullptr;
        SVI_TRY(up.up(hp, &p)); d.pose[q] = const_cast<double*>(p);
        SVI_TRY(up.up(hl, &p)); d.lm[q] = const_cast<double*>(p);
    }
    SVI_TRY(up.up(b.pose_red, &d.pose_red));
    SVI_TRY(up.up(b.lm_fixed, &d.lm_fixed));
    SVI_TRY(up.up(b.e_pose, &d.e_pose));
    SVI_TRY(up.up(b.e_lm, &d.e_lm));
    SVI_TRY(up.up(b.lm_ptr, &d.lm_ptr));
    SVI_TRY(up.up(b.lb_lm, &d.lb_lm));
    SVI_TRY(up.up(b.lb_rec, &d.lb_rec));
    {
        const int* p = nullptr;
        SVI_TRY(up.up(b.red_slot, &p)); ba->red_slot = const_cast<int*>(p);
        SVI_TRY(up.up(b.loc, &p)); ba->e_orig = const_cast<int*>(p);
    }
    // the edge values, laid out by two gather kernels from the device-side log
    const int np2 = (3 + b.planes + 1) / 2; // double2 planes of the packed [z | information | pad] record
    {
        const int* pm_dev = nullptr;
        SVI_TRY(up.up(b.pm, &pm_dev));
        double *zi = nullptr, *pzi = nullptr;
        uint8_t *fl = nullptr, *pfl = nullptr;
        int* pml = nullptr;
        SVI_TRY(up.alloc((size_t)2 * np2 * std::max(E, 1), &zi, false));
        SVI_TRY(up.alloc((size_t)std::max(E, 1), &fl, false));
        SVI_TRY(up.alloc((size_t)2 * np2 * std::max(E, 1), &pzi, false)); // the pose-major copy uses the same plane stride E (the kernels share load_edge)
        SVI_TRY(up.alloc((size_t)std::max(E, 1), &pfl, false));
        SVI_TRY(up.alloc((size_t)std::max(E, 1), &pml, false));
        ba_gather_edges(ba->raw_log.as<double>(), ba->raw_flags.as<uint8_t>(), ba->e_orig, nullptr, E, E, b.planes, zi, fl, nullptr, nullptr, ba->stream);
        ba_gather_edges(ba->raw_log.as<double>(), ba->raw_flags.as<uint8_t>(), ba->e_orig, pm_dev, b.Epm, E, b.planes, pzi, pfl, d.e_lm, pml, ba->stream);
        SVI_HIP(hipGetLastError());
        d.e_zi = zi; d.e_flags = fl; d.pm_zi = pzi; d.pm_flags = pfl; d.pm_lm = pml;
    }
    SVI_TRY(up.up(b.chunk_pose, &d.chunk_pose));
    SVI_TRY(up.up(b.chunk_begin, &d.chunk_begin));
    SVI_TRY(up.up(b.pose_chunk_ptr, &d.pose_chunk_ptr));
    SVI_TRY(up.up(b.se3_i, &d.se3_i));
    SVI_TRY(up.up(b.se3_j, &d.se3_j));
    SVI_TRY(up.up(b.se3_Z, &d.se3_Z));
    SVI_TRY(up.up(b.se3_info, &d.se3_info));
    SVI_TRY(up.up(b.se3_robust, &d.se3_robust));
    SVI_TRY(up.up(b.acc_pose, &d.acc_pose));
    SVI_TRY(up.up(b.acc_a, &d.acc_a));
    SVI_TRY(up.up(b.acc_info, &d.acc_info));
    SVI_TRY(up.up(b.ll_free, &d.ll_free));
    SVI_TRY(up.up(b.ll_ref, &d.ll_ref));
    SVI_TRY(up.up(b.ll_z, &d.ll_z));
    SVI_TRY(up.up(b.ll_info, &d.ll_info));
    SVI_TRY(up.up(b.ll_robust, &d.ll_robust));
    SVI_TRY(up.up(b.lm_ll_ptr, &d.lm_ll_ptr));
    SVI_TRY(up.up(b.pose_aux_ptr, &d.pose_aux_ptr));
    SVI_TRY(up.up(b.pose_aux_ref, &d.pose_aux_ref));
    SVI_TRY(up.alloc((size_t)12 * E, &d.NZ));
    SVI_TRY(up.alloc((size_t)6 * Ll, &d.Hll));
    SVI_TRY(up.alloc((size_t)3 * Ll, &d.bl));
    SVI_TRY(up.alloc((size_t)6 * Ll, &d.Hinv));
    SVI_TRY(up.alloc((size_t)12 * std::max(Ll, 1), &d.HinvB));
    SVI_TRY(up.alloc((size_t)27 * (b.n_chunks + 2), &d.chunk_out)); // (+2: pose_finalize_block requests two chunks unconditionally)
    SVI_TRY(up.alloc((size_t)120 * d.n_se3, &d.se3_out));
    SVI_TRY(up.alloc((size_t)42 * d.n_accel, &d.acc_out));
    d.lin_count = 27 * Pf + 2 + o.n_ranks;
    SVI_TRY(up.alloc((size_t)d.lin_count, &d.lin_buf));
    d.Hpp = d.lin_buf; d.bp = d.lin_buf + (size_t)21 * Pf; d.lin_scal = d.lin_buf + (size_t)27 * Pf;
    SVI_TRY(up.alloc((size_t)16 * std::max(b.n_lm_blocks, 1), &d.block_part));
    SVI_TRY(up.alloc((size_t)16 * std::max(b.n_lm_blocks, 1), &d.lin_part));
    d.TS = TS; d.NT = NT; d.n_tiles = n_tiles;
    SVI_TRY(up.up(b.tile_map, &d.tile_map));
    // [ g | tiles with contributions | fill-in tiles ]: the all-reduce payload is the prefix g + contributing tiles
    d.red_count = NT * TS + b.n_tiles_orig * TS * TS;
    SVI_TRY(up.alloc(2 + (size_t)NT * TS + (size_t)n_tiles * TS * TS, &d.red_base)); // two doubles in front: see linearize()
    d.g = d.red_base + 2;
    d.S = d.g + (size_t)NT * TS;
    d.upd[0] = d.upd[1] = nullptr;
    if (b.n_stages > 1) for (int q = 0; q < 2; ++q) SVI_TRY(up.alloc((size_t)NT * TS + (size_t)n_tiles * TS * TS, &d.upd[q])); // (zeroed)
    ba->upd_clean[0] = ba->upd_clean[1] = true;
    SVI_TRY(up.alloc((size_t)n_tiles * TS * TS, &d.Lt));
    SVI_TRY(up.alloc((size_t)NT * TS * TS, &d.Linv));
    SVI_TRY(up.alloc((size_t)NT * TS, &d.dx));
    SVI_TRY(up.alloc(1, &d.chol_status));
    d.n_items = b.n_items; d.n_jobs = n_jobs; d.n_sub = n_sub;
    SVI_TRY(up.up(b.it_pack, &d.it_pack));
    SVI_TRY(up.up(b.qj_begin, &d.qj_begin));
    SVI_TRY(up.up(b.qj_end, &d.qj_end));
    SVI_TRY(up.up(b.qj_diag, &d.qj_diag));
    SVI_TRY(up.up(b.job_merged, &d.job_merged));
    SVI_TRY(up.up(b.job_len, &d.job_len));
    d.n_stages = b.n_stages;
    SVI_TRY(up.alloc((size_t)std::max(n_jobs, 1) * b.n_stages * 36 * 64, &d.slab, false));
    SVI_TRY(up.alloc((size_t)std::max(n_jobs, 1) * b.n_stages * 4 * 6 * 4, &d.gslab, false));
    SVI_TRY(up.alloc((size_t)b.n_stages + 1, &d.stage_count));
    SVI_TRY(up.alloc((size_t)4 * std::max(n_sub, 1), &d.cell_count));
    SVI_TRY(up.alloc((size_t)2 * kMaxStages * 8, &d.ticket));
    ba->schur_launch_wgs = 2 * std::max(device_compute_units(ba->opt.device), 8);
    ba->schur_reserve_per_se = b.schur_reserve_per_se;
    SVI_TRY(up.up(b.qj_cell, &d.qj_cell));
    SVI_TRY(up.up(b.orphan_ptr, &d.orphan_ptr));
    SVI_TRY(up.up(b.orphan_cell, &d.orphan_cell));
    ba->sub_stage_ptr = b.sub_stage_ptr;
    ba->level_stage = b.level_stage;
    SVI_TRY(up.up(b.cell_qj_ptr, &d.cell_qj_ptr));
    SVI_TRY(up.up(b.cell_qj, &d.cell_qj));
    SVI_TRY(up.up(b.sub_cx, &d.sub_cx));
    SVI_TRY(up.up(b.sub_cy, &d.sub_cy));
    SVI_TRY(up.up(b.sub_tile, &d.sub_tile));
    SVI_TRY(up.up(b.sub_aux_ptr, &d.sub_aux_ptr));
    SVI_TRY(up.up(b.sub_aux_ref, &d.sub_aux_ref));
    d.add_pose_terms = d.add_aux_blocks = (o.rank == 0) ? 1 : 0;
    // Staged reduction: the tiles are assembled inside k_schur, cell by cell, by whichever wave delivers a cell's last slab - a stage
    // that is reported complete IS assembled, and the factorisation starts on it without a launch in between (SVI_ASM_KERNEL=1: by
    // k_assemble launches on the factorisation's stream, adding to zeroed tiles).  One stage: the k_assemble launch (summing the
    // cells at the end of the one Schur launch, where every wave finishes together, measured 17 us of tail against 15 us of launch).
    d.asm_in_schur = (b.n_stages > 1 && getenv("SVI_ASM_KERNEL") == nullptr) ? 1 : 0;
    if (const char* e = getenv("SVI_ASM_IN_SCHUR")) d.asm_in_schur = atoi(e); // (ablation)
    d.lin_from_red = 0;
    SVI_TRY(up.alloc(16, &d.scal));
    d.aux_blocks = std::max(1, (std::max(d.n_se3, d.n_accel) + 63) / 64);
    SVI_TRY(up.alloc((size_t)2 * d.aux_blocks, &d.aux_part));
    SVI_TRY(up.alloc(1, &d.aux_count));
    SVI_TRY(up.alloc((size_t)4 * 40, &d.tr_part));
    SVI_TRY(up.alloc(1, &d.tr_count));
    if (o.n_ranks > 1) SVI_TRY(up.alloc((size_t)3 * b.Ltot, &ba->lm_all));

    CholPlan& p = ba->plan;
    p.TS = TS; p.NT = NT; p.n_steps = b.n_steps;
    ba->h_step_ptr = b.h_step_ptr; ba->h_tgt_ptr = b.h_tgt_ptr; ba->h_trsm_ptr = b.h_trsm_ptr;
    p.h_step_ptr = ba->h_step_ptr.data(); p.h_tgt_ptr = ba->h_tgt_ptr.data(); p.h_trsm_ptr = ba->h_trsm_ptr.data();
    SVI_TRY(up.up(b.h_col_ptr, &p.col_ptr));
    SVI_TRY(up.up(b.trsm_tile, &p.trsm_tile));
    SVI_TRY(up.up(b.trsm_row, &p.trsm_row));
    SVI_TRY(up.up(b.step_col, &p.step_col));
    {
        // everything a chain / back-substitution workgroup needs to know about its column in ONE record (two 16-byte loads
        // side by side instead of a chain of three dependent index loads at the start of every launch)
        std::vector<int> step_desc((size_t)8 * std::max<size_t>(b.step_col.size(), 1), 0);
        for (size_t q = 0; q < b.step_col.size(); ++q) {
            const int c = b.step_col[q];
            int* r = &step_desc[8 * q];
            r[0] = c; r[1] = b.diag_tile[c]; r[2] = b.pre_ptr[c]; r[3] = b.pre_ptr[c + 1] - b.pre_ptr[c];
            r[4] = b.h_col_ptr[c]; r[5] = b.h_col_ptr[c + 1] - b.h_col_ptr[c];
        }
        SVI_TRY(up.up(step_desc, &p.step_desc));
        // the same per level as kernel arguments, where a level is small enough (ba_chol.hip)
        ba->chain_inl.assign((size_t)b.n_steps, ChainInline{});
        ba->solve_inl.assign((size_t)b.n_steps, SolveInline{});
        for (int st = 0; st < b.n_steps; ++st) {
            const int q0 = b.h_step_ptr[st], nc = b.h_step_ptr[st + 1] - q0;
            bool chain_ok = nc <= kInlineCols, solve_ok = nc <= kInlineCols;
            for (int q = q0; q < q0 + nc; ++q) {
                const int c = b.step_col[q];
                chain_ok = chain_ok && b.pre_ptr[c + 1] - b.pre_ptr[c] <= kInlinePre;
                solve_ok = solve_ok && b.h_col_ptr[c + 1] - b.h_col_ptr[c] <= kInlineSub;
            }
            for (int q = q0; q < q0 + nc; ++q) {
                const int c = b.step_col[q], i = q - q0;
                if (chain_ok) {
                    ChainRec& r = ba->chain_inl[st].c[i];
                    r.k = c; r.tile = b.diag_tile[c]; r.npre = b.pre_ptr[c + 1] - b.pre_ptr[c];
                    for (int w = 0; w < r.npre; ++w) { r.pre_tile[w] = b.pre_tile[b.pre_ptr[c] + w]; r.pre_col[w] = b.pre_col[b.pre_ptr[c] + w]; }
                }
                if (solve_ok) {
                    SolveRec& r = ba->solve_inl[st].c[i];
                    r.k = c; r.nq = b.h_col_ptr[c + 1] - b.h_col_ptr[c];
                    for (int w = 0; w < r.nq; ++w) { r.tile[w] = b.trsm_tile[b.h_col_ptr[c] + w]; r.row[w] = b.trsm_row[b.h_col_ptr[c] + w]; }
                }
            }
            ba->chain_inl[st].n = chain_ok ? nc : 0;
            ba->solve_inl[st].n = solve_ok ? nc : 0;
        }
        p.h_chain_inl = ba->chain_inl.data(); p.h_solve_inl = ba->solve_inl.data();
        {   // the same records once more, for the one-launch backward substitution: levels n_steps-2 .. 0, a column per workgroup
            std::vector<SolveRec> recs;
            bool ok = TS == 48 && b.n_steps >= 2;
            std::vector<int> lvl((size_t)NT, 0); // dependency level of every tile column
            for (int st = 0; st < b.n_steps; ++st)
                for (int q = b.h_step_ptr[st]; q < b.h_step_ptr[st + 1]; ++q) lvl[b.step_col[q]] = st;
            for (int st = b.n_steps - 2; st >= 0 && ok; --st)
                for (int q = b.h_step_ptr[st]; q < b.h_step_ptr[st + 1]; ++q) {
                    const int c = b.step_col[q], nq = b.h_col_ptr[c + 1] - b.h_col_ptr[c];
                    if (nq > kInlineSub) { ok = false; break; }
                    SolveRec r{};
                    r.k = c; r.nq = nq;
                    // rows by level, highest first (= the order in which their x arrives in the backward substitution)
                    int ord[kInlineSub];
                    for (int w = 0; w < nq; ++w) ord[w] = b.h_col_ptr[c] + w;
                    std::sort(ord, ord + nq, [&](int x, int y) {
                        const int lx = lvl[b.trsm_row[x]], ly = lvl[b.trsm_row[y]];
                        return lx != ly ? lx > ly : b.trsm_row[x] > b.trsm_row[y];
                    });
                    for (int w = 0; w < nq; ++w) { r.tile[w] = b.trsm_tile[ord[w]]; r.row[w] = b.trsm_row[ord[w]]; }
                    recs.push_back(r);
                }
            p.solve_recs = nullptr; p.n_solve_cols = 0; p.ybuf = nullptr;
            // every workgroup of that launch (a column each + the pose workgroup) must be resident at once: a workgroup spins on
            // values other workgroups of the same launch publish, and HIP promises no dispatch order (ba_chol.hip)
            int n_cu = device_compute_units(ba->opt.device);
            if (n_cu <= 0) n_cu = 64;
            if (ok && !recs.empty() && (int)recs.size() + 1 <= n_cu) {
                SVI_TRY(up.up(recs, &p.solve_recs));
                p.n_solve_cols = (int)recs.size();
                double* yb = nullptr;
                SVI_TRY(up.alloc((size_t)NT * TS, &yb));
                p.ybuf = yb;
            }
        }
        SVI_TRY(up.up(b.diag_tile, &p.diag_tile));
        SVI_TRY(up.up(b.pre_ptr, &p.pre_ptr));
        SVI_TRY(up.up(b.pre_tile, &p.pre_tile));
        SVI_TRY(up.up(b.pre_col, &p.pre_col));
        SVI_TRY(up.up(b.tgt_tile, &p.tgt_tile));
        SVI_TRY(up.up(b.tgt_row, &p.tgt_row));
        SVI_TRY(up.up(b.tgt_pair_ptr, &p.tgt_pair_ptr));
        SVI_TRY(up.up(b.pair_a, &p.pair_a));
        SVI_TRY(up.up(b.pair_b, &p.pair_b));
        SVI_TRY(up.up(b.pair_src, &p.pair_src));
        SVI_TRY(up.up(b.st_tile, &p.st_tile));
        SVI_TRY(up.up(b.st_col, &p.st_col));
        if (!ba->h_scal) SVI_HIP(hipHostMalloc(reinterpret_cast<void**>(&ba->h_scal), 16 * sizeof(double)));
        if (!ba->h_status) SVI_HIP(hipHostMalloc(reinterpret_cast<void**>(&ba->h_status), sizeof(int) * 4));
        ba->h_status[0] = ba->h_status[1] = 0;
        ba->pub_seq = 0;
        {   // the finished structure once more, in device memory (BaDev::self)
            BaDev* self = nullptr;
            SVI_TRY(up.alloc((size_t)1, &self, false));
            d.self = self;
            ba->d_copy = d;   // (the source of the copy must outlive the asynchronous transfer)
            SVI_HIP(hipMemcpyAsync(self, &ba->d_copy, sizeof(BaDev), hipMemcpyHostToDevice, ba->stream));
        }
        ba_configure_kernels(TS);
        // the host vectors of this call are read by the copies above: drain before they go out of scope
        SVI_HIP(hipStreamSynchronize(ba->stream));
    }
    return SVI_OK;
}

} // namespace

namespace svi {

// the graph is the one the device structures were built for: only the estimates go back to the device
int reupload_state(svi_ba* ba)
{
    BaDev& d = ba->d;
    std::vector<double> hp((size_t)12 * d.Pn), hl((size_t)3 * std::max(d.Ll, 1));
    for (int s = 0; s < d.Pn; ++s) memcpy(&hp[(size_t)12 * s], ba->poses[ba->pose_order[s]].T, 96);
    for (int l = 0; l < d.Ll; ++l) memcpy(&hl[(size_t)3 * l], ba->lms[ba->lm_order[ba->L0 + l]].p, 24);
    for (int q = 0; q < 2; ++q) {
        if (d.Pn) SVI_HIP(hipMemcpyAsync(d.pose[q], hp.data(), sizeof(double) * 12 * d.Pn, hipMemcpyHostToDevice, ba->stream));
        if (d.Ll) SVI_HIP(hipMemcpyAsync(d.lm[q], hl.data(), sizeof(double) * 3 * d.Ll, hipMemcpyHostToDevice, ba->stream));
    }
    SVI_HIP(hipMemsetAsync(d.chol_status, 0, sizeof(int), ba->stream));
    SVI_HIP(hipMemsetAsync(d.aux_count, 0, sizeof(int), ba->stream));
    SVI_HIP(hipMemsetAsync(d.tr_count, 0, sizeof(int), ba->stream));
    ba->hinv_valid = false; ba->lin_post_deferred = false; ba->spec_lin_state = -1;
    ba->h_status[0] = ba->h_status[1] = 0;
    ba->pub_seq = 0;
    ba->lin_local = false;
    SVI_HIP(hipStreamSynchronize(ba->stream));
    return SVI_OK;
}

int build_structure(svi_ba* ba)
{
    if (!ba->build_ctx) ba->build_ctx = std::shared_ptr<void>(new Build(), [](void* p) { delete static_cast<Build*>(p); });
    Build& b = *static_cast<Build*>(ba->build_ctx.get());
    b.clear_all();
    b.ba = ba;
    b.dbg = getenv("SVI_DEBUG_PLAN") != nullptr;
    b.t0 = std::chrono::steady_clock::now();
    BaDev& d = ba->d;
    const svi_ba_options& o = ba->opt;
    d.fx = o.fx; d.fy = o.fy; d.cx = o.cx; d.cy = o.cy; d.cauchy_delta = o.cauchy_delta;
    b.mark(0);
    SVI_TRY(edges_flush(ba));      // whatever the add_* calls have not sent yet travels while the host sorts
    order_vertices(b);
    b.mark(1);
    sort_edges(b);
    b.mark(2);
    pose_coupling(b);
    b.mark(21);
    elimination_order(b);
    b.mark(3);
    SVI_TRY(local_edges(b));
    b.mark(4);
    SVI_TRY(aux_edges(b));
    b.mark(5);
    SVI_TRY(tile_structure(b));
    b.mark(6);
    SVI_TRY(schur_work_lists(b));
    b.mark(7);
    SVI_TRY(upload(b));
    b.mark(8);

    svi_ba_stats& st = ba->stats;
    const uint64_t it0 = st.lm_iterations, tr0 = st.lm_trials, cf0 = st.chol_failures, bt0 = st.backsolve_timeouts;
    st = svi_ba_stats{};
    st.lm_iterations = it0; st.lm_trials = tr0; st.chol_failures = cf0; st.backsolve_timeouts = bt0;
    st.n_poses = b.Pn; st.n_poses_free = b.Pf; st.n_landmarks = b.Ltot; st.n_landmarks_local = b.Ll;
    st.n_edges_proj = b.Etot; st.n_edges_proj_local = b.E;
    st.n_edges_se3 = (int64_t)ba->se3.size(); st.n_edges_accel = (int64_t)ba->acc.size(); st.n_edges_lmlm = (int64_t)ba->lmlm.size();
    st.n_schur_tiles = b.n_jobs; st.n_window_blocks = b.total_pairs;
    st.chol_n = b.n; st.chol_tile = b.TS; st.chol_tiles_nnz = b.n_tiles; st.chol_steps = b.n_steps;
    st.reduce_doubles = d.red_count;
    st.chol_flops = b.chol_flops;
    return SVI_OK;
}

} // namespace svi
