// ba_device.h — HBM layout of the bundle-adjustment path and the launchers of its kernels.
//
// Everything the LM loop touches is resident on the device; the host only sees a handful of
// scalars per trial (chi2, step scale, Cholesky status).
//
// Index spaces
//   pose slot    s in [0, Pn)   poses in ascending id (g2o index-mapping order)
//   reduced pose r in [0, Pf)   free poses only, same order; pose_red[s] = r or -1 (fixed)
//   landmark     l in [0, Ll)   this rank's landmarks in ascending id (contiguous id range)
//   lm-major edge e in [0, E)   this rank's projection edges sorted by (l, s)
//   pose-major edge             the same edges sorted by (s, l)
//   reduced system               n = 6*Pf scalars, tiled TS x TS (TS multiple of 48, default 96),
//                                 only the lower block triangle is stored, only tiles that can be
//                                 non-zero after fill-in ("tile-sparse")
#pragma once
#include <cstdint>

namespace svi {

enum : int { kTypeXYZ = 0, kTypeDepth = 1, kTypeDisparity = 2 };
// e_flags bits
enum : unsigned { kFlagTypeMask = 3u, kFlagRobust = 4u };

constexpr int kLmBlockEdges = 256;   // lm-major kernels: one workgroup owns whole landmarks, <= 256 edges
#ifndef SVI_POSE_CHUNK
#define SVI_POSE_CHUNK 1024
#endif
constexpr int kPoseChunk    = SVI_POSE_CHUNK;  // pose-major kernel: one workgroup sums <= 1024 edges of one pose
constexpr int kMaxTile      = 96;
constexpr int kMaxStages    = 8;     // stages of the Schur reduction

struct BaDev {
    // camera / robust kernel
    double fx, fy, cx, cy, cauchy_delta;

    // sizes
    int Pn, Pf, Ll, E;
    int n_lm_blocks;     // lm-major workgroups
    int n_chunks;        // pose-major workgroups
    int n_se3, n_accel, n_lmlm;
    int info_planes;     // 3: all informations diagonal (only 00,11,22 stored), 6: upper triangles

    // state (double buffered: [cur] is the accepted estimate, [cur^1] the trial)
    double* pose[2];     // [Pn][12]  R row-major (9), t (3)
    double* lm[2];       // [Ll][3]
    const int* pose_red; // [Pn]
    const uint8_t* lm_fixed; // [Ll]

    // lm-major projection edges (SoA)
    const int*     e_pose;  // [E] pose slot
    const int*     e_lm;    // [E] landmark
    const uint8_t* e_flags; // [E]
    const double*  e_zi;    // [(3 + info_planes + 1) / 2][E] double2: z (3), information (info_planes), pad
    const int*     lm_ptr;  // [Ll+1] edges of landmark l
    const int*     lb_lm;   // [n_lm_blocks+1] landmarks of lm-major workgroup b
    const int*     lb_rec;  // [n_lm_blocks][4] {first landmark, #landmarks, first edge, end edge} of workgroup b (one load)
    // pose-major copy
    const int*     pm_lm;    // [E]
    const uint8_t* pm_flags; // [E]
    const double*  pm_zi;    // the same, pose-major order
    const int*     chunk_pose;  // [n_chunks] pose slot
    const int*     chunk_begin; // [n_chunks+1]
    const int*     pose_chunk_ptr; // [Pn+1] chunks of pose slot s

    // pose-only / landmark-only edges (rank 0 holds se3 + accel; lmlm lives with its free landmark)
    const int*    se3_i;    // [n_se3] pose slots
    const int*    se3_j;
    const double* se3_Z;    // [n_se3][12]
    const double* se3_info; // [n_se3][21]
    const uint8_t* se3_robust;
    const int*    acc_pose; // [n_accel]
    const double* acc_a;    // [n_accel][3]  (already rotated by the offset: R_off a)
    const double* acc_info; // [n_accel][6]
    const int*    ll_free;  // [n_lmlm] local landmark that is optimised
    const double* ll_ref;   // [n_lmlm][3]  position of the fixed partner
    const double* ll_z;     // [n_lmlm][3]  measurement, sign-adjusted so that e = p_free - ref - z
    const double* ll_info;  // [n_lmlm][6]
    const uint8_t* ll_robust;
    const int*    lm_ll_ptr; // [Ll+1] lmlm edges per landmark (sorted by ll_free)
    // per pose: references into the aux-edge outputs that add to its diagonal block / rhs
    const int*    pose_aux_ptr; // [Pn+1]
    const int*    pose_aux_ref; // se3 edge k as (k<<2)|0 (role i) or |1 (role j); accel edge k as (k<<2)|2

    // linearisation outputs
    double* NZ;       // [6][E] double2: per edge N = A'(rho1 Omega)A R' (3x3 row-major, values 0-8) and 2Z, Z = R'(p-t) (values 9-11);
                      //          H_pl = [ -N ; -2[Z]x N ] is never materialised
    double* Hll;      // [6][Ll]  upper triangle of H_ll
    double* bl;       // [3][Ll]
    double* Hinv;     // [6][Ll]  (H_ll + lambda I)^-1, per trial
    double* HinvB;    // [Ll][12] the same inverse (6) and b_l (3) as one record per landmark: what the Schur kernel stages
    double* chunk_out;// [n_chunks][27]  21 (upper of 6x6) + 6
    double* se3_out;  // [n_se3][120]  Hii(36) Hjj(36) Hij(36) bi(6) bj(6)
    double* acc_out;  // [n_accel][42] H(36) b(6)
    // lin_buf (one all-reduce per linearisation): Hpp[Pf][21] | bp[Pf][6] | chi_robust chi_plain | maxdiag[n_ranks]
    double* lin_buf;
    double* Hpp;      // = lin_buf
    double* bp;       // = lin_buf + 21*Pf
    double* lin_scal; // = lin_buf + 27*Pf : [0] robust chi2, [1] plain chi2, [2..2+n_ranks) max diag of H_ll per rank
    int     lin_count;
    double* block_part; // [n_lm_blocks][4 waves][4] per-wave partial sums of the landmark-major kernels
    double* lin_part;   // the same for the linearisation (k_linearize_lm): it outlives the trial that follows, whose sums are in block_part

    // reduced system
    int TS, NT;            // tile edge, tiles per side
    int n_tiles;           // stored tiles
    const int* tile_map;   // [NT][NT] lower triangle -> tile id or -1
    double* g;             // red_buf: g[NT*TS] | tiles with contributions | fill-in tiles; the first red_count doubles
    double* S;             // = g + NT*TS : [n_tiles][TS*TS]                       (g + contributing tiles) are all-reduced per trial
    int     red_count;
    double* upd[2];        // staged reduction only: [ g_upd[NT*TS] | S_upd[n_tiles][TS*TS] ], twice - what the factorisation has subtracted from
                           // g / S so far in a trial (it reads g + g_upd, S + S_upd, ba_chol.hip); the two take turns from trial to trial
    double* Lt;            // [n_tiles][TS*TS] Cholesky factor tiles
    double* Linv;          // [NT][TS*TS] inverses of the diagonal Cholesky factors
    double* dx;            // [NT*TS] solution (pose increments)
    int*    chol_status;   // [1] 0 ok, k+1: pivot failure in tile column k

    // Schur reduction: cells of 4 x 4 poses (four per stored 48 x 48 sub-tile: cell = 4 sub + 2 u + v), items
    // (landmark x cell), quarter jobs (runs of items of one cell, 16 lanes), wavefront jobs (4 quarter jobs), slabs
    int n_items, n_jobs, n_sub; // n_jobs: wavefront jobs PER STAGE = 4 x the workgroups of k_schur; job slot = stage * n_jobs + job
    int n_stages;               // stages of the Schur reduction (by dependency level of a tile's column, ba_structure.cpp); 1 = unstaged
    int* stage_count;           // [n_stages] arrival counters of k_schur's waves (zero between launches)
    int* ticket;                // [2][kMaxStages][8] staged launches: next group of quarter jobs per stage and XCD (two sets, by launch parity)
    int* cell_count;            // [4 n_sub] slabs of a cell that have arrived (zero between launches): the last arrival assembles the cell
    const int* qj_cell;         // [n_stages][4 n_jobs] cell a quarter job's slab belongs to, -1 if it leaves none
    const int* orphan_ptr;      // [n_stages + 1] cells without any slab, by stage
    const int* orphan_cell;
    int asm_in_schur;           // 1: k_schur assembles the tiles itself (no k_assemble launch)
    const BaDev* self;          // this structure in device memory (what a non-inlined device function reads the layout from)
    const int* it_pack;      // [n_items][4]: landmark, first edge of the row segment, first edge of the column segment, maskI | maskJ << 8
    const int* qj_begin;     // [n_stages][4 n_jobs] first item of quarter job 4 job + quarter
    const int* qj_end;       // [n_stages][4 n_jobs]
    const int* qj_diag;      // [n_stages][4 n_jobs] 1: diagonal cell (lower blocks only, carries g)
    const int* job_merged;   // [n_stages][n_jobs] 1: the four quarter jobs are pieces of one cell - the wave adds them up, quarter 0 carries the sum
    const int* job_len;      // [n_stages][n_jobs] longest of the four quarter jobs
    double* slab;            // [n_stages][n_jobs][36][64]  element q of the 6x6 block of lane (quarter = lane>>4, i = (lane>>2)&3, j = lane&3)
    double* gslab;           // [n_stages][4 n_jobs][6][4]
    const int* cell_qj_ptr;  // [4 n_sub + 1] quarter jobs of a cell, in summation order
    const int* cell_qj;
    const int* sub_cx;       // [n_sub] sub-tile row / column in units of 48
    const int* sub_cy;
    const int* sub_tile;     // [n_sub] tile id that holds it
    const int* sub_aux_ptr;  // [n_sub+1]
    const int* sub_aux_ref;  // (se3 edge << 1) | transposed
    int add_pose_terms;       // this rank adds Hpp / bp: rank 0 when they hold the all-reduced totals, every rank when they
                              // hold its own partial sums (summed with the reduced system then)
    int add_aux_blocks;       // rank 0 adds the odometry off-diagonal blocks
    int lin_from_red;         // the chi2 of the linearisation travels in front of g (red_base[0..1]) instead of in lin_buf
    double* red_base;         // = g - 2

    // LM scalars on the device
    double* aux_part;  // [aux_blocks][2] chi2 partials of the pose-only edges
    double* tr_part;   // [2 * 16 + 1][4] records of the workgroups of the trial's closing reduction
    int* tr_count;     //   their arrival counter (zero between launches)
    int*    aux_count; // arrival counter of k_aux_edges (zero between launches)
    int     aux_blocks;
    double* scal;   // [8]: 0 chi_robust(trial) 1 chi_plain(trial) 2 scale_lm 3 scale_pose 4 spare...
};

// tile-sparse Cholesky of the reduced system (ba_chol.hip)
// What a chain workgroup has to know about its column, handed over in the KERNEL ARGUMENTS when the level is small: every
// index list in global memory is a dependent round trip at the start of a launch (0.6 - 0.8 us each for data that sits behind
// the L2 of another XCD), and a level's launch is on the critical path of the factorisation.
constexpr int kInlineCols = 8, kInlinePre = 2, kInlineSub = 12;
struct ChainRec { int k, tile, npre, pad; int pre_tile[kInlinePre], pre_col[kInlinePre]; };
struct ChainInline { int n; int pad[3]; ChainRec c[kInlineCols]; };                      // n = 0: read the lists instead
struct SolveRec { int k, nq; int tile[kInlineSub], row[kInlineSub]; };
struct SolveInline { int n; int pad[3]; SolveRec c[kInlineCols]; };

// The pose update of a trial (g2o VertexSE3::oplusImpl + the pose part of computeScale), run by one extra workgroup of the
// one-launch backward substitution: it waits for its dx entries like the column workgroups wait for theirs (src == nullptr: off)
struct PoseTail {
    const double* src; double* dst;   // pose [cur], pose [cur ^ 1]
    const int* pose_red;              // [Pn] reduced index or -1 (fixed)
    const double* bp;                 // [6 Pf]
    double* scal;                     // scal[3] <- sum dx (wl dx + wb b_p); scal[8], [9] <- red_base[0], [1] if lin_from_red
    const double* red_base;
    int Pn, lin_from_red;
    double wl, wb;
};

// what a wave of k_schur does when it leaves a stage: the last wave to arrive publishes `seq` at sig[stage] (memory a stream
// waits on with hipStreamWaitValue64; nullptr: nobody waits)
struct StageSignals { unsigned long long* sig[kMaxStages]; unsigned long long seq; };

struct CholPlan {
    int TS = 0, NT = 0, n_steps = 0;
    // host: ranges of one dependency level (= one launch) in the device lists
    const int* h_step_ptr = nullptr;  // [n_steps+1] into step_col
    const int* h_tgt_ptr = nullptr;   // [n_steps+1] into tgt_*: targets updated by the launch of step s
    const int* h_trsm_ptr = nullptr;  // [n_steps+1] into st_tile / st_col
    // device
    const int* step_col = nullptr;    // tile columns of each level, ascending
    const int* step_desc = nullptr;   // [len(step_col)][8]: column, its diagonal tile, pre_ptr, #pre, col_ptr, #sub-diagonal tiles, 0, 0
    const int* diag_tile = nullptr;   // [NT] tile id of (k,k)
    const int* pre_ptr = nullptr;     // [NT+1]: updates of (k,k) applied by the workgroup that factorises it
    const int* pre_tile = nullptr;    //   tile (k,q)
    const int* pre_col = nullptr;     //   q  (y_q feeds the forward substitution)
    const int* tgt_tile = nullptr;    // target tile of a grouped update
    const int* tgt_row = nullptr;     //   its tile row if it is a diagonal tile (g rides along), else -1
    const int* tgt_pair_ptr = nullptr;//   [n_targets+1] into pair_*
    const int* pair_a = nullptr;      // tile (i,q)
    const int* pair_b = nullptr;      // tile (j,q)
    const int* pair_src = nullptr;    // q
    const int* st_tile = nullptr;     // trsm items of a level: tile (i,k) ...
    const int* st_col = nullptr;      // ... and its column k
    const int* col_ptr = nullptr;     // [NT+1] sub-diagonal tiles of every column (back substitution)
    const int* trsm_tile = nullptr;
    const int* trsm_row = nullptr;
    // host: the same per level as kernel arguments (n = 0 where a level does not fit)
    const ChainInline* h_chain_inl = nullptr; // [n_steps]
    const SolveInline* h_solve_inl = nullptr; // [n_steps]
    // one-launch backward substitution: a record per column below the last level, highest level first; the forward vector
    const SolveRec* solve_recs = nullptr;    // device
    int n_solve_cols = 0;                    // 0: not available (a column with more than kInlineSub tiles, or another tile size)
    double* ybuf = nullptr;                  // device, [NT * TS]
};


} // namespace svi
