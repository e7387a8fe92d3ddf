// main.cpp — drives the host side of the BA C ABI (include/svi_hot.h) under AddressSanitizer + UBSan without a GPU:
// random trajectory graphs through the reference-shaped construction calls, svi_ba_initialize (the whole structure
// analysis: vertex order, nested dissection, edge sorting, tile structure, Schur work lists, every upload size) for 1, 2
// and 3 ranks and both tile sizes / elimination orders, the .g2o writer and reader, pruning and the write-back.
// The LM loop itself needs the kernels and is not run here (tests/test_ba_gpu.py does that on the GPU).
// usage: host_san [n_graphs] [seed] ; prints one line per graph and "host_san: ok"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "svi_hot.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != SVI_OK) { std::fprintf(stderr, "%s:%d: %s -> %s: %s\n", __FILE__, __LINE__, #x, svi_status_string(rc_), svi_last_error()); return 1; } } while (0)

static void pose_at(double yaw, double x, double y, double z, double T[12])
{
    const double c = std::cos(yaw), s = std::sin(yaw);
    const double R[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
    for (int i = 0; i < 9; ++i) T[i] = R[i];
    T[9] = x; T[10] = y; T[11] = z;
}

static int build(svi_ba* ba, int n_kf, int n_lm, int track, std::mt19937_64& rng, bool imu)
{
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    const double fx = 450.5, cx = 376.0, cy = 222.0, fb = 49.63;
    std::vector<double> T((size_t)12 * n_kf);
    for (int k = 0; k < n_kf; ++k) pose_at(0.03 * k, 0.1 * k, 0.0, 0.7 * k, &T[(size_t)12 * k]);
    const double off[12] = {-1, 0, 0, 0, -1, 0, 0, 0, 1, 0.06, 0.005, 0.01};
    if (imu) CHECK(svi_ba_set_imu_offset(ba, off));
    CHECK(svi_ba_add_pose(ba, 1000000, &T[0], 1));
    const double a0[3] = {0, -1, 0}, I3[6] = {1, 0, 0, 1, 0, 1};
    CHECK(svi_ba_add_edge_accel(ba, 1000000, a0, imu ? off : nullptr, I3));
    std::vector<int> first(n_lm);
    for (int l = 0; l < n_lm; ++l) {
        first[l] = (int)((uint64_t)rng() % (uint64_t)std::max(1, n_kf - 1));
        const double* P = &T[(size_t)12 * first[l]];
        const double pc[3] = {2.0 * U(rng), 0.8 * U(rng), 3.0 + 6.0 * (U(rng) + 1.0)};
        double pw[3];
        for (int r = 0; r < 3; ++r) pw[r] = P[3 * r] * pc[0] + P[3 * r + 1] * pc[1] + P[3 * r + 2] * pc[2] + P[9 + r];
        CHECK(svi_ba_add_landmark(ba, l, pw, 0));
    }
    for (int k = 0; k < n_kf; ++k) {
        if (k > 0) {
            const double a[3] = {0.1 * U(rng), 0.1 * U(rng), -0.99};
            CHECK(svi_ba_add_keyframe(ba, 1000000 + k, 1000000 + k - 1, &T[(size_t)12 * k], nullptr, imu ? a : nullptr));
        }
        std::vector<int64_t> ids;
        std::vector<float> uvL, uvR;
        std::vector<double> xyz;
        for (int l = 0; l < n_lm; ++l) {
            if (k < first[l] || k >= first[l] + track) continue;
            double pw[3];
            CHECK(svi_ba_get_landmark(ba, l, pw));
            const double* P = &T[(size_t)12 * k];
            double pc[3];
            for (int c = 0; c < 3; ++c) pc[c] = P[c] * (pw[0] - P[9]) + P[3 + c] * (pw[1] - P[10]) + P[6 + c] * (pw[2] - P[11]);
            if (pc[2] < 0.5) continue;
            const float u = (float)(fx * pc[0] / pc[2] + cx), v = (float)(fx * pc[1] / pc[2] + cy), d = (float)std::max(1.0, std::rint(fb / pc[2]));
            ids.push_back(l);
            uvL.push_back(u); uvL.push_back(v); uvR.push_back(u - d); uvR.push_back(v);
            const double z = fb / d;
            xyz.push_back(z * (u - cx) / fx); xyz.push_back(z * (v - cy) / fx); xyz.push_back(z);
        }
        int64_t stored[3];
        CHECK(svi_ba_add_measurements(ba, 1000000 + k, (int64_t)ids.size(), ids.data(), uvL.data(), uvR.data(), xyz.data(), stored));
    }
    return 0;
}

int main(int argc, char** argv)
{
    const int n_graphs = argc > 1 ? std::atoi(argv[1]) : 12;
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
    const std::string tmp = argc > 3 ? argv[3] : "/tmp/host_san.g2o";
    std::mt19937_64 rng(seed);
    for (int gi = 0; gi < n_graphs; ++gi) {
        const int n_kf = 3 + (int)(rng() % 140), n_lm = 20 + (int)(rng() % 1500), track = 2 + (int)(rng() % 12);
        const int ranks = 1 + gi % 3, tile = (gi / 3) % 2 ? 48 : 96, order = (gi / 6) % 2;
        int64_t n_edges = 0;
        for (int rank = 0; rank < ranks; ++rank) {
            svi_ba_options o;
            svi_ba_options_default(&o);
            o.fx = o.fy = 450.5; o.cx = 376.0; o.cy = 222.0; o.baseline_m = 0.1102;
            o.rank = rank; o.n_ranks = ranks; o.chol_tile = tile; o.chol_order = order;
            svi_ba* ba = nullptr;
            CHECK(svi_ba_create(&o, &ba));
            std::mt19937_64 g2(seed * 7919 + gi); // the same graph on every rank
            if (build(ba, n_kf, n_lm, track, g2, gi % 2 == 1)) return 1;
            CHECK(svi_ba_initialize(ba));
            svi_ba_stats st;
            CHECK(svi_ba_get_stats(ba, &st));
            CHECK(svi_ba_num_edges(ba, &n_edges));
            if (rank == 0) {
                // writer -> reader -> structure analysis of the loaded graph
                CHECK(svi_ba_save_g2o(ba, tmp.c_str()));
                svi_ba* rb = nullptr;
                CHECK(svi_ba_create(&o, &rb));
                CHECK(svi_ba_load_g2o(rb, tmp.c_str()));
                int64_t ne2 = 0;
                CHECK(svi_ba_num_edges(rb, &ne2));
                if (ne2 != n_edges) { std::fprintf(stderr, "graph %d: %lld edges written, %lld read\n", gi, (long long)n_edges, (long long)ne2); return 1; }
                CHECK(svi_ba_initialize(rb));
                // write-back with one diverged landmark, then the pruned graph is analysed again
                const double far[3] = {2e6, 0, 0};
                CHECK(svi_ba_add_landmark(rb, 900000, far, 0));
                int64_t nl = 0, np = 0, erased = 0;
                CHECK(svi_ba_num_landmarks(rb, &nl));
                CHECK(svi_ba_num_poses(rb, &np));
                std::vector<int64_t> lid((size_t)nl), kid((size_t)np);
                std::vector<double> lx((size_t)3 * nl), kT((size_t)12 * np);
                std::vector<uint8_t> kept((size_t)nl);
                const double shift[3] = {1, 2, 3};
                CHECK(svi_ba_apply_optimization(rb, shift, lid.data(), lx.data(), kept.data(), kid.data(), kT.data(), &erased));
                if (erased != 1) { std::fprintf(stderr, "graph %d: %lld landmarks erased, expected 1\n", gi, (long long)erased); return 1; }
                CHECK(svi_ba_initialize(rb));
                CHECK(svi_ba_destroy(rb));
            }
            std::printf("graph %d rank %d/%d tile %d order %d: %lld kf %lld lm %lld edges, %lld tiles, %lld levels\n", gi, rank, ranks, tile, order,
                        (long long)st.n_poses, (long long)st.n_landmarks, (long long)n_edges, (long long)st.chol_tiles_nnz, (long long)st.chol_steps);
            CHECK(svi_ba_destroy(ba));
        }
    }
    std::remove(tmp.c_str());
    std::printf("host_san: ok\n");
    return 0;
}
