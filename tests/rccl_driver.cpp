// rccl_driver.cpp — a torchless, Python-less multi-process driver of the landmark-sharded BA: one process per GPU, every rank
// builds the same synthetic graph through the C ABI, owns its landmark shard (svi_ba_options.rank / n_ranks) and sums the
// reduced camera system through the library's native RCCL hook (svi_rccl_*).  Test infrastructure (tests/test_rccl_native.py);
// it is also the shape of a C++ host that runs the reference's optimizer on a multi-GPU node.
//   rccl_driver <n_ranks> <n_keyframes> <n_landmarks> <id_file> [shards]   n_ranks == 0: one process, no communicator;
//   shards (one-GPU boxes): the graph is cut into that many shards although the communicator has n_ranks ranks - rank r then
//   solves the sub-problem of its own landmarks, every collective of the sharded path runs through RCCL
// Every rank prints "rank r/n: nominal executed chi2_plain chi2_robust checksum" (checksum = sum of all pose entries).
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "svi_hot.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != SVI_OK) { std::fprintf(stderr, "rank %d: %s -> %s: %s\n", g_rank, #x, svi_status_string(rc_), svi_last_error()); return 1; } } while (0)
static int g_rank = 0;

static int build(svi_ba* ba, int n_kf, int n_lm)
{
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    const double fx = 718.856, cx = 607.1928, cy = 185.2157, fb = 386.1448;
    std::vector<double> T((size_t)12 * n_kf), Tn((size_t)12 * n_kf);
    for (int k = 0; k < n_kf; ++k) {
        const double yaw = 0.02 * k, c = std::cos(yaw), s = std::sin(yaw);
        const double R[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
        memcpy(&T[(size_t)12 * k], R, 72);
        T[12 * k + 9] = 0.05 * k; T[12 * k + 10] = 0.0; T[12 * k + 11] = 1.0 * k;
        memcpy(&Tn[(size_t)12 * k], &T[(size_t)12 * k], 96);
        if (k > 0) for (int q = 9; q < 12; ++q) Tn[12 * k + q] += 0.03 * N(rng);    // perturbed initial estimates
    }
    CHECK(svi_ba_add_pose(ba, 1000000, &Tn[0], 1));
    const double a0[3] = {0, 0, 0}, I3[6] = {1, 0, 0, 1, 0, 1};
    CHECK(svi_ba_add_edge_accel(ba, 1000000, a0, nullptr, I3));
    const int track = 8;
    std::vector<int> first(n_lm);
    std::vector<double> P((size_t)3 * n_lm);
    for (int l = 0; l < n_lm; ++l) {
        first[l] = (int)((uint64_t)rng() % (uint64_t)std::max(1, n_kf - 2));
        const double* Q = &T[(size_t)12 * first[l]];
        const double pc[3] = {6.0 * U(rng), 1.5 * U(rng), 6.0 + 25.0 * (U(rng) + 1.0)};
        for (int r = 0; r < 3; ++r) P[3 * l + r] = Q[3 * r] * pc[0] + Q[3 * r + 1] * pc[1] + Q[3 * r + 2] * pc[2] + Q[9 + r];
        const double pn[3] = {P[3 * l] + 0.05 * N(rng), P[3 * l + 1] + 0.05 * N(rng), P[3 * l + 2] + 0.05 * N(rng)};
        CHECK(svi_ba_add_landmark(ba, l, pn, 0));
    }
    for (int k = 0; k < n_kf; ++k) {
        if (k > 0) CHECK(svi_ba_add_keyframe(ba, 1000000 + k, 1000000 + k - 1, &Tn[(size_t)12 * k], nullptr, nullptr));
        std::vector<int64_t> ids;
        std::vector<float> uvL, uvR;
        std::vector<double> xyz;
        const double* Q = &T[(size_t)12 * k];
        for (int l = 0; l < n_lm; ++l) {
            if (k < first[l] || k >= first[l] + track) continue;
            double pc[3];
            for (int c = 0; c < 3; ++c) pc[c] = Q[c] * (P[3 * l] - Q[9]) + Q[3 + c] * (P[3 * l + 1] - Q[10]) + Q[6 + c] * (P[3 * l + 2] - Q[11]);
            if (pc[2] < 1.0) continue;
            const float u = (float)(fx * pc[0] / pc[2] + cx + 0.3 * N(rng)), v = (float)(fx * pc[1] / pc[2] + cy + 0.3 * N(rng));
            const float d = (float)std::max(1.0, std::rint(fb / pc[2]));
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

static int run_rank(int rank, int n_ranks, int n_kf, int n_lm, const char* id_file, int shards)
{
    g_rank = rank;
    const int world = n_ranks > 0 ? n_ranks : 1;
    const int n_dev = svi_device_count();
    if (n_dev <= 0) { std::fprintf(stderr, "no device\n"); return 3; }
    const int device = rank % n_dev;
    svi_rccl* comm = nullptr;
    if (n_ranks > 0) {
        char id[128];
        if (rank == 0) {
            CHECK(svi_rccl_unique_id(id));
            const std::string tmp = std::string(id_file) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, 128, f) != 128) return 2;
            fclose(f);
            if (rename(tmp.c_str(), id_file) != 0) return 2;
        } else {
            FILE* f = nullptr;
            for (int spin = 0; spin < 600 && !(f = fopen(id_file, "rb")); ++spin) usleep(100000);
            if (!f || fread(id, 1, 128, f) != 128) return 2;
            fclose(f);
        }
        CHECK(svi_rccl_create(id, rank, world, device, &comm));
    }
    svi_ba_options o;
    svi_ba_options_default(&o);
    o.fx = o.fy = 718.856; o.cx = 607.1928; o.cy = 185.2157; o.baseline_m = 0.54;
    o.device = device; o.rank = rank; o.n_ranks = shards > 0 ? shards : world;
    svi_ba* ba = nullptr;
    CHECK(svi_ba_create(&o, &ba));
    if (build(ba, n_kf, n_lm)) return 1;
    if (comm) CHECK(svi_ba_set_allreduce(ba, svi_rccl_allreduce, comm));
    CHECK(svi_ba_initialize(ba));
    uint64_t nominal = 0, executed = 0;
    CHECK(svi_ba_optimize_until(ba, 0.99, 1, 10, &nominal, &executed));
    double plain = 0, robust = 0;
    CHECK(svi_ba_chi2(ba, &plain, &robust));
    int64_t np = 0;
    CHECK(svi_ba_num_poses(ba, &np));
    std::vector<double> T((size_t)12 * np);
    CHECK(svi_ba_get_poses(ba, nullptr, T.data()));   // collective with several ranks (gathers the landmark shards too)
    double sum = 0;
    for (double x : T) sum += x;
    svi_ba_stats st;
    CHECK(svi_ba_get_stats(ba, &st));
    std::printf("rank %d/%d: %llu %llu %.15g %.15g %.15g local_landmarks %lld\n", rank, world, (unsigned long long)nominal, (unsigned long long)executed, plain,
                robust, sum, (long long)st.n_landmarks_local);
    std::fflush(stdout);
    CHECK(svi_ba_destroy(ba));
    if (comm) CHECK(svi_rccl_destroy(comm));
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s n_ranks n_kf n_lm id_file\n", argv[0]); return 2; }
    const int n_ranks = std::atoi(argv[1]), n_kf = std::atoi(argv[2]), n_lm = std::atoi(argv[3]), shards = argc > 5 ? std::atoi(argv[5]) : 0;
    if (n_ranks <= 1) return run_rank(0, n_ranks, n_kf, n_lm, argv[4], shards);
    // one process per rank, forked before anything touches the GPU
    std::vector<pid_t> kids;
    for (int r = 0; r < n_ranks; ++r) {
        const pid_t p = fork();
        if (p < 0) return 2;
        if (p == 0) _exit(run_rank(r, n_ranks, n_kf, n_lm, argv[4], shards));
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1; }
    return bad;
}
