// kernel_stubs.cpp — the launchers of ba_kernels.hip / ba_chol.hip as no-ops: the sanitizer build has no device code, it
// exercises everything the host does AROUND the launches (tools/host_san/Makefile).  Test infrastructure only.
#include "ba_host.h"

namespace svi {
void ba_linearize_lm(const BaDev&, int, void*) {}
void ba_linearize_pose(const BaDev&, int, void*) {}
void ba_linearize_aux(const BaDev&, int, int, void*) {}
void ba_chi2_aux(const BaDev&, int, int, void*) {}
void ba_pose_finalize(const BaDev&, const int*, int, int, int, double, void*) {}
void ba_publish(const BaDev&, int, double*, int*, int, void*) {}
void ba_lin_post(const BaDev&, int, void*) {}
void ba_invert_landmarks(const BaDev&, double, void*) {}
void ba_schur(const BaDev&, const StageSignals*, unsigned long long, int, int, void*) {}
void ba_assemble(const BaDev&, int, int, int, void*) {}
void ba_update_poses(const BaDev&, int, double, int, int, void*) {}
void ba_backsub_chi2(const BaDev&, int, double, void*) {}
void ba_chi2_only(const BaDev&, int, void*) {}
void ba_reduce_trial_scalars(const BaDev&, int, int, int, double*, int*, int, void*) {}
void ba_debug_jacobians(const BaDev&, int, const int*, double*, double*, double*, void*) {}
void ba_debug_aux_jacobians(const BaDev&, int, double*, double*, double*, double*, double*, void*) {}
void ba_gather_edges(const double*, const uint8_t*, const int*, const int*, int, int, int, double*, uint8_t*, const int*, int*, void*) {}
void ba_configure_kernels(int) {}
int chol_potrf_probe(int, int, int, double*) { return 0; }
int chol_factor_solve(const CholPlan&, double*, double*, double*, double*, double*, double, int, int*, void*, const PoseTail*, int* done) { if (done) *done = 0; return 0; }
int chol_factor_range(const CholPlan&, double*, double*, double*, double*, double*, double, int, int*, void*, const PoseTail*, int* done, int, int, int, double*, double*) { if (done) *done = 0; return 0; }
} // namespace svi
