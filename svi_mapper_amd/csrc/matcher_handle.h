// matcher_handle.h — the svi_matcher handle shared by matcher.hip and tracker.hip.
#pragma once
#include <vector>

#include "common.h"

struct svi_matcher {
    int         device = 0;
    hipStream_t stream = nullptr;
    bool        own_stream = false;
    int         n_cu = 256;
    int         gate_path = 0; // svi_matcher_set_gate_path
    svi::DevBuf keys;      // split-mode packed minima
    svi::DevBuf scratch;   // host-pointer entry points stage through here
    svi::DevBuf track;     // tracker.hip: fundamental matrices + uploaded transforms of one plan call
    std::vector<double> track_host; // host copy of those transforms while their upload is in flight
    hipEvent_t  track_ev = nullptr; // recorded after the upload
};
