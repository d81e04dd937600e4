// em_internal.hpp -- what model.hip needs from an rsem_em_ctx (defined in em.hip) without seeing it.
#pragma once
#include "common.hpp"

namespace rsem {

struct EmDeviceView {
    int device;
    hipStream_t stream;
    int32_t M;
    uint64_t N1, nnz;
    const uint64_t* d_row_ptr;  // caller-order CSR
    const int32_t* d_sid;
    double* d_cp;               // conprb per alignment (caller order)
    double* d_ncp;              // noise conprb per read
    double* d_w;                // posterior weight per alignment / per read of the last weights pass
    double* d_wn;
};

// device pointers of the ctx (allocates the weight buffers on first use)
int em_device_view(rsem_em_ctx* c, EmDeviceView* v);
// d_cp / d_ncp were rewritten on the device: rebuild the sliced value planes
int em_values_changed(rsem_em_ctx* c);
// E step with posterior write-back into d_w / d_wn (EM.cpp:199-244, calcExpectedWeights-style) followed by the M
// step; host outputs as rsem_em_step.  The weights stay on the device for the model accumulation kernels.
int em_step_with_weights(rsem_em_ctx* c, const double* theta, double N0, double* counts, double* theta_new,
                         double* sum, double* bChange, int32_t* totNum);

}  // namespace rsem
