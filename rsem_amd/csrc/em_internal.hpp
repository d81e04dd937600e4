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

// Where the hot-loop layout keeps the values of caller row i, for a producer that writes them in place (model.hip's round
// kernel).  shapes points at the device copy of the layout's Shape table (sell_shape.hpp).
struct EmPlanesView {
    const uint32_t* d_rank;  // caller row -> sorted row
    const void* d_shapes;
    int n_shapes;
    uint32_t T, n_sell_rows;
    unsigned char* d_sval;
    double* d_sncp;
};

// device pointers of the ctx (d_w / d_wn are NULL until a weights pass has run)
int em_device_view(rsem_em_ctx* c, EmDeviceView* v);
// A model context keeps the view's d_sid / d_cp for its lifetime: between hold and release the EM context does not free them
// (option "release_csr" is refused).
void em_view_hold(rsem_em_ctx* c);
void em_view_release(rsem_em_ctx* c);
// d_cp / d_ncp were rewritten on the device: rebuild the sliced value planes
int em_values_changed(rsem_em_ctx* c);
// The planes of the current layout, for in-place writers.  RSEM_ERR_STATE when the layout cannot take doubles in place (Q32
// shapes) -- the caller then falls back to em_values_changed.  After writing d_cp / d_ncp AND the planes:
int em_planes_view(rsem_em_ctx* c, EmPlanesView* v);
// Can the layout take doubles in place (F64 planes, no split rows)?  A question, not an error: nothing is written to the
// last-error string (callers that only want to know must not use em_planes_view as a probe).
bool em_planes_writable(const rsem_em_ctx* c);
int em_values_written_in_place(rsem_em_ctx* c);

}  // namespace rsem
