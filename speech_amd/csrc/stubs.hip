// stubs.hip -- entry points declared in include/speech_amd.h whose kernels are not written yet.
// They FAIL LOUDLY (CTC_STATUS_EXECUTION_FAILED); nothing falls back to a CPU path.  Each stub is deleted
// when its kernel lands.
#include "common.h"

extern "C" {
size_t sa_ctc_beam_workspace_bytes(int, int, int, int) { return 0; }
ctcStatus_t sa_ctc_beam_decode(const float*, long, long, const int*, int, int, int, int, int, int, int*, int*, float*,
                               void*, size_t, void*) { return CTC_STATUS_EXECUTION_FAILED; }
ctcStatus_t sa_ctc_greedy_decode(const float*, long, long, const int*, int, int, int, int, int*, int*, void*) {
    return CTC_STATUS_EXECUTION_FAILED;
}
ctcStatus_t sa_gemm_f32(int, int, int, int, int, float, const float*, long, const float*, long, float, float*, long,
                        const float*, void*) { return CTC_STATUS_EXECUTION_FAILED; }
ctcStatus_t sa_conv2d_relu_fwd(const float*, const float*, const float*, float*, int, int, int, int, int, int, int,
                               int, long, long, long, void*) { return CTC_STATUS_EXECUTION_FAILED; }
size_t sa_conv2d_bwd_workspace_bytes(int, int, int, int, int, int, int, int) { return 0; }
ctcStatus_t sa_conv2d_relu_bwd(const float*, const float*, const float*, const float*, float*, float*, float*, int,
                               int, int, int, int, int, int, int, long, long, long, void*, size_t, void*) {
    return CTC_STATUS_EXECUTION_FAILED;
}
ctcStatus_t sa_gru_fwd(const float*, const float*, const float*, float*, long, long, float*, int, int, int, int,
                       void*) { return CTC_STATUS_EXECUTION_FAILED; }
size_t sa_gru_bwd_workspace_bytes(int, int, int) { return 0; }
ctcStatus_t sa_gru_bwd(const float*, long, long, const float*, const float*, const float*, float*, float*, int, int,
                       int, int, void*, size_t, void*) { return CTC_STATUS_EXECUTION_FAILED; }
ctcStatus_t sa_colsum_f32(const float*, long, int, int, float*, int, void*) { return CTC_STATUS_EXECUTION_FAILED; }
ctcStatus_t sa_add_rows_f32(const float*, long, const float*, long, float*, long, int, int, void*) {
    return CTC_STATUS_EXECUTION_FAILED;
}
size_t sa_sgd_workspace_bytes(size_t) { return 0; }
ctcStatus_t sa_clip_sgd_step(float*, float*, float*, size_t, float, float, float, float, float*, void*, size_t,
                             void*) { return CTC_STATUS_EXECUTION_FAILED; }
}
