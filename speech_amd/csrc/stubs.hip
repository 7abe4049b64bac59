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
}
