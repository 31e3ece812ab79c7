// Entry points declared in include/drnmf.h whose kernels are not written yet: they fail loudly.
#include "common.h"

extern "C" int32_t drnmf_stft_frames(int64_t nsampl, int32_t N, int32_t hop) {
    if (nsampl < 0 || N <= 0 || hop <= 0) return -1;
    const int64_t nfram = (nsampl + hop - 1) / hop;
    return (int32_t)(1 + (nfram * hop + 2 * (int64_t)N - N) / hop);
}
extern "C" int32_t drnmf_stft_mag(drnmf_handle_t h, int32_t, int64_t, int32_t, int32_t, int32_t,
                                  const void*, float*, void*) {
    DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "drnmf_stft_mag: not implemented yet");
}
