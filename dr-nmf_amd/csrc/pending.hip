// Entry points declared in include/drnmf.h whose kernels are not written yet: they fail loudly.
#include "common.h"

extern "C" size_t drnmf_ista_workspace_bytes(int64_t, int32_t, int32_t) { return 0; }
extern "C" int32_t drnmf_ista_forward(drnmf_handle_t h, int64_t, int32_t, int32_t, int32_t, int32_t,
                                      float, float, float, const float*, const float*, float*,
                                      void*, size_t, void*) {
    DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "drnmf_ista_forward: not implemented yet");
}
extern "C" size_t drnmf_mu_workspace_bytes(int64_t, int32_t, int32_t) { return 0; }
extern "C" int32_t drnmf_mu_forward(drnmf_handle_t h, int64_t, int32_t, int32_t, int32_t, float,
                                    float, const float*, const float*, float*, float*, float*,
                                    void*, size_t, void*) {
    DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "drnmf_mu_forward: not implemented yet");
}
extern "C" int32_t drnmf_stft_frames(int64_t nsampl, int32_t N, int32_t hop) {
    if (nsampl < 0 || N <= 0 || hop <= 0) return -1;
    const int64_t nfram = (nsampl + hop - 1) / hop;
    return (int32_t)(1 + (nfram * hop + 2 * (int64_t)N - N) / hop);
}
extern "C" int32_t drnmf_stft_mag(drnmf_handle_t h, int32_t, int64_t, int32_t, int32_t, int32_t,
                                  const void*, float*, void*) {
    DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "drnmf_stft_mag: not implemented yet");
}
