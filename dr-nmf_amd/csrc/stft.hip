// STFT-magnitude front end on gfx950.
//
// Reference: wavread's /32768 scaling (util.py:29-35), stft_mc (util.py:171-201: pad the signal
// to a multiple of hop, N zeros on both sides, librosa stft(center=False)), the sqrt-Hann window
// (audio_dataset.py:194) and the 'mag' transform sqrt(re^2 + im^2) (audio_dataset.py:22-23).
// librosa 0.5.1 conjugates the spectrum; the magnitude this path produces is unaffected.
//
// One workgroup per (signal, frame): window + zero-padded framing on load, an in-LDS radix-2
// FFT (bit-reversed load, log2 N butterfly stages, twiddles from a per-workgroup table), and the
// magnitude of bins 0..N/2 written as one coalesced row.  The workload is HBM-trivial
// (2 bytes in, ~2 floats out per sample-hop); fp32 arithmetic as in the reference (scipy
// fftpack on float32 input).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
stft_mag_kernel(const void* __restrict__ pcm, int is_int16, int64_t nsampl, int N, int logN, int hop,
                int nf, float* __restrict__ mag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* buf = (float2*)smem;            // N complex points
    float2* tw = buf + N;                   // N/2 twiddles e^{-2 pi i k / N}
    const int tid = threadIdx.x;
    const int frame = blockIdx.x, sig = blockIdx.y;
    const int64_t base = (int64_t)frame * hop - N;   // first sample of the frame (N leading zeros)

    for (int k = tid; k < N / 2; k += 256) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)N, &s, &c);
        tw[k] = make_float2(c, s);
    }
    for (int i = tid; i < N; i += 256) {
        const int64_t idx = base + i;
        float v = 0.f;
        if (idx >= 0 && idx < nsampl) {
            v = is_int16 ? (float)((const short*)pcm)[(size_t)sig * nsampl + idx] / 32768.0f
                         : ((const float*)pcm)[(size_t)sig * nsampl + idx];
        }
        // sqrt(hann(N, sym=False)) with the Hann value rounded to float32 first
        const float hann = (float)(0.5 - 0.5 * cospi(2.0 * (double)i / (double)N));
        v *= sqrtf(hann);
        const unsigned rev = __brev((unsigned)i) >> (32 - logN);
        buf[rev] = make_float2(v, 0.f);
    }
    __syncthreads();
    for (int s = 1; s <= logN; ++s) {
        const int half = 1 << (s - 1);
        const int tstride = N >> s;
        for (int b = tid; b < N / 2; b += 256) {
            const int pos = b & (half - 1);
            const int i0 = ((b >> (s - 1)) << s) + pos, i1 = i0 + half;
            const float2 w = tw[pos * tstride];
            const float2 u = buf[i0], v = buf[i1];
            const float2 t = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            buf[i0] = make_float2(u.x + t.x, u.y + t.y);
            buf[i1] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
    float* out = mag + ((size_t)sig * nf + frame) * (N / 2 + 1);
    for (int k = tid; k <= N / 2; k += 256) {
        const float2 z = buf[k];
        out[k] = sqrtf(z.x * z.x + z.y * z.y);
    }
}

}  // namespace

extern "C" int32_t drnmf_stft_frames(int64_t nsampl, int32_t N, int32_t hop) {
    if (nsampl < 0 || N <= 0 || hop <= 0) return -1;
    const int64_t nfram = (nsampl + hop - 1) / hop;               // util.py:183
    return (int32_t)(1 + (nfram * hop + 2 * (int64_t)N - N) / hop);   // librosa framing
}

extern "C" int32_t drnmf_stft_mag(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, int32_t N,
                                  int32_t hop, int32_t is_int16, const void* pcm, float* mag,
                                  void* stream_) {
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_sig <= 0 || nsampl <= 0 || hop <= 0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "stft_mag: bad shape n_sig=%d nsampl=%lld hop=%d",
                   n_sig, (long long)nsampl, hop);
    if (N < 64 || N > 4096 || (N & (N - 1)))
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "stft_mag: N=%d must be a power of two in [64,4096]",
                   N);
    if (!pcm || !mag) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "stft_mag: NULL pointer argument");
    int logN = 0;
    while ((1 << logN) < N) ++logN;
    const int nf = drnmf_stft_frames(nsampl, N, hop);
    const size_t shmem = (size_t)(N + N / 2) * sizeof(float2);
    hipLaunchKernelGGL(stft_mag_kernel, dim3((unsigned)nf, (unsigned)n_sig), dim3(256), shmem,
                       (hipStream_t)stream_, pcm, is_int16, nsampl, N, logN, hop, nf, mag);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}
