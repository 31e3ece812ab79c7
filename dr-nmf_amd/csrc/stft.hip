// STFT-magnitude front end on gfx950.
//
// Reference: wavread's /32768 scaling (util.py:29-35), stft_mc (util.py:171-201: pad the signal
// to a multiple of hop, N zeros on both sides, librosa stft(center=False)), the sqrt-Hann window
// (audio_dataset.py:194) and the 'mag' transform sqrt(re^2 + im^2) (audio_dataset.py:22-23).
// librosa 0.5.1 conjugates the spectrum; the magnitude this path produces is unaffected.
//
// One workgroup per (signal, frame): window + zero-padded framing on load, an in-LDS radix-2
// FFT (bit-reversed load, two butterfly stages per pass; window and twiddles from per-size tables), and the
// magnitude of bins 0..N/2 written as one coalesced row.  fp32 arithmetic as in the reference (scipy
// fftpack on float32 input).  Measured (bench.py extra.stft_front_end, 64 x 10 s at 16 kHz, N = 1024,
// hop = 256): 217 us = 186 M frames/s, 475 GB/s in + out -- 5.9 % of the HBM rate: the FFT passes are
// LDS-latency / barrier bound (two radix-2 stages per pass: 293 -> 217 us together with the tables),
// not memory bound; the front end is ~600x faster than the recurrent cell consumes frames.
#include "common.h"

namespace {

// in-place radix-2 DIT FFT of N complex points in LDS (input already bit-reversed); tw[k] =
// e^{-2 pi i k / N}, k < N/2
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

__device__ __forceinline__ void fft_lds(float2* buf, const float2* tw, int N, int logN, int tid) {
    int s = 1;
    if (logN & 1) {                      // odd log2 N: one radix-2 stage first
        for (int b = tid; b < N / 2; b += 256) {
            const float2 u = buf[2 * b], v = buf[2 * b + 1];      // twiddle 1
            buf[2 * b] = make_float2(u.x + v.x, u.y + v.y);
            buf[2 * b + 1] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        s = 2;
    }
    // two radix-2 stages (s, s+1) per pass, the intermediate values in registers: half the LDS
    // round trips and barriers of the stage-by-stage loop (the kernel is bound by those, not by
    // memory); the butterflies and their order are exactly the radix-2 ones
    for (; s < logN; s += 2) {
        const int half = 1 << (s - 1);
        const int ts1 = N >> s, ts2 = N >> (s + 1);
        for (int b = tid; b < N / 4; b += 256) {
            const int pos = b & (half - 1);
            const int base = ((b >> (s - 1)) << (s + 1)) + pos;
            const float2 w1 = tw[pos * ts1], w2 = tw[pos * ts2], w3 = tw[(pos + half) * ts2];
            const float2 p0 = buf[base], p1 = buf[base + half], p2 = buf[base + 2 * half],
                         p3 = buf[base + 3 * half];
            const float2 t1 = cmul(p1, w1), t3 = cmul(p3, w1);
            const float2 a0 = make_float2(p0.x + t1.x, p0.y + t1.y);
            const float2 a1 = make_float2(p0.x - t1.x, p0.y - t1.y);
            const float2 a2 = make_float2(p2.x + t3.x, p2.y + t3.y);
            const float2 a3 = make_float2(p2.x - t3.x, p2.y - t3.y);
            const float2 u2 = cmul(a2, w2), u3 = cmul(a3, w3);
            buf[base] = make_float2(a0.x + u2.x, a0.y + u2.y);
            buf[base + 2 * half] = make_float2(a0.x - u2.x, a0.y - u2.y);
            buf[base + half] = make_float2(a1.x + u3.x, a1.y + u3.y);
            buf[base + 3 * half] = make_float2(a1.x - u3.x, a1.y - u3.y);
        }
        __syncthreads();
    }
}

// Window and twiddle tables per FFT size (64 .. 4096 = 2^6 .. 2^12), in static device memory and
// refilled by every call (a handful of threads; concurrent callers write identical values): the
// per-frame workgroups read them from L2 instead of evaluating 2N transcendental functions each
// (the window needs a double-precision cospi to reproduce float32(hann) exactly).
constexpr int TAB_LOG_MIN = 6, TAB_LOG_MAX = 12, TAB_N_MAX = 1 << TAB_LOG_MAX;
__device__ float g_window[TAB_LOG_MAX - TAB_LOG_MIN + 1][TAB_N_MAX];
__device__ float2 g_twiddle[TAB_LOG_MAX - TAB_LOG_MIN + 1][TAB_N_MAX / 2];

__device__ __forceinline__ float sqrt_hann(int i, int N) {
    // sqrt(hann(N, sym=False)) with the Hann value rounded to float32 first (audio_dataset.py:194)
    const float hann = (float)(0.5 - 0.5 * cospi(2.0 * (double)i / (double)N));
    return sqrtf(hann);
}

__global__ void __launch_bounds__(256) fft_tables_kernel(int N, int logN) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) g_window[logN - TAB_LOG_MIN][i] = sqrt_hann(i, N);
    if (i < N / 2) {
        float sn, cs;
        sincospif(-2.0f * (float)i / (float)N, &sn, &cs);
        g_twiddle[logN - TAB_LOG_MIN][i] = make_float2(cs, sn);
    }
}

__global__ void __launch_bounds__(256)
stft_kernel(const void* __restrict__ pcm, int is_int16, int64_t nsampl, int N, int logN, int hop,
            int nf, float* __restrict__ mag, float* __restrict__ re, float* __restrict__ im) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* buf = (float2*)smem;            // N complex points
    float2* tw = buf + N;                   // N/2 twiddles e^{-2 pi i k / N}
    const int tid = threadIdx.x;
    const int frame = blockIdx.x, sig = blockIdx.y;
    const int64_t base = (int64_t)frame * hop - N;   // first sample of the frame (N leading zeros)

    const float* __restrict__ win = g_window[logN - TAB_LOG_MIN];
    for (int k = tid; k < N / 2; k += 256) tw[k] = g_twiddle[logN - TAB_LOG_MIN][k];
    for (int i = tid; i < N; i += 256) {
        const int64_t idx = base + i;
        float v = 0.f;
        if (idx >= 0 && idx < nsampl) {
            v = is_int16 ? (float)((const short*)pcm)[(size_t)sig * nsampl + idx] / 32768.0f
                         : ((const float*)pcm)[(size_t)sig * nsampl + idx];
        }
        v *= win[i];
        const unsigned rev = __brev((unsigned)i) >> (32 - logN);
        buf[rev] = make_float2(v, 0.f);
    }
    __syncthreads();
    fft_lds(buf, tw, N, logN, tid);
    const size_t o = ((size_t)sig * nf + frame) * (N / 2 + 1);
    for (int k = tid; k <= N / 2; k += 256) {
        const float2 z = buf[k];
        if (mag) mag[o + k] = sqrtf(z.x * z.x + z.y * z.y);
        if (re) re[o + k] = z.x;
        if (im) im[o + k] = -z.y;    // librosa 0.5.1 conjugates the spectrum (util.py:195 via stft)
    }
}


// ---------------------------------------------------------------------------------------------
// Fast path for the sizes the reference uses (N = 512 shipped, N = 1024 in BASELINE): the input is
// real, so a frame is ONE complex FFT of M = N/2 points, z[n] = x[2n] + i x[2n+1], followed by the
// split X[k] = (Z[k] + conj Z[M-k]) / 2 - i e^{-2 pi i k / N} (Z[k] - conj Z[M-k]) / 2.  The M-point
// FFT is a Stockham autosort of radix-R passes (R = 8 for M = 512, R = 4 for M = 256) with the R
// points of a butterfly in registers: M / R = 64 butterflies = ONE WAVE per frame, so the passes
// exchange through a wave-private LDS slice with no workgroup barrier; a 256-thread workgroup
// carries 4 frames.  Against the radix-2 kernel above (one workgroup per frame, a complex FFT twice
// the size, a barrier every two stages) it moves the front end from LDS latency towards HBM.
template <int R>
__device__ __forceinline__ void dft_r(float2 (&v)[R]);

template <>
__device__ __forceinline__ void dft_r<4>(float2 (&v)[4]) {
    const float2 a = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
    const float2 b = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    const float2 c = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
    const float2 d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
    v[0] = make_float2(a.x + c.x, a.y + c.y);
    v[2] = make_float2(a.x - c.x, a.y - c.y);
    v[1] = make_float2(b.x + d.y, b.y - d.x);      // b - i d
    v[3] = make_float2(b.x - d.y, b.y + d.x);      // b + i d
}

template <>
__device__ __forceinline__ void dft_r<8>(float2 (&v)[8]) {
    float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft_r<4>(e);
    dft_r<4>(o);
    const float h = 0.70710678118654752440f;
    // o[r] *= e^{-2 pi i r / 8}
    o[1] = make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
    o[2] = make_float2(o[2].y, -o[2].x);
    o[3] = make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = make_float2(e[r].x + o[r].x, e[r].y + o[r].y);
        v[r + 4] = make_float2(e[r].x - o[r].x, e[r].y - o[r].y);
    }
}

template <int R, int P>       // M = R^P complex points, N = 2M real samples
__global__ void __launch_bounds__(256)
stft_real_kernel(const void* __restrict__ pcm, int is_int16, int64_t nsampl, int logN, int hop,
                 int nf, float* __restrict__ mag, float* __restrict__ re, float* __restrict__ im) {
    constexpr int M = (R == 8 ? (P == 3 ? 512 : 64) : (P == 4 ? 256 : 64)), N = 2 * M;
    __shared__ float2 tw[N / 2];              // e^{-2 pi i k / N}, k < N/2
    // per wave: ONE buffer, rewritten in place by every pass -- a wave executes its LDS reads of a
    // pass (all 64 lanes, into registers) before that pass's writes are issued, and LDS serves one
    // wave's requests in order, so the Stockham reorder needs no second buffer (the ping-pong pair
    // it replaced held the kernel to 3 workgroups per CU; 7 now).  Index i lives at i + i/8: the
    // scatter of the early passes (stride R, then R*R) would otherwise put 8-16 lanes on one bank
    constexpr int MP = M + M / 8;
    __shared__ float2 bufs[4][MP];
    auto pad = [](int i) { return i + (i >> 3); };
    const int tid = threadIdx.x, wv = tid >> 6, j = tid & 63;
    const int frame = blockIdx.x * 4 + wv, sig = blockIdx.y;
    for (int k = tid; k < N / 2; k += 256) tw[k] = g_twiddle[logN - TAB_LOG_MIN][k];
    __syncthreads();
    if (frame >= nf) return;                  // (whole waves: no barrier below)
    const float* __restrict__ win = g_window[logN - TAB_LOG_MIN];
    auto twid = [&](int idx) {                // e^{-2 pi i idx / N}, 0 <= idx < N
        const float2 t = tw[idx & (N / 2 - 1)];
        return idx >= N / 2 ? make_float2(-t.x, -t.y) : t;
    };
    // pass 0 input straight from the signal: z[n] = (x[2n] w[2n], x[2n+1] w[2n+1]), n = j + r M/R
    const int64_t base = (int64_t)frame * hop - N;       // first sample of the frame (N leading zeros)
    float2 v[R];
    // interior frames (all but the first N/hop and the last few): no bounds tests, one 8-byte load
    // per sample pair -- this kernel is VALU-issue bound (~1000 instructions per lane and frame),
    // the 16 64-bit range tests were a tenth of them
    const size_t e0 = (size_t)sig * nsampl + (size_t)(base > 0 ? base : 0);   // first element
    const bool interior = base >= 0 && base + N <= nsampl &&
                          ((((uintptr_t)pcm >> (is_int16 ? 1 : 2)) + e0) & 1) == 0;
    if (interior && is_int16) {
        const short* p = (const short*)pcm + e0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = j + r * (M / R);
            const short2 x2 = *(const short2*)(p + 2 * n);
            const float2 w2 = *(const float2*)(win + 2 * n);
            v[r] = make_float2((float)x2.x / 32768.0f * w2.x, (float)x2.y / 32768.0f * w2.y);
        }
    } else if (interior) {
        const float* p = (const float*)pcm + e0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = j + r * (M / R);
            const float2 x2 = *(const float2*)(p + 2 * n);
            const float2 w2 = *(const float2*)(win + 2 * n);
            v[r] = make_float2(x2.x * w2.x, x2.y * w2.y);
        }
    } else
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = j + r * (M / R);
        const int64_t i0 = base + 2 * n;
        float x0 = 0.f, x1 = 0.f;
        if (is_int16) {
            const short* p = (const short*)pcm + (size_t)sig * nsampl;
            if (i0 >= 0 && i0 < nsampl) x0 = (float)p[i0] / 32768.0f;
            if (i0 + 1 >= 0 && i0 + 1 < nsampl) x1 = (float)p[i0 + 1] / 32768.0f;
        } else {
            const float* p = (const float*)pcm + (size_t)sig * nsampl;
            if (i0 >= 0 && i0 < nsampl) x0 = p[i0];
            if (i0 + 1 >= 0 && i0 + 1 < nsampl) x1 = p[i0 + 1];
        }
        const float2 w2 = *(const float2*)(win + 2 * n);
        v[r] = make_float2(x0 * w2.x, x1 * w2.y);
    }
    float2* cur = bufs[wv];
    float2* nxt = bufs[wv];
    int Ns = 1;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (p > 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = cur[pad(j + r * (M / R))];
        }
        const int k = j & (Ns - 1);
        if (p > 0) {                          // twiddles e^{-2 pi i k r / (Ns R)} (pass 0: k = 0)
            const int step = k * (N / (Ns * R));
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], twid(step * r));
        }
        dft_r<R>(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) nxt[pad(j0 + r * Ns)] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        Ns *= R;
    }
    // split + output: k = j + 64 i, i < M/64, and k = M
    const size_t o = ((size_t)sig * nf + frame) * (M + 1);
    auto emit = [&](int k) {
        const float2 zk = cur[pad(k & (M - 1))];
        const float2 zc = cur[pad((M - k) & (M - 1))];     // Z[M-k] (Z[M] = Z[0])
        const float2 a = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));   // (Z_k + conj Z_{M-k}) / 2
        const float2 b = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y));   // (Z_k - conj Z_{M-k}) / 2
        const float2 w = k == M ? make_float2(-1.f, 0.f) : tw[k];
        const float2 wb = cmul(b, w);                       // X = a - i w b
        const float xr = a.x + wb.y, xi = a.y - wb.x;
        if (mag) mag[o + k] = sqrtf(xr * xr + xi * xi);
        if (re) re[o + k] = xr;
        if (im) im[o + k] = -xi;      // librosa 0.5.1 conjugates the spectrum (util.py:195 via stft)
    };
#pragma unroll
    for (int i = 0; i < M / 64; ++i) emit(j + 64 * i);
    if (j == 0) emit(M);
}

static bool stft_fast(int N) { return N == 512 || N == 1024; }

// The window / twiddle tables of a size are filled once per handle (= per device); later calls on
// any stream only wait for the event recorded behind that fill.
static int32_t ensure_fft_tables(drnmf_handle_t h, int N, int logN, hipStream_t stream) {
    const int slot = logN - TAB_LOG_MIN;
    if (!h->fft_ready[slot]) {
        hipLaunchKernelGGL(fft_tables_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0,
                           stream, N, logN);
        DRNMF_HIP(h, hipGetLastError());
        DRNMF_HIP(h, hipEventCreateWithFlags(&h->fft_event[slot], hipEventDisableTiming));
        DRNMF_HIP(h, hipEventRecord(h->fft_event[slot], stream));
        h->fft_ready[slot] = true;
    } else {
        DRNMF_HIP(h, hipStreamWaitEvent(stream, h->fft_event[slot], 0));
    }
    return DRNMF_OK;
}
static void launch_stft_real(int N, int logN, const void* pcm, int is_int16, int64_t nsampl, int hop,
                             int nf, int n_sig, float* mag, float* re, float* im, hipStream_t st) {
    const dim3 grid((unsigned)((nf + 3) / 4), (unsigned)n_sig);
    if (N == 1024)
        hipLaunchKernelGGL((stft_real_kernel<8, 3>), grid, dim3(256), 0, st, pcm, is_int16, nsampl,
                           logN, hop, nf, mag, re, im);
    else
        hipLaunchKernelGGL((stft_real_kernel<4, 4>), grid, dim3(256), 0, st, pcm, is_int16, nsampl,
                           logN, hop, nf, mag, re, im);
}

// one workgroup per (signal, frame): masked spectrum -> Hermitian extension -> inverse FFT -> real
// part * window * 2/(N/hop)  (util.py:48-169 istft_noDiv with center=False)
__global__ void __launch_bounds__(256)
istft_frames_kernel(const float* __restrict__ re, const float* __restrict__ im,
                    const float* __restrict__ mask, int N, int logN, int hop, int nf,
                    float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* buf = (float2*)smem;
    float2* tw = buf + N;
    const int tid = threadIdx.x;
    const int frame = blockIdx.x, sig = blockIdx.y;
    const int F = N / 2 + 1;
    const float* __restrict__ win = g_window[logN - TAB_LOG_MIN];
    for (int k = tid; k < N / 2; k += 256) tw[k] = g_twiddle[logN - TAB_LOG_MIN][k];
    // ifft(z) = conj(fft(conj(z)))/N.  With S the stored (conjugated) spectrum the reference
    // builds z = [conj(S_0..S_{N/2}), S_{N/2-1}..S_1], so conj(z) = [S_k ; conj(S_{N-k})].
    const size_t o = ((size_t)sig * nf + frame) * F;
    for (int k = tid; k < N; k += 256) {
        const int kk = k <= N / 2 ? k : N - k;
        const float m = mask ? mask[o + kk] : 1.f;
        float2 z = make_float2(m * re[o + kk], m * im[o + kk]);
        if (k > N / 2) z.y = -z.y;
        const unsigned rev = __brev((unsigned)k) >> (32 - logN);
        buf[rev] = z;
    }
    __syncthreads();
    fft_lds(buf, tw, N, logN, tid);
    const float scale = (2.0f / ((float)N / (float)hop)) / (float)N;
    float* out = frames + ((size_t)sig * nf + frame) * N;
    for (int i = tid; i < N; i += 256) out[i] = buf[i].x * scale * win[i];
}

// overlap-add by gathering (deterministic), with istft_mc's trimming of the N padding samples on
// both sides and the crop to nsampl (util.py:203-226)
__global__ void __launch_bounds__(256)
overlap_add_kernel(const float* __restrict__ frames, int N, int hop, int nf, int64_t nsampl,
                   float* __restrict__ y) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int sig = blockIdx.y;
    if (s >= nsampl) return;
    const int64_t p = s + N;                                  // index in the untrimmed signal
    const int64_t total = (int64_t)N + (int64_t)hop * (nf - 1);
    float acc = 0.f;
    if (p < total - N) {
        int64_t j0 = (p - N + hop) / hop;                     // ceil((p - N + 1) / hop)
        if (j0 < 0) j0 = 0;
        int64_t j1 = p / hop;
        if (j1 > nf - 1) j1 = nf - 1;
        for (int64_t j = j0; j <= j1; ++j)
            acc += frames[((size_t)sig * nf + j) * N + (p - j * hop)];
    }
    y[(size_t)sig * nsampl + s] = acc;
}

// SNR = 10 log10(sum ref^2 / sum (ref - est)^2)  (score_audio.m:209), one workgroup per signal
__global__ void __launch_bounds__(256)
snr_kernel(const float* __restrict__ est, const float* __restrict__ ref, int64_t nsampl,
           float* __restrict__ out_db) {
    __shared__ double s0[256], s1[256];
    const int sig = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < nsampl; i += 256) {
        const double r = ref[(size_t)sig * nsampl + i], e = est[(size_t)sig * nsampl + i];
        a += r * r;
        b += (r - e) * (r - e);
    }
    s0[threadIdx.x] = a;
    s1[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s0[threadIdx.x] += s0[threadIdx.x + o];
            s1[threadIdx.x] += s1[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out_db[sig] = (float)(10.0 * log10(s0[0] / s1[0]));
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
    DRNMF_LOCK(h);
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
    {
        const int32_t trc = ensure_fft_tables(h, N, logN, (hipStream_t)stream_);
        if (trc) return trc;
    }
    if (stft_fast(N))
        launch_stft_real(N, logN, pcm, is_int16, nsampl, hop, nf, n_sig, mag, nullptr, nullptr,
                         (hipStream_t)stream_);
    else
        hipLaunchKernelGGL(stft_kernel, dim3((unsigned)nf, (unsigned)n_sig), dim3(256), shmem,
                           (hipStream_t)stream_, pcm, is_int16, nsampl, N, logN, hop, nf, mag,
                           (float*)nullptr, (float*)nullptr);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

static int check_fft_size(drnmf_handle_t h, int N, const char* who) {
    if (N < 64 || N > 4096 || (N & (N - 1)))
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "%s: N=%d must be a power of two in [64,4096]", who, N);
    return DRNMF_OK;
}

extern "C" int32_t drnmf_stft(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, int32_t N,
                              int32_t hop, int32_t is_int16, const void* pcm, float* re, float* im,
                              float* mag, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_sig <= 0 || nsampl <= 0 || hop <= 0 || !pcm || !re || !im)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "stft: bad argument");
    int rc = check_fft_size(h, N, "stft");
    if (rc) return rc;
    int logN = 0;
    while ((1 << logN) < N) ++logN;
    const int nf = drnmf_stft_frames(nsampl, N, hop);
    const size_t shmem = (size_t)(N + N / 2) * sizeof(float2);
    {
        const int32_t trc = ensure_fft_tables(h, N, logN, (hipStream_t)stream_);
        if (trc) return trc;
    }
    if (stft_fast(N))
        launch_stft_real(N, logN, pcm, is_int16, nsampl, hop, nf, n_sig, mag, re, im,
                         (hipStream_t)stream_);
    else
        hipLaunchKernelGGL(stft_kernel, dim3((unsigned)nf, (unsigned)n_sig), dim3(256), shmem,
                           (hipStream_t)stream_, pcm, is_int16, nsampl, N, logN, hop, nf, mag, re,
                           im);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" size_t drnmf_istft_workspace_bytes(int32_t n_sig, int32_t n_frames, int32_t N) {
    if (n_sig <= 0 || n_frames <= 0 || N <= 0) return 0;
    return round_up_sz((size_t)n_sig * n_frames * N * sizeof(float), 256);
}

extern "C" int32_t drnmf_istft_masked(drnmf_handle_t h, int32_t n_sig, int32_t n_frames,
                                      int64_t nsampl, int32_t N, int32_t hop, const float* re,
                                      const float* im, const float* mask, float* y,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_sig <= 0 || n_frames <= 0 || nsampl <= 0 || hop <= 0 || !re || !im || !y || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "istft_masked: bad argument");
    int rc = check_fft_size(h, N, "istft_masked");
    if (rc) return rc;
    if (workspace_bytes < drnmf_istft_workspace_bytes(n_sig, n_frames, N))
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "istft_masked: workspace too small");
    int logN = 0;
    while ((1 << logN) < N) ++logN;
    hipStream_t stream = (hipStream_t)stream_;
    float* frames = (float*)workspace;
    const size_t shmem = (size_t)(N + N / 2) * sizeof(float2);
    {
        const int32_t trc = ensure_fft_tables(h, N, logN, stream);
        if (trc) return trc;
    }
    hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)n_frames, (unsigned)n_sig), dim3(256),
                       shmem, stream, re, im, mask, N, logN, hop, n_frames, frames);
    hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)((nsampl + 255) / 256), (unsigned)n_sig),
                       dim3(256), 0, stream, frames, N, hop, n_frames, nsampl, y);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_snr(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, const float* est,
                             const float* ref, float* out_db, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_sig <= 0 || nsampl <= 0 || !est || !ref || !out_db)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "snr: bad argument");
    hipLaunchKernelGGL(snr_kernel, dim3((unsigned)n_sig), dim3(256), 0, (hipStream_t)stream_, est,
                       ref, nsampl, out_db);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}
