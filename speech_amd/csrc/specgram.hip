// specgram.hip -- on-device log power spectrogram: the featuriser of /root/reference/speech/loader.py:156-166
// (scipy.signal.spectrogram(audio, fs, window='hann', nperseg, noverlap, detrend=False) -> log(spec.T + eps)) and the
// per-bin normalisation of loader.py:65-69.  SURVEY.md 8(f) rank 4: the host-side feature computation becomes the
// bottleneck once the model step is fast (train.py's data_time).
//
// A spectrogram is frames x DFT: frame i = audio[i*hop : i*hop + nperseg], i.e. the audio buffer viewed as a matrix
// with row stride `hop` (rows overlap) -- the fp32 MFMA GEMM takes that view directly (lda = hop), no framing copy.
// The (nperseg x 2*nbins) matrix holds the periodic Hann window folded into cos / -sin columns, built once per
// nperseg in double precision.  An epilogue kernel forms |X|^2, applies scipy's one-sided 'density' scaling,
// adds eps, takes the log and (optionally) normalises.
#include "common.h"
#include "internal.h"

namespace {

__global__ __launch_bounds__(256) void dft_table_kernel(float* __restrict__ tab, int N, int nbins) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * nbins) return;
    const int k = idx / nbins, j = idx % nbins;
    const double w = 0.5 - 0.5 * cospi(2.0 * (double)k / (double)N);  // periodic Hann (scipy get_window('hann', N))
    const long jk = ((long)j * k) % N;                               // exact argument reduction
    const double ang = 2.0 * (double)jk / (double)N;                 // in units of pi
    tab[(long)k * 2 * nbins + 2 * j] = (float)(w * cospi(ang));
    tab[(long)k * 2 * nbins + 2 * j + 1] = (float)(-w * sinpi(ang));
}

__global__ __launch_bounds__(256) void i16_to_f32_kernel(const short* __restrict__ a, float* __restrict__ o, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) o[i] = (float)a[i];
}

__global__ __launch_bounds__(256) void specgram_epilogue_kernel(const float* __restrict__ ri, int frames, int nbins,
                                                                int N, float scale, float eps,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ std, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)frames * nbins) return;
    const int j = (int)(idx % nbins);
    const float re = ri[2 * idx], im = ri[2 * idx + 1];
    float p = (re * re + im * im) * scale;
    const bool edge = j == 0 || ((N & 1) == 0 && j == nbins - 1);  // DC and (even N) Nyquist are not doubled
    if (!edge) p *= 2.0f;
    float v = logf(p + eps);
    if (mean) v = (v - mean[j]) / std[j];
    out[idx] = v;
}

}  // namespace

extern "C" int sa_specgram_frames(int n, int nperseg, int hop) {
    if (n < nperseg || nperseg <= 0 || hop <= 0) return 0;
    return 1 + (n - nperseg) / hop;
}

extern "C" ctcStatus_t sa_specgram_build_dft(float* d_dft, int nperseg, void* stream_) {
    SA_CLEAR_ERR();
    if (!d_dft || nperseg <= 0) return CTC_STATUS_INVALID_VALUE;
    const int nbins = nperseg / 2 + 1;
    hipLaunchKernelGGL(dft_table_kernel, dim3((nperseg * nbins + 255) / 256), dim3(256), 0, (hipStream_t)stream_,
                       d_dft, nperseg, nbins);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}

extern "C" size_t sa_log_specgram_workspace_bytes(int n, int nperseg, int hop) {
    const int frames = sa_specgram_frames(n, nperseg, hop);
    if (frames <= 0) return 0;
    const int nbins = nperseg / 2 + 1;
    return sa_align_up((size_t)n * sizeof(float), 256) + sa_align_up((size_t)frames * 2 * nbins * sizeof(float), 256);
}

extern "C" ctcStatus_t sa_log_specgram(const short* d_audio, int n, int sample_rate, int nperseg, int hop,
                                       const float* d_dft, const float* d_mean, const float* d_std, float eps,
                                       float* out, void* workspace, size_t workspace_bytes, void* stream_) {
    SA_CLEAR_ERR();
    if (!d_audio || !d_dft || !out || !workspace || sample_rate <= 0 || (d_mean != nullptr) != (d_std != nullptr))
        return CTC_STATUS_INVALID_VALUE;
    const int frames = sa_specgram_frames(n, nperseg, hop);
    if (frames <= 0 || workspace_bytes < sa_log_specgram_workspace_bytes(n, nperseg, hop))
        return CTC_STATUS_INVALID_VALUE;
    hipStream_t stream = (hipStream_t)stream_;
    const int nbins = nperseg / 2 + 1;
    float* af = (float*)workspace;
    float* ri = (float*)((char*)workspace + sa_align_up((size_t)n * sizeof(float), 256));
    int g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(i16_to_f32_kernel, dim3(g), dim3(256), 0, stream, d_audio, af, (long)n);
    SA_CHECK_LAUNCH();
    // frames (rows overlap: lda = hop) x windowed DFT
    // (the f32-input MFMA kernel whatever the batch size: no packed operands, no workspace)
    SaGemmOpts gopts{};
    gopts.no_split = 1; gopts.exact = 1;
    const float* Ap = af; const float* Bp = d_dft; float* Cp = ri; const float* biasp = nullptr;
    ctcStatus_t st = sa_gemm_f32_group_impl(1, 0, 0, frames, 2 * nbins, nperseg, 1.0f, &Ap, hop, &Bp, 2 * nbins, 0.f, &Cp,
                                            2 * nbins, &biasp, nullptr, nullptr, 0, stream, &gopts);
    if (st != CTC_STATUS_SUCCESS) return st;
    // scipy 'density' scaling: 1 / (fs * sum(w^2)); a periodic Hann window of length N has sum(w^2) = 3N/8
    const float scale = (float)(1.0 / ((double)sample_rate * 0.375 * (double)nperseg));
    const long total = (long)frames * nbins;
    hipLaunchKernelGGL(specgram_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ri, frames,
                       nbins, nperseg, scale, eps, d_mean, d_std, out);
    SA_CHECK_LAUNCH();
    return CTC_STATUS_SUCCESS;
}
