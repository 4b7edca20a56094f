// wav -> log-mel tile in ONE kernel for gfx950 (reference feature_extraction/extract_mel_spectrogram.py:
// MelSpectrogram.__call__ :36-38 = |librosa.stft(x, 1024, 256)| then mel_basis @ spec; TRANSFORMS :141-151;
// crop + 2x-1 of extract_codes.py:42-43).  One 256-thread workgroup per STFT frame:
//   reflect-padded (center=True) frame gather straight from the waveform, periodic-Hann window,
//   1024-point FFT in LDS (radix-2, 10 stages, twiddles generated once per workgroup with sincospi),
//   |X[k]| for the 513 one-sided bins, triangular Slaney mel filters applied as banded dot products
//   (each of the 80 rows of mel_basis is non-zero only on [lo, hi]), then
//   clip((log10(max(min_val, m)) * mult - sub + add) / div, lo, hi) and the two outputs:
//     mel  (n_clips, n_mels, n_keep) f32          - what get_spectrogram saves as *_mel.npy
//     tile (n_clips, n_mels, crop_len) f32/bf16   - 2*mel-1 of columns [crop0, crop0+crop_len): the VQ-VAE input
// HBM traffic per 10 s clip: 882 KB of PCM in, 275 KB + 136..271 KB out; everything else stays on chip.
#include "common.h"

namespace {

constexpr int NFFT = 1024, NBINS = 513, LOG2N = 10;

struct MelParams {
  const float* wav;  // (n_clips, L)
  long long L;
  int hop, n_frames;  // frames computed: 0 .. n_frames-1
  const float* basis;  // (n_mels, NBINS) f32
  const int* band_lo;  // (n_mels,) first / last non-zero bin of each filter
  const int* band_hi;
  int n_mels;
  float min_val, mult, sub, add, div, clip_lo, clip_hi;
  float* mel;  // optional
  int n_keep;
  void* tile;  // optional
  int tile_bf16, crop0, crop_len;
};

__global__ __launch_bounds__(256) void mel_frame_kernel(MelParams p) {
  __shared__ float re[NFFT], im[NFFT], twr[NFFT / 2], twi[NFFT / 2], mag[NBINS + 3];
  const int t = threadIdx.x;
  const int f = blockIdx.x, clip = blockIdx.y;
  const float* y = p.wav + (long long)clip * p.L;

  // twiddles W^j = exp(-2 pi i j / 1024), j < 512
  for (int j = t; j < NFFT / 2; j += 256) {
    float s, c;
    sincospif(-2.0f * (float)j / (float)NFFT, &s, &c);
    twr[j] = c;
    twi[j] = s;
  }
  // windowed frame, bit-reversed into place.  center=True: frame f covers padded samples [f*hop, f*hop+1024),
  // padded index q maps to y[q-512] with np.pad(mode='reflect') at both ends
  for (int n = t; n < NFFT; n += 256) {
    long long j = (long long)f * p.hop + n - NFFT / 2;
    if (j < 0) j = -j;
    if (j >= p.L) j = 2 * (p.L - 1) - j;
    if (j < 0) j = 0;  // only for absurdly short inputs
    float c = cospif(2.0f * (float)n / (float)NFFT);
    float w = 0.5f - 0.5f * c;  // scipy.signal.get_window('hann', 1024, fftbins=True)
    unsigned r = __brev((unsigned)n) >> (32 - LOG2N);
    re[r] = y[j] * w;
    im[r] = 0.f;
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < LOG2N; ++s) {
    const int half = 1 << s;
    for (int b = t; b < NFFT / 2; b += 256) {
      const int pos = b & (half - 1);
      const int i0 = ((b >> s) << (s + 1)) + pos, i1 = i0 + half;
      const int tw = pos << (LOG2N - 1 - s);
      const float wr = twr[tw], wi = twi[tw];
      const float br = re[i1] * wr - im[i1] * wi, bi = re[i1] * wi + im[i1] * wr;
      const float ar = re[i0], ai = im[i0];
      re[i0] = ar + br; im[i0] = ai + bi;
      re[i1] = ar - br; im[i1] = ai - bi;
    }
    __syncthreads();
  }
  for (int k = t; k < NBINS; k += 256) mag[k] = sqrtf(re[k] * re[k] + im[k] * im[k]);
  __syncthreads();
  if (t < p.n_mels) {
    const float* w = p.basis + (long long)t * NBINS;
    float acc = 0.f;
    for (int k = p.band_lo[t]; k <= p.band_hi[t]; ++k) acc = fmaf(w[k], mag[k], acc);
    float v = fmaxf(p.min_val, acc);
    v = ((log10f(v) * p.mult - p.sub) + p.add) / p.div;
    v = fminf(fmaxf(v, p.clip_lo), p.clip_hi);
    if (p.mel && f < p.n_keep) p.mel[((long long)clip * p.n_mels + t) * p.n_keep + f] = v;
    const int fc = f - p.crop0;
    if (p.tile && fc >= 0 && fc < p.crop_len) {
      const long long o = ((long long)clip * p.n_mels + t) * p.crop_len + fc;
      const float x = 2.0f * v - 1.0f;
      if (p.tile_bf16) ((bf16_t*)p.tile)[o] = f32_to_bf16(x);
      else ((float*)p.tile)[o] = x;
    }
  }
}

}  // namespace

extern "C" int melgpt_mel_frontend_fwd(const float* wav, int n_clips, long long n_samples, int n_fft, int hop,
                                       const float* mel_basis, const int* band_lo, const int* band_hi, int n_mels,
                                       float min_val, float mult, float sub, float add, float div, float clip_lo,
                                       float clip_hi, float* mel_out, int n_keep, void* tile_out, int tile_dtype,
                                       int crop0, int crop_len, void* stream) {
  MELGPT_CHECK(wav && mel_basis && band_lo && band_hi && n_clips > 0 && n_samples > 1 && hop > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(n_fft == NFFT && n_mels > 0 && n_mels <= 256, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(mel_out || tile_out, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(!tile_out || tile_dtype == MELGPT_F32 || tile_dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  const long long total_frames = 1 + n_samples / hop;  // librosa: 1 + len(y)//hop with center=True
  int need = 0;
  if (mel_out) need = n_keep;
  if (tile_out && crop0 + crop_len > need) need = crop0 + crop_len;
  MELGPT_CHECK(need > 0 && need <= total_frames && n_keep >= 0 && crop0 >= 0, MELGPT_ERR_BAD_ARG);
  MelParams p{};
  p.wav = wav; p.L = n_samples; p.hop = hop; p.n_frames = need;
  p.basis = mel_basis; p.band_lo = band_lo; p.band_hi = band_hi; p.n_mels = n_mels;
  p.min_val = min_val; p.mult = mult; p.sub = sub; p.add = add; p.div = div; p.clip_lo = clip_lo; p.clip_hi = clip_hi;
  p.mel = mel_out; p.n_keep = n_keep; p.tile = tile_out; p.tile_bf16 = tile_dtype == MELGPT_BF16;
  p.crop0 = crop0; p.crop_len = crop_len;
  hipLaunchKernelGGL(mel_frame_kernel, dim3(need, n_clips), dim3(256), 0, (hipStream_t)stream, p);
  return melgpt_launch_status();
}
