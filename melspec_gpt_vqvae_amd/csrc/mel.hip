// wav -> log-mel tile in ONE kernel for gfx950 (reference feature_extraction/extract_mel_spectrogram.py:
// MelSpectrogram.__call__ :36-38 = |librosa.stft(x, 1024, 256)| then mel_basis @ spec; TRANSFORMS :141-151;
// crop + 2x-1 of extract_codes.py:42-43).
//   A 256-thread workgroup takes 16 consecutive frames of one clip; ONE WAVE per STFT frame (4 frames each), no
//   workgroup barrier inside a frame:
//   - the reflect-padded (center=True) frame is gathered straight from the waveform as 512 complex numbers
//     z[m] = x[2m] + i x[2m+1] (8 coalesced 8-byte loads per lane, the wave's NEXT frame requested one frame ahead),
//     periodic-Hann window from registers;
//   - 512-point complex FFT as three radix-8 passes held in registers (8 points per lane), two exchanges through a
//     4 KiB LDS block private to the wave; twiddles live in registers for the life of the (persistent) wave;
//   - real-input unpacking X[k] = E[k] + W^k O[k] for the 513 one-sided bins, |X[k]| to the block's LDS image
//     mag[frame][bin].
//   Then, once per block: mel = basis x mag as exact-f32 MFMAs (v_mfma_f32_16x16x4_f32: 16 filters x 16 frames per
//   accumulator tile).  The triangular Slaney filters are banded, so a 16-filter tile only walks the bins its rows
//   touch (92 k-steps of 4 bins for the reference's 80 x 513 basis instead of 5 x 129); its weights sit in LDS as a
//   dense 16 x width block (zero outside each row's [band_lo, band_hi]).  As VALU work the same filters cost more
//   than the three FFT passes together (measured 4 700 of 7 800 cycles per frame); on the matrix pipe they run under
//   the other waves' butterflies.  The accumulator tile goes through
//   clip((log10(max(min_val, m)) * mult - sub + add) / div, lo, hi) and is written as 64-byte row segments:
//     mel  (n_clips, n_mels, n_keep) f32          - what get_spectrogram saves as *_mel.npy
//     tile (n_clips, n_mels, crop_len) f32/bf16   - 2*mel-1 of columns [crop0, crop0+crop_len): the VQ-VAE input
// HBM traffic per 10 s clip: 882 KB of PCM in, 275 KB + 136..271 KB out; everything else stays on chip.  The bound is
// fp32 VALU, not HBM: ~30 MFLOP per clip of butterflies and unpacking against ~1.2 MB of traffic.
#include "common.h"


namespace {

constexpr int NFFT = 1024, NBINS = 513, NZ = 512;
constexpr int FB = 16, MEL_WAVES = 4, MAXMEL = 256, MAXTILES = MAXMEL / 16;
constexpr int MSTRIDE = 514;  // floats per frame of the magnitude image: = 2 mod 32, so the 16 frames x 4 bins of an
                              // MFMA operand read fall on 64 different banks
constexpr int WCAP = 7168;    // floats of dense filter blocks in LDS (the reference basis needs 6 816)
constexpr size_t MEL_LDS = sizeof(f32x2) * MEL_WAVES * NZ + sizeof(float) * (FB * MSTRIDE + 8 + WCAP) +
                           sizeof(int) * (2 * MAXMEL + 4 + 5 * MAXTILES);

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
  int n_clips, blocks_per_clip;
};

typedef f32x2 cplx;
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {  // two packed ops: a.x * b + (-a.y, a.y) * (b.y, b.x)
  return cplx{a.x, a.x} * b + cplx{-a.y, a.y} * cplx{b.y, b.x};
}
__device__ __forceinline__ cplx mul_mi(cplx a) { return cplx{a.y, -a.x}; }  // a * (-i)
__device__ __forceinline__ cplx cis(float turns) {  // exp(2 pi i turns)
  float s, c;
  sincospif(2.0f * turns, &s, &c);
  return cplx{c, s};
}

// TRANSFORMS[1:] of the reference (LowerThresh, Log10, Multiply, Subtract, Add, Divide, Clip; :141-150) on one value and
// its two destinations: mel[clip][m][f] for f < n_keep (TrimSpec) and the VQ-VAE tile 2 v - 1 of the cropped columns
__device__ __forceinline__ void mel_emit(const MelParams& p, float inv_div, int clip, int m, int f, float acc) {
  float v = fmaxf(p.min_val, acc);  // > 0: v_log_f32 (1 ulp) is enough for log10
  v = ((__builtin_amdgcn_logf(v) * 0.30102999566398120f * p.mult - p.sub) + p.add) * inv_div;
  v = fminf(fmaxf(v, p.clip_lo), p.clip_hi);
  if (p.mel && f < p.n_keep) p.mel[((long long)clip * p.n_mels + m) * p.n_keep + f] = v;
  const int fc = f - p.crop0;
  if (p.tile && fc >= 0 && fc < p.crop_len) {
    const long long o = ((long long)clip * p.n_mels + m) * p.crop_len + fc;
    const float x = 2.0f * v - 1.0f;
    if (p.tile_bf16) ((bf16_t*)p.tile)[o] = f32_to_bf16(x);
    else ((float*)p.tile)[o] = x;
  }
}

// the tail alone on mel magnitudes already in memory, (n_clips, n_mels, n_frames) f32: the same mel_emit the fused kernel
// ends with - the operator form of the reference's TRANSFORMS.transforms[1:] (melgpt_mel_transforms_fwd)
__global__ __launch_bounds__(256) void mel_tail_kernel(MelParams p, const float* mag) {
  const long long total = (long long)p.n_clips * p.n_mels * p.n_frames;
  const float inv_div = 1.0f / p.div;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int f = (int)(e % p.n_frames);
    const long long r = e / p.n_frames;
    mel_emit(p, inv_div, (int)(r / p.n_mels), (int)(r % p.n_mels), f, mag[e]);
  }
}

// forward 8-point DFT in place: a[k] <- sum_n a[n] exp(-2 pi i n k / 8)
__device__ __forceinline__ void dft8(cplx (&a)[8]) {
  const float h = 0.70710678118654752440f;
  const cplx b0 = a[0] + a[4], b1 = a[1] + a[5], b2 = a[2] + a[6], b3 = a[3] + a[7];
  const cplx b4 = a[0] - a[4], d5 = a[1] - a[5], d6 = a[2] - a[6], d7 = a[3] - a[7];
  const cplx b5 = cplx{(d5.x + d5.y) * h, (d5.y - d5.x) * h};    // * exp(-i pi/4)
  const cplx b6 = mul_mi(d6);                                     // * (-i)
  const cplx b7 = cplx{(d7.y - d7.x) * h, -(d7.x + d7.y) * h};   // * exp(-3 i pi/4)
  const cplx e0 = b0 + b2, e1 = b0 - b2, e2 = b1 + b3, e3 = mul_mi(b1 - b3);
  const cplx o0 = b4 + b6, o1 = b4 - b6, o2 = b5 + b7, o3 = mul_mi(b5 - b7);
  a[0] = e0 + e2; a[4] = e0 - e2; a[2] = e1 + e3; a[6] = e1 - e3;
  a[1] = o0 + o2; a[5] = o0 - o2; a[3] = o1 + o3; a[7] = o1 - o3;
}

__global__ __launch_bounds__(64 * MEL_WAVES) void mel_block_kernel(MelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* zbuf = (cplx*)smem;                        // [MEL_WAVES][NZ] per wave: exchange block of the FFT, then Z
  float* magall = (float*)(zbuf + MEL_WAVES * NZ);  // [FB][MSTRIDE] (+8): |X| of the block's frames
  float* wd = magall + FB * MSTRIDE + 8;            // [WCAP] dense filter blocks, tile after tile
  int* slo = (int*)(wd + WCAP);                     // [MAXMEL] first bin of each filter
  int* sn = slo + MAXMEL;                           // [MAXMEL] its width
  int* tklo = sn + MAXMEL;                          // per 16-filter tile: first bin of the union band,
  int* tsteps = tklo + MAXTILES;                    //   k-steps (4 bins each),
  int* tstride = tsteps + MAXTILES;                 //   row stride (= 2 mod 32) and
  int* tbase = tstride + MAXTILES;                  //   offset of its dense block in wd
  int* twave = tbase + MAXTILES;                    //   wave that multiplies it
  int* sflag = twave + MAXTILES;                    // [0] dense blocks fit
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ntiles = (p.n_mels + 15) / 16;

  // ---- once per workgroup: band table, tile plan, dense filter blocks
  for (int m = t; m < MAXMEL; m += 64 * MEL_WAVES) {
    const bool ok = m < p.n_mels;
    const int lo = ok ? max(0, p.band_lo[m]) : 0, hi = ok ? min(NBINS - 1, p.band_hi[m]) : -1;
    slo[m] = lo;
    sn[m] = max(0, hi - lo + 1);
  }
  for (int j = t; j < FB * MSTRIDE + 8; j += 64 * MEL_WAVES) magall[j] = 0.f;  // the pads must stay finite
  __syncthreads();
  if (t < ntiles) {  // one thread per tile: the union of its rows' bands
    int klo = NBINS, khi = 0;
    for (int m = 16 * t; m < min(16 * t + 16, p.n_mels); ++m) {
      const int lo = slo[m], n = sn[m];
      if (n > 0) {
        klo = min(klo, lo);
        khi = max(khi, lo + n);
      }
    }
    if (khi <= klo) klo = khi = 0;
    const int steps = (khi - klo + 3) / 4;
    int stride = 4 * steps;
    stride += (34 - (stride & 31)) & 31;  // smallest value >= 4 steps that is 2 mod 32
    tklo[t] = klo; tsteps[t] = steps; tstride[t] = stride;
  }
  __syncthreads();
  if (t == 0) {
    int base = 0, load[MEL_WAVES] = {0, 0, 0, 0};
    for (int r = ntiles - 1; r >= 0; --r) {  // widest tiles first, each to the least loaded wave
      tbase[r] = base;
      base += 16 * tstride[r];
      int best = 0;
      for (int q = 1; q < MEL_WAVES; ++q)
        if (load[q] < load[best]) best = q;
      twave[r] = best;
      load[best] += tsteps[r] + 2;
    }
    sflag[0] = base <= WCAP;
  }
  __syncthreads();
  const bool dense = sflag[0] != 0;
  if (dense)
    for (int r = 0; r < ntiles; ++r) {
      // thread -> (row ml = t / 16, columns kk = t % 16 + 16 i): 16 consecutive lanes read 64 consecutive bytes of a
      // basis row; four loads in flight per trip (a load -> store loop pays one memory round trip per element)
      const int stride = tstride[r], klo = tklo[r], ml = t >> 4, m = 16 * r + ml;
      const int lo = slo[m], hi = lo + sn[m];
      const float* src = p.basis + (long long)min(m, p.n_mels - 1) * NBINS;
      float* dst = wd + tbase[r] + ml * stride;
      for (int kk0 = t & 15; kk0 < stride; kk0 += 64) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = klo + kk0 + 16 * u;
          v[u] = src[min(k, NBINS - 1)];
          if (!(m < p.n_mels && k >= lo && k < hi)) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (kk0 + 16 * u < stride) dst[kk0 + 16 * u] = v[u];
      }
    }
  // ---- once per wave: window and twiddles of the samples / butterflies this lane owns
  float win[16];  // window at samples 128 n1 + 2 lane, +1
#pragma unroll
  for (int n1 = 0; n1 < 8; ++n1) {
    const int n = 128 * n1 + 2 * lane;
    win[2 * n1] = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)NFFT);  // scipy get_window('hann', 1024, fftbins=True)
    win[2 * n1 + 1] = 0.5f - 0.5f * cospif(2.0f * (float)(n + 1) / (float)NFFT);
  }
  cplx tw1[8], tw2[8], twx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    tw1[k] = cis(-(float)(lane * k) / 512.0f);        // pass 1: lane = n2, output k1 = k: W_512^(n2 k1)
    tw2[k] = cis(-(float)((lane & 7) * k) / 64.0f);   // pass 2: lane = (k1, b), output c = k: W_64^(b c)
    twx[k] = cis(-(float)(lane + 64 * k) / 1024.0f);  // unpacking of bin lane + 64 k: W_1024^bin
  }
  __syncthreads();

  cplx* zb = zbuf + w * NZ;
  const int total_blocks = p.n_clips * p.blocks_per_clip;

  // frame `fi` of block `blk` as 512 raw (unwindowed) complex samples: lane holds z[64 n1 + lane], n1 = 0..7.
  // center=True: frame f covers padded samples [f*hop, f*hop+1024); padded index q is y[q-512] with
  // np.pad(mode='reflect') at both ends.  Interior frames (all but the first and last two or three of a clip) are
  // eight 8-byte loads at constant offsets from one address.
  auto gather = [&](int blk, int fi, cplx (&raw)[8]) -> bool {
    const int clip = blk / p.blocks_per_clip, f = (blk - clip * p.blocks_per_clip) * FB + fi;
    if (blk >= total_blocks || f >= p.n_frames) return false;  // wave-uniform
    const float* y = p.wav + (long long)clip * p.L;
    const long long first = (long long)f * p.hop - NFFT / 2;
    if (first >= 0 && first + NFFT <= p.L) {
      const float* src = y + first + 2 * lane;
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) raw[n1] = cplx{src[128 * n1], src[128 * n1 + 1]};
    } else {
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) {
        long long ja = first + 2 * lane + 128 * n1, jb = ja + 1;
        if (ja < 0) ja = -ja;
        if (ja >= p.L) ja = 2 * (p.L - 1) - ja;
        if (ja < 0) ja = 0;  // only for absurdly short inputs
        if (jb < 0) jb = -jb;
        if (jb >= p.L) jb = 2 * (p.L - 1) - jb;
        if (jb < 0) jb = 0;
        raw[n1] = cplx{y[ja], y[jb]};
      }
    }
    return true;
  };
  // log compression and the two outputs of one value
  const float inv_div = 1.0f / p.div;
  auto emit = [&](int clip, int m, int f, float acc) { mel_emit(p, inv_div, clip, m, f, acc); };

  int blk = blockIdx.x, fi = w;
  cplx nxt[8];
  bool nvalid = gather(blk, fi, nxt);
  while (blk < total_blocks) {
    cplx a[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) a[n1] = cplx{nxt[n1].x * win[2 * n1], nxt[n1].y * win[2 * n1 + 1]};
    const bool valid = nvalid;
    // the wave's next frame is requested now and lands under this frame's butterflies
    int nblk = blk, nfi = fi + MEL_WAVES;
    if (nfi >= FB) {
      nblk = blk + gridDim.x;
      nfi = w;
    }
    nvalid = gather(nblk, nfi, nxt);
    float* mg = magall + fi * MSTRIDE;
    if (valid) {
      // ---- 512-point FFT of z, n = 64 n1 + n2, k = k1 + 8 (c + 8 d)
      dft8(a);  // over n1; lane = n2
#pragma unroll
      for (int k = 0; k < 8; ++k) zb[k * 64 + lane] = k ? cmul(a[k], tw1[k]) : a[k];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = zb[(lane >> 3) * 64 + 8 * q + (lane & 7)];  // lane = (k1, b), over a: n2 = 8 a + b
      __builtin_amdgcn_wave_barrier();
      dft8(a);
#pragma unroll
      for (int c = 0; c < 8; ++c) zb[(lane >> 3) * 64 + c * 8 + (lane & 7)] = c ? cmul(a[c], tw2[c]) : a[c];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = zb[lane * 8 + q];  // lane = (k1, c), over b
      __builtin_amdgcn_wave_barrier();
      dft8(a);
#pragma unroll
      for (int d = 0; d < 8; ++d) zb[(lane >> 3) + 8 * (lane & 7) + 64 * d] = a[d];  // Z[k1 + 8 c + 64 d]
      __builtin_amdgcn_wave_barrier();
      // ---- one-sided spectrum of the real frame: X[k] = E + W_1024^k O,  E = (Z[k] + conj Z[512-k]) / 2,
      //      O = -i (Z[k] - conj Z[512-k]) / 2
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = lane + 64 * q;
        const cplx zk = zb[k], zr = zb[(NZ - k) & (NZ - 1)];
        const cplx e = cplx{0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y)};
        const cplx o = cplx{0.5f * (zk.y + zr.y), -0.5f * (zk.x - zr.x)};
        const cplx x = e + cmul(twx[q], o);
        mg[k] = __builtin_amdgcn_sqrtf(x.x * x.x + x.y * x.y);  // v_sqrt_f32: 1 ulp
        if (k == 0) mg[NZ] = fabsf(e.x - o.x);  // bin 512: W = -1
      }
      __builtin_amdgcn_wave_barrier();
    } else if (blk < total_blocks) {
      for (int k = lane; k < NBINS; k += 64) mg[k] = 0.f;  // a frame past the clip's end: finite operand for the MFMAs
    }
    if (nblk != blk) {  // this wave's last frame of the block (the same iteration for all four waves)
      __syncthreads();
      const int clip = blk / p.blocks_per_clip, f0 = (blk - clip * p.blocks_per_clip) * FB;
      if (dense) {
        // ---- mel[16 filters][16 frames] = W[16][4 s] x mag^T[4 s][16]: lane l feeds A[row l&15][k l>>4],
        // B[k l>>4][col l&15] and owns D[rows 4 (l>>4) + v][col l&15]
        for (int r = 0; r < ntiles; ++r) {
          if (twave[r] != w) continue;  // wave-uniform
          const float* ap = wd + tbase[r] + (lane & 15) * tstride[r] + (lane >> 4);
          const float* bp = magall + (lane & 15) * MSTRIDE + tklo[r] + (lane >> 4);
          const int steps = tsteps[r];
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: a dependent MFMA waits ~2x its issue
          int sidx = 0;
          for (; sidx + 4 <= steps; sidx += 4) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              av[u] = ap[4 * (sidx + u)];
              bv[u] = bp[4 * (sidx + u)];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
          }
          for (; sidx < steps; ++sidx) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * sidx], bp[4 * sidx], acc, 0, 0, 0);
          acc += acc1;
          const int f = f0 + (lane & 15);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int m = 16 * r + 4 * (lane >> 4) + v;
            if (m < p.n_mels && f < p.n_frames) emit(clip, m, f, acc[v]);
          }
        }
      } else {
        // filter banks too wide for the LDS blocks: banded dot products straight from the basis in memory
        for (int e = t; e < p.n_mels * FB; e += 64 * MEL_WAVES) {
          const int m = e / FB, f = f0 + e % FB;
          if (f >= p.n_frames) continue;
          const float* wr = p.basis + (long long)m * NBINS + slo[m];
          const float* mr = magall + (e % FB) * MSTRIDE + slo[m];
          float acc = 0.f;
          for (int j = 0; j < sn[m]; ++j) acc = fmaf(wr[j], mr[j], acc);
          emit(clip, m, f, acc);
        }
      }
      __syncthreads();
    }
    blk = nblk;
    fi = nfi;
  }
}

}  // namespace

extern "C" int melgpt_mel_transforms_fwd(const float* mel_in, int n_clips, int n_mels, int n_frames, float min_val,
                                         float mult, float sub, float add, float div, float clip_lo, float clip_hi,
                                         float* mel_out, int n_keep, void* tile_out, int tile_dtype, int crop0,
                                         int crop_len, void* stream) {
  MELGPT_CHECK(mel_in && n_clips > 0 && n_mels > 0 && n_frames > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(mel_out || tile_out, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(!tile_out || tile_dtype == MELGPT_F32 || tile_dtype == MELGPT_BF16, MELGPT_ERR_UNSUPPORTED);
  MELGPT_CHECK(n_keep >= 0 && crop0 >= 0 && (!mel_out || n_keep <= n_frames) && (!tile_out || crop0 + crop_len <= n_frames),
               MELGPT_ERR_BAD_ARG);
  MelParams p{};
  p.n_frames = n_frames; p.n_mels = n_mels; p.n_clips = n_clips;
  p.min_val = min_val; p.mult = mult; p.sub = sub; p.add = add; p.div = div; p.clip_lo = clip_lo; p.clip_hi = clip_hi;
  p.mel = mel_out; p.n_keep = n_keep; p.tile = tile_out; p.tile_bf16 = tile_dtype == MELGPT_BF16;
  p.crop0 = crop0; p.crop_len = crop_len;
  const long long total = (long long)n_clips * n_mels * n_frames;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(mel_tail_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, mel_in);
  return melgpt_launch_status();
}

extern "C" int melgpt_mel_frontend_fwd(const float* wav, int n_clips, long long n_samples, int n_fft, int hop,
                                       const float* mel_basis, const int* band_lo, const int* band_hi, int n_mels,
                                       float min_val, float mult, float sub, float add, float div, float clip_lo,
                                       float clip_hi, float* mel_out, int n_keep, void* tile_out, int tile_dtype,
                                       int crop0, int crop_len, void* stream) {
  MELGPT_CHECK(wav && mel_basis && band_lo && band_hi && n_clips > 0 && n_samples > 1 && hop > 0, MELGPT_ERR_BAD_ARG);
  MELGPT_CHECK(n_fft == NFFT && n_mels > 0 && n_mels <= MAXMEL, MELGPT_ERR_UNSUPPORTED);
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
  p.n_clips = n_clips;
  p.blocks_per_clip = (need + FB - 1) / FB;
  static int slots = 0;  // workgroups the device holds at once: the persistent grid
  if (!slots) {
    int dev = 0, ncu = 0, per_cu = 0;
    if (hipFuncSetAttribute((const void*)mel_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MEL_LDS) != hipSuccess)
      return MELGPT_ERR_LAUNCH;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      ncu = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mel_block_kernel, 64 * MEL_WAVES, MEL_LDS) != hipSuccess || per_cu < 1)
      per_cu = 1;
    slots = ncu * per_cu;
  }
  const long long total = (long long)n_clips * p.blocks_per_clip;
  MELGPT_CHECK(total < 0x7FFFFFFF, MELGPT_ERR_UNSUPPORTED);
  const int grid = (int)(total < slots ? total : slots);
  hipLaunchKernelGGL(mel_block_kernel, dim3(grid), dim3(64 * MEL_WAVES), MEL_LDS, (hipStream_t)stream, p);
  return melgpt_launch_status();
}
