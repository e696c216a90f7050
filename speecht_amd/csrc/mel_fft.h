// 512-point complex FFT of one wave (64 lanes x 8 points) as three radix-8 Stockham stages, used by
// melspec.hip to transform TWO real frames at once (frame A in the real part, frame B in the imaginary part).
//
// Stockham autosort, radix R = 8, N = 512 = 8^3, one butterfly per lane per stage (lane j of 64):
//   stage with sub-transform size Ns (1, 8, 64):
//     v[r]  = in[j + 64 r] * exp(-2 pi i (j mod Ns) r / (8 Ns))        r = 0..7
//     v     = DFT8(v)
//     out[(j / Ns) * 8 Ns + (j mod Ns) + r Ns] = v[r]
//   natural order in, natural order out; the first stage has no twiddles and reads straight from the audio,
//   the last stage leaves Z[j + 64 r] in lane j's registers.
// Two real frames from one complex transform: with Z = FFT(a + i b),
//   |A[k]|^2 = |Z[k] + conj(Z[N-k])|^2 / 4,   |B[k]|^2 = |Z[k] - conj(Z[N-k])|^2 / 4,
// and Z[N-k] for k = j + 64 r lives in lane (64 - j) mod 64, register 7 - r (register 8 - r for lane 0).
//
// The header is plain C++ so that tests/test_mel_fft_host.py can compile the same butterflies with g++ and
// check the index algebra against a direct DFT without a GPU.
#pragma once

#ifdef __HIPCC__
#define ST_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define ST_HD inline
#endif

namespace melfft {

struct cf {
  float x, y;
};

ST_HD cf zadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
ST_HD cf zsub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
ST_HD cf zmul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
ST_HD cf mul_neg_i(cf a) { return cf{a.y, -a.x}; }                       // a * (-i)

// in-place 8-point DFT (forward, e^{-2 pi i nk/8}), natural order in and out
ST_HD void dft8(cf* v) {
  const float h = 0.70710678118654752f;
  // radix-2 split (decimation in frequency): even outputs from b0..b3, odd outputs from b4..b7
  cf b0 = zadd(v[0], v[4]), b1 = zadd(v[1], v[5]), b2 = zadd(v[2], v[6]), b3 = zadd(v[3], v[7]);
  cf d4 = zsub(v[0], v[4]), d5 = zsub(v[1], v[5]), d6 = zsub(v[2], v[6]), d7 = zsub(v[3], v[7]);
  cf b4 = d4;
  cf b5 = cf{(d5.x + d5.y) * h, (d5.y - d5.x) * h};                       // * (1 - i) / sqrt 2
  cf b6 = mul_neg_i(d6);
  cf b7 = cf{(d7.y - d7.x) * h, -(d7.x + d7.y) * h};                      // * (-1 - i) / sqrt 2
  // 4-point DFTs
  cf e0 = zadd(b0, b2), e1 = zadd(b1, b3), e2 = zsub(b0, b2), e3 = mul_neg_i(zsub(b1, b3));
  cf o0 = zadd(b4, b6), o1 = zadd(b5, b7), o2 = zsub(b4, b6), o3 = mul_neg_i(zsub(b5, b7));
  v[0] = zadd(e0, e1); v[4] = zsub(e0, e1); v[2] = zadd(e2, e3); v[6] = zsub(e2, e3);
  v[1] = zadd(o0, o1); v[5] = zsub(o0, o1); v[3] = zadd(o2, o3); v[7] = zsub(o2, o3);
}

// LDS index of complex element i: one pad element every 8 keeps the stride-8 writes of stage 1 and the stride-1
// reads conflict-free
ST_HD int pad(int i) { return i + (i >> 3); }
constexpr int LDS_COMPLEX = 512 + 64;

// where lane j writes its r-th output of stage `stage` (0, 1): see the header comment
ST_HD int out_index(int stage, int j, int r) {
  return stage == 0 ? 8 * j + r : ((j >> 3) * 64 + (j & 7) + 8 * r);
}

// power spectra of the two packed frames at bin k from Z[k] = a and Z[N-k] = b
ST_HD void pair_power(cf a, cf b, float* pa, float* pb) {
  const float sx = a.x + b.x, dy = a.y - b.y, dx = a.x - b.x, sy = a.y + b.y;
  *pa = 0.25f * (sx * sx + dy * dy);
  *pb = 0.25f * (dx * dx + sy * sy);
}

}  // namespace melfft
