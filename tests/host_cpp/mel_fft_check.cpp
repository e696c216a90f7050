// Host emulation of the wave-level FFT of speecht_amd/csrc/mel_fft.h: 64 "lanes" executed phase by phase
// (every lane reads, then every lane writes -- what a wavefront does), checked against a direct DFT.
// Built and run by tests/test_mel_fft_host.py with g++; prints the maximum errors.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mel_fft.h"

using melfft::cf;

int main() {
  const int N = 512;
  std::vector<double> a(N), b(N);
  srand(7);
  for (int i = 0; i < N; ++i) { a[i] = rand() / (double)RAND_MAX - 0.5; b[i] = rand() / (double)RAND_MAX - 0.5; }
  // reference power spectra of the two real frames
  std::vector<double> pa(257), pb(257);
  for (int k = 0; k <= 256; ++k) {
    double ar = 0, ai = 0, br = 0, bi = 0;
    for (int n = 0; n < N; ++n) {
      const double c = cos(-2 * M_PI * k * n / N), s = sin(-2 * M_PI * k * n / N);
      ar += a[n] * c; ai += a[n] * s; br += b[n] * c; bi += b[n] * s;
    }
    pa[k] = ar * ar + ai * ai; pb[k] = br * br + bi * bi;
  }
  std::vector<cf> lds(melfft::LDS_COMPLEX);
  cf v[64][8];
  // stage 0: straight from the input, no twiddles
  for (int j = 0; j < 64; ++j) {
    for (int r = 0; r < 8; ++r) v[j][r] = cf{(float)a[j + 64 * r], (float)b[j + 64 * r]};
    melfft::dft8(v[j]);
  }
  for (int j = 0; j < 64; ++j) for (int r = 0; r < 8; ++r) lds[melfft::pad(melfft::out_index(0, j, r))] = v[j][r];
  // stage 1: Ns = 8
  for (int j = 0; j < 64; ++j) {
    for (int r = 0; r < 8; ++r) {
      const double ang = -2 * M_PI * (j & 7) * r / 64.0;
      v[j][r] = melfft::zmul(lds[melfft::pad(j + 64 * r)], cf{(float)cos(ang), (float)sin(ang)});
    }
    melfft::dft8(v[j]);
  }
  for (int j = 0; j < 64; ++j) for (int r = 0; r < 8; ++r) lds[melfft::pad(melfft::out_index(1, j, r))] = v[j][r];
  // stage 2: Ns = 64, result stays in registers: Z[j + 64 r] = v[j][r]
  for (int j = 0; j < 64; ++j) {
    for (int r = 0; r < 8; ++r) {
      const double ang = -2 * M_PI * j * r / 512.0;
      v[j][r] = melfft::zmul(lds[melfft::pad(j + 64 * r)], cf{(float)cos(ang), (float)sin(ang)});
    }
    melfft::dft8(v[j]);
  }
  // power of both frames: partner of (lane j, reg r) is lane (64 - j) & 63, reg 7 - r (8 - r for lane 0)
  double ea = 0, eb = 0, scale = 0;
  for (int j = 0; j < 64; ++j) {
    const int pl = (64 - j) & 63;
    for (int r = 0; r < (j == 0 ? 5 : 4); ++r) {
      const int k = j + 64 * r;
      const int pr = j == 0 ? (r == 0 ? 0 : 8 - r) : 7 - r;
      float qa, qb;
      melfft::pair_power(v[j][r], v[pl][pr], &qa, &qb);
      ea = fmax(ea, fabs(qa - pa[k])); eb = fmax(eb, fabs(qb - pb[k]));
      scale = fmax(scale, fmax(pa[k], pb[k]));
    }
  }
  printf("max_err_a %.3e max_err_b %.3e scale %.3e\n", ea, eb, scale);
  return (ea < 2e-5 * scale && eb < 2e-5 * scale) ? 0 : 1;
}
