"""The three-product (Gauss) form of the wide frequency-domain layer's per-bin complex products (csrc/conv_gemm.hip
gemm_nn_g3_kernel / gemm_tn_g3_kernel, csrc/conv_fft.hip g3_form): the kernels against float64 products of the same planes, and
the layer entry points (speech_model.py:155,173,177,78 -- tf.nn.conv1d 'SAME' + bias + relu and its gradients) in both forms
against the float64 oracle at the frequency-domain tests' tolerance (2e-5 of the tensor's max)."""
import ctypes

import numpy as np
import pytest

from oracle import w2l_oracle as O

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  return torch.device('cuda:0')


def P(t):
  return ctypes.c_void_p(t.data_ptr())


def i64x3(v):
  return (ctypes.c_int64 * 3)(*[int(x) for x in v])


@pytest.mark.parametrize('bins,M,K,N,bt', [(9, 256, 128, 256, False), (9, 256, 256, 128, True), (3, 192, 64, 128, False),
                                           (10, 64, 128, 256, True), (2, 100, 64, 128, False), (17, 320, 64, 128, True)])
def test_gemm_nn_g3_against_float64(dev, bins, M, K, N, bt):
  # C[:, :N] = A0 B0 + A1 B1, C[:, c_off2:] = A0 B0 + A2 B2; planes at arbitrary (16-byte aligned) offsets; M = 100: a partial
  # last tile; 192 / 320 rows: whole 128-row tiles + a launch of the last 64; more than 8 bins: a second set on the XCDs
  from speecht_amd._lib import call
  rng = np.random.default_rng(bins * 1000 + M + K)
  lda = 3 * K + 8
  A = rng.standard_normal((bins, M, lda)).astype(np.float32)
  a_off = [2 * K + 4, 0, K + 4]
  if bt:
    ldb = K + 4
    Bm = rng.standard_normal((bins, 3, N, ldb)).astype(np.float32)          # planes hold B_p^T
    b_off = [2 * N * ldb, N * ldb, 0]
    planes = [Bm[:, 2, :, :K].transpose(0, 2, 1), Bm[:, 1, :, :K].transpose(0, 2, 1), Bm[:, 0, :, :K].transpose(0, 2, 1)]
  else:
    ldb = N + 8
    Bm = rng.standard_normal((bins, 3, K, ldb)).astype(np.float32)
    b_off = [K * ldb, 0, 2 * K * ldb + 4]
    planes = [Bm[:, 1, :, :N], Bm[:, 0, :, :N], Bm[:, 2, :, 4:4 + N]]
  ldc, c_off2 = 2 * N + 12, N + 8
  Ad, Bd = torch.as_tensor(A).to(dev), torch.as_tensor(Bm).to(dev)
  Cd = torch.full((bins, M, ldc), 7.0, device=dev)
  call('st_gemm_nn_g3_batched_f32', P(Ad), lda, M * lda, i64x3(a_off), P(Bd), ldb, Bm[0].size, i64x3(b_off), int(bt), P(Cd), ldc, M * ldc,
       c_off2, M, K, N, bins, None)
  C = Cd.cpu().numpy()
  Ap = [A[:, :, o:o + K].astype(np.float64) for o in a_off]
  Bp = [p.astype(np.float64) for p in planes]
  k1 = Ap[0] @ Bp[0]
  re, im = k1 + Ap[1] @ Bp[1], k1 + Ap[2] @ Bp[2]
  scale = max(np.abs(re).max(), np.abs(im).max())
  assert np.abs(C[:, :, :N] - re).max() < 2e-6 * scale
  assert np.abs(C[:, :, c_off2:c_off2 + N] - im).max() < 2e-6 * scale
  # nothing outside the two column ranges is written
  assert float(np.abs(C[:, :, N:c_off2] - 7.0).max()) == 0.0 and float(np.abs(C[:, :, c_off2 + N:] - 7.0).max()) == 0.0


def test_split_reduction_adds_two_halves_and_is_bit_reproducible(dev):
  # the 32-tap layer's back-prop shape in small: few output tiles, a long reduction -> 64-row tiles, each phase's reduction in two
  # halves ADDED into a zeroed output by float atomics.  Two addends onto +0: a + b == b + a, so whichever workgroup arrives first
  # the bits are the same -- twenty launches over a dirty output, all identical, and equal to the unsplit launch to rounding
  from speecht_amd._lib import call, launch_trace, set_tuning
  bins, M, K, N = 48, 256, 512, 256
  rng = np.random.default_rng(7)
  A = torch.as_tensor(rng.standard_normal((bins, M, 3 * K)).astype(np.float32)).to(dev)
  Bt = torch.as_tensor(rng.standard_normal((bins, 3, N, K)).astype(np.float32)).to(dev)
  C = torch.empty((bins, M, 2 * N), device=dev)

  def run():
    C.fill_(float('nan'))
    with launch_trace() as tr:
      call('st_gemm_nn_g3_batched_f32', P(A), 3 * K, M * 3 * K, i64x3([0, 2 * K, K]), P(Bt), K, 3 * N * K, i64x3([0, 2 * N * K, N * K]), 1,
           P(C), 2 * N, M * 2 * N, N, M, K, N, bins, None)
    torch.cuda.synchronize()
    return C.clone(), tr.lines[0]

  first, line = run()
  assert 'gemm_nn_g3<64,bt>' in line and 'ksplit=2' in line, line
  assert bool(torch.isfinite(first).all())
  for _ in range(20):
    again, _ = run()
    assert torch.equal(first, again)
  try:
    set_tuning('g3_tile', 2)                       # 128-row tiles, the whole reduction in one workgroup
    whole, line = run()
    assert 'ksplit=1' in line, line
  finally:
    set_tuning('g3_tile', 0)
  assert float((first - whole).abs().max()) < 2e-6 * float(whole.abs().max())


@pytest.mark.parametrize('bins,M,K,N', [(9, 256, 128, 256), (3, 64, 256, 128), (17, 96, 128, 128)])
def test_gemm_tn_g3_against_float64(dev, bins, M, K, N):
  # out[bin][0] = A0^T Z0 + A1^T Z1, out[bin][1] = A2^T Z2 - A0^T Z0
  from speecht_amd._lib import call
  rng = np.random.default_rng(bins * 1000 + M + K + 1)
  lda, ldz = 4 * K, 3 * N
  A = rng.standard_normal((bins, M, lda)).astype(np.float32)
  Z = rng.standard_normal((bins, M, ldz)).astype(np.float32)
  a_off, z_off = [2 * K, 3 * K, 0], [0, 2 * N, N]
  o_part = K * N + 64
  out = torch.full((bins, 2 * o_part), 7.0, device=dev)
  Ad, Zd = torch.as_tensor(A).to(dev), torch.as_tensor(Z).to(dev)
  call('st_gemm_tn_g3_batched_f32', P(Ad), lda, M * lda, i64x3(a_off), P(Zd), ldz, M * ldz, i64x3(z_off), P(out), 2 * o_part, o_part, M, K, N,
       bins, None)
  o = out.cpu().numpy()
  Ap = [A[:, :, a:a + K].astype(np.float64) for a in a_off]
  Zp = [Z[:, :, z:z + N].astype(np.float64) for z in z_off]
  k1 = np.einsum('bmk,bmn->bkn', Ap[0], Zp[0])
  re = k1 + np.einsum('bmk,bmn->bkn', Ap[1], Zp[1])
  im = np.einsum('bmk,bmn->bkn', Ap[2], Zp[2]) - k1
  scale = max(np.abs(re).max(), np.abs(im).max())
  assert np.abs(o[:, :K * N].reshape(bins, K, N) - re).max() < 2e-6 * scale
  assert np.abs(o[:, o_part:o_part + K * N].reshape(bins, K, N) - im).max() < 2e-6 * scale
  assert float(np.abs(o[:, K * N:o_part] - 7.0).max()) == 0.0


# (W, B, T, cin, cout): 64 rows per bin (the 64-row kernel), 128 rows, 192 rows (128 + a launch of the last 64); 12 taps over an
# input of 120 channels (spectra halves of 128 columns), 32 taps over 250 (256); the last: an input whose spectra do NOT tile the
# kernels (130 channels: halves of 192 columns) -- its gradient spectra keep three-part rows, read by the four-product kernels
@pytest.mark.parametrize('W,B,T,cin,cout,relu,form', [(32, 3, 300, 250, 600, True, 2), (32, 5, 1000, 250, 520, True, 2),
                                                      (12, 3, 4096, 120, 520, False, 2), (12, 2, 100, 130, 600, True, 1)])
def test_three_product_layer_matches_oracle_in_both_forms(dev, W, B, T, cin, cout, relu, form):
  from speecht_amd import _lib
  from speecht_amd._lib import launch_trace, set_tuning
  from speecht_amd.engine import channel_pitch
  from tests.test_gpu_fft_conv import _fft_conv_matches_oracle
  lib = _lib.load()
  try:
    for no_g3 in (0, 1):
      set_tuning('no_g3', no_g3)
      assert lib.st_conv1d_fft_three_products(W, channel_pitch(cin), cout) == (0 if no_g3 else form)
      with launch_trace() as tr:
        _fft_conv_matches_oracle(dev, W, B, T, cin, cout, relu)
      g3 = [l for l in tr.lines if l.startswith('gemm_nn_g3') or l.startswith('gemm_tn_g3')]
      # forward + back-prop to the input (+ their 64-row launches at 192 rows) and the lag products, or none of them
      assert (len(g3) >= 3) == (form == 2 and not no_g3), tr.lines
  finally:
    set_tuning('no_g3', 0)
