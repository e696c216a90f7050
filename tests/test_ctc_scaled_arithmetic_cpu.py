"""The arithmetic of the device's CTC recursion (csrc/ctc.hip since round 3), restated in numpy float32 and held against
the float64 log-space oracle on the CPU.

A lattice value is m * 2^e with an int exponent PER STATE; a step aligns the three predecessors to their largest exponent
(ldexp), adds, multiplies by the emission's mantissa in [1, 2) and re-normalises with frexp.  What this file pins down is the
claim the kernel's design rests on: the form has the range of the log domain -- no state is ever lost, however far
below its column's maximum it lies -- with a float's relative precision, where a COLUMN-wide scale (the variant of round 2) was
3.7 nats off on a 1 377-frame utterance.  The kernel itself is tested against the same oracle under -m gpu
(tests/test_gpu_parity.py); this model is not used by it or by any product path."""
import numpy as np
import pytest

from oracle import w2l_oracle as O

EZ = -(1 << 28)


def scaled_alpha_loss(logits, label, column_scale=False):
  """-log p(label | logits [T, C]) by the forward recursion in float32 scaled arithmetic.  column_scale=True: ONE exponent
  per frame (the column maximum's) instead of one per state -- the variant that loses the paths that finish."""
  T, C = logits.shape
  blank = C - 1
  L = len(label)
  U = 2 * L + 1
  ext = np.full(U, blank)
  ext[1::2] = label
  skip = np.zeros(U, dtype=bool)
  skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
  l2 = (O.log_softmax(logits.astype(np.float64)) / np.log(2.0)).astype(np.float32)       # log2-softmax as the device stores it
  fl = np.floor(l2)
  em_m, em_e = np.exp2(l2 - fl).astype(np.float32), fl.astype(np.int64)                  # 2^l2 = em_m * 2^em_e, em_m in [1, 2)
  m = np.zeros(U, dtype=np.float32)
  e = np.full(U, EZ, dtype=np.int64)
  for u in range(min(U, 2)):
    m[u], e[u] = np.float32(0.5) * em_m[0, ext[u]], em_e[0, ext[u]] + 1

  def shifted(v, k, fill):
    r = np.full(U, fill, dtype=v.dtype)
    if U > k:
      r[k:] = v[:U - k]
    return r

  for t in range(1, T):
    m1, e1 = shifted(m, 1, 0), shifted(e, 1, EZ)
    m2, e2 = shifted(m, 2, 0), np.where(skip, shifted(e, 2, EZ), EZ)
    big = np.maximum(np.maximum(e, e1), e2)
    if column_scale:
      big = np.full(U, big.max())
    clip = lambda d: np.maximum(d, -300)                           # ldexp flushes to zero far before that
    s = (np.ldexp(m, clip(e - big)).astype(np.float32) + np.ldexp(m1, clip(e1 - big)).astype(np.float32) +
         np.ldexp(m2, clip(e2 - big)).astype(np.float32)).astype(np.float32)
    v = (s * em_m[t, ext]).astype(np.float32)
    mant, ex = np.frexp(v)
    m, e = mant.astype(np.float32), np.where(v > 0, big + em_e[t, ext] + ex, EZ)
  tail = range(max(U - 2, 0), U)
  big = max(e[u] for u in tail)
  total = sum(float(np.ldexp(np.float64(m[u]), int(max(e[u] - big, -1000)))) for u in tail)
  return -(big + np.log2(total)) * np.log(2.0) if total > 0 else np.inf


def _case(seed, T, L, C=29, repeats=True, scale=1.0):
  rng = np.random.default_rng(seed)
  logits = (rng.normal(size=(T, C)) * scale).astype(np.float32)
  label = rng.integers(0, C - 1, size=L).tolist()
  if repeats and L >= 4:
    label[1], label[3] = label[0], label[2]
  return logits, label


@pytest.mark.parametrize('seed,T,L,scale', [(0, 50, 0, 1.0), (1, 50, 1, 1.0), (2, 120, 45, 1.0), (3, 300, 140, 1.0),
                                             (4, 501, 150, 3.0), (5, 60, 20, 12.0)])
def test_scaled_recursion_matches_float64_log_space(seed, T, L, scale):
  logits, label = _case(seed, T, L, scale=scale)
  ref, _ = O.ctc_loss_and_grad(logits[:, None, :], [label], [T])
  got = scaled_alpha_loss(logits, label)
  assert got == pytest.approx(ref[0], rel=2e-6)


def test_per_state_exponent_keeps_the_paths_a_column_scale_loses():
  """A long utterance with a long label and peaked emissions (logit scale 3): the states that complete the label lie more than
  2^126 below the column maximum for part of the utterance.  One exponent per state follows them (1e-9 relative to float64 here);
  one exponent per column flushes them and comes out ~125 nats high -- what the round-2 variant did on real shapes."""
  T, L = 1377, 400
  logits, label = _case(7, T, L, scale=3.0)
  ref, _ = O.ctc_loss_and_grad(logits[:, None, :], [label], [T])
  assert scaled_alpha_loss(logits, label) == pytest.approx(ref[0], rel=1e-7)
  lost = scaled_alpha_loss(logits, label, column_scale=True)
  assert not np.isfinite(lost) or lost - ref[0] > 10.0
