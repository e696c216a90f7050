"""What the engine core asks of an arithmetic mode."""


class ModeBase:
  """A mode = the buffers derived from the weights and the activations that only this arithmetic needs, plus the launch
  sequences of the forward and the backward pass.  State lives in the engine core -- shared buffers, streams, freshness flags,
  and the mode's own tensors too (`e.fft`, `e.Xb`, ...: where tests, bench.py and the profiling scripts look for them): a mode
  holds a reference to its engine, `self.e`, and nothing else."""

  def __init__(self, engine):
    self.e = engine

  # ---- the interface -------------------------------------------------------------------------------------------
  # Engine attributes `alloc` assigns that are functions of the shape alone: the engine keeps them per (B, T) and puts them back
  # when a shape comes round again (`Wav2LetterEngine._reenter_shape`) instead of calling `alloc`.  Empty: never cached.
  shape_attrs = ()

  def shape_token(self):
    """What `reenter` needs to know about the shape just described (kept with the cached description)."""
    return None

  def reenter(self, token):
    """A cached shape is current again: redo what `alloc` does that depends on the shape LEFT BEHIND (freshness flags)."""

  def alloc(self, batch):
    """Buffers of this mode for the shape `_ensure_shape` has just described (X, dZ, geo exist)."""
    raise NotImplementedError

  def forward(self):
    raise NotImplementedError

  def backward(self, on_layer_done, wanted):
    raise NotImplementedError

  def refresh_under_ctc(self):
    """Operands of back-prop derived from the weights, rebuilt on the side stream while the CTC recursion runs."""

  def refresh_after_update(self):
    """Operands of the NEXT forward pass derived from the weights, right after clip + Adam."""

  def prepare_forward_graph(self):
    """Everything `forward()` would rebuild or wait for on demand, done before a forward graph is captured / replayed."""
