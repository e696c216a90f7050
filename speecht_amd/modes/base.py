"""What the engine core asks of an arithmetic mode."""


class ModeBase:
  """A mode = the buffers derived from the weights and the activations that only this arithmetic needs, plus the launch
  sequences of the forward and the backward pass.  State lives in the engine (shared buffers, streams, freshness flags):
  attribute reads that the mode class does not define fall through to the engine, attribute writes go to the engine."""

  def __init__(self, engine):
    object.__setattr__(self, 'e', engine)

  def __getattr__(self, name):                       # only reached when the mode class has no such attribute
    return object.__getattribute__(object.__getattribute__(self, 'e'), name)

  def __setattr__(self, name, value):
    setattr(object.__getattribute__(self, 'e'), name, value)

  # ---- the interface -------------------------------------------------------------------------------------------
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
