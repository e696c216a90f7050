"""Data-parallel training: utterance sharding + bucketed gradient all-reduce (RCCL over xGMI).

The reference is single-replica (training.py:46); DP is new design constrained only by the math
of speech_model.py:75-82: avg_loss is the mean over the GLOBAL batch, so every rank scales its
CTC gradient by 1/(B_local * world) and the flat gradient buffers are SUM-all-reduced before the
global-norm clip -- all replicas then apply the identical clip + Adam update and stay bit-identical.

One process per GPU; ``torch.distributed`` (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests) is
used purely as the collective transport.  Buckets are contiguous slices of the flat gradient buffer
in the order back-prop finishes them (L10+L9, L8, L7..L0), launched asynchronously so that the
xGMI transfer overlaps the remaining back-prop kernels.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
  """Contiguous equal shards; requires n_items % world == 0 so mean-of-means is exact."""
  if n_items % world:
    raise ValueError('global batch {} is not divisible by world size {}'.format(n_items, world))
  per = n_items // world
  return rank * per, (rank + 1) * per


def default_buckets(layer_sizes, layer_offsets):
  """layer_offsets[i] = (start, end) of layer i (filters+bias) in the flat buffer.
  Returns [(first_layer, start, end)] in launch order: a bucket is ready when back-prop has
  produced its lowest-numbered layer."""
  n = len(layer_offsets)
  if n < 3:
    return [(0, layer_offsets[0][0], layer_offsets[-1][1])]
  big = max(range(n), key=lambda i: layer_sizes[i])
  groups = []
  if big + 1 < n:
    groups.append((big + 1, n - 1))
  groups.append((big, big))
  if big > 0:
    groups.append((0, big - 1))
  return [(lo, layer_offsets[lo][0], layer_offsets[hi][1]) for lo, hi in groups]


class GradientAllReducer:
  """Sum-all-reduces slices of one flat gradient tensor as back-prop completes them."""

  def __init__(self, flat_grads, layer_offsets, group=None, force=False):
    self.flat = flat_grads
    self.group = group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.active = self.world > 1 or (force and dist.is_initialized())   # force: exercise the collective on 1 rank
    sizes = [e - s for s, e in layer_offsets]
    self.buckets = default_buckets(sizes, layer_offsets)
    self._ready_at = {lo: (s, e) for lo, s, e in self.buckets}
    self._pending = []

  def on_layer_done(self, i):
    if not self.active or i not in self._ready_at:
      return
    s, e = self._ready_at[i]
    self._pending.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

  def finish(self):
    for w in self._pending:
      w.wait()
    self._pending = []


def all_reduce_mean_scalar(value, device, group=None):
  """Mean of a host scalar over ranks (loss reporting)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return value
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, group=group)
  return float(t[0]) / dist.get_world_size(group)
