"""Data-parallel training: utterance sharding + bucketed gradient all-reduce (RCCL over xGMI).

The reference is single-replica (training.py:46); DP is new design constrained only by the math
of speech_model.py:75-82: avg_loss is the mean over the GLOBAL batch, so every rank scales its
CTC gradient by 1/(B_local * world) and the flat gradient buffers are SUM-all-reduced before the
global-norm clip -- all replicas then apply the identical clip + Adam update and stay bit-identical.

One process per GPU; ``torch.distributed`` (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests) is
used purely as the collective transport.  Buckets are contiguous slices of the flat gradient buffer
in the order back-prop finishes them (L10+L9, L8, L7..L4, L3..L1, L0), launched asynchronously so that the
xGMI transfer overlaps the remaining back-prop kernels.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def job():
  """(rank, world) of the running job: torch.distributed's view once the process group exists, (0, 1) otherwise."""
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(), dist.get_world_size()
  return 0, 1


def init_job(flags=None):
  """Join the data-parallel job a launcher started this process for (`torchrun --nproc-per-node N speecht-cli train ...`:
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment): one process per GPU, ``flags.device`` becomes this rank's GPU,
  backend nccl (= RCCL; ST_DIST_BACKEND=gloo and ST_SHARE_GPU=1 are test knobs: host transport, every rank on cuda:0).  Without
  WORLD_SIZE > 1 in the environment nothing happens.  Returns (rank, world)."""
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world <= 1 or dist.is_initialized():
    return job()
  rank, local_rank = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29571')
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # the host driver only supports dmabuf IPC
  backend = os.environ.get('ST_DIST_BACKEND', 'nccl')
  device = getattr(flags, 'device', None) or 'cuda:0'
  if str(device).startswith('cuda'):
    if os.environ.get('ST_SHARE_GPU'):
      local_rank = 0
    device = 'cuda:%d' % local_rank
    torch.cuda.set_device(local_rank)
  if flags is not None:
    flags.device = device
  if backend == 'nccl':
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(device))
  else:
    dist.init_process_group(backend, rank=rank, world_size=world)
  return rank, world


def broadcast_seed(group=None):
  """One random 31-bit seed, the same on every rank (rank 0 draws it)."""
  import random
  box = [random.SystemRandom().randrange(1 << 31)]
  if dist.is_initialized() and dist.get_world_size(group) > 1:
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
  return int(box[0])


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
  if big >= 4:
    # the layers below the big one finish last: the exchange of the final bucket is the part of the communication that
    # no kernel hides, so they go in three pieces -- the upper half, the lower half without the bottom layer, and the bottom
    # layer alone (3.8 MB of the model's 96: what is left over when back-prop ends; round 4 left 9.4 MB, and a 2.5 ms bf16 step
    # has no slack for it -- bench.py's comm_model_8gpu prices the tail per assumed bus bandwidth)
    mid = big // 2
    groups.append((mid, big - 1))
    if mid > 1:
      groups.append((1, mid - 1))
      groups.append((0, 0))
    else:
      groups.append((0, mid - 1))
  elif big > 0:
    groups.append((0, big - 1))
  return [(lo, layer_offsets[lo][0], layer_offsets[hi][1]) for lo, hi in groups]


class RcclCommunicator:
  """An RCCL communicator owned by libspeecht_hip.so (``st_comm_*`` / ``st_allreduce_*`` of
  include/speecht_hip.h) plus the side stream its collectives run on.

  ``torch.distributed`` is only the bootstrap channel here: rank 0's 128-byte unique id is broadcast as
  a Python object, then every rank joins with ``st_comm_init``.  Collectives are ordered after the
  compute stream with an event and joined back with another, so the xGMI transfer of one gradient
  bucket overlaps the back-prop kernels of the layers below it.
  """

  def __init__(self, device, group=None):
    from . import _lib
    self._lib = _lib
    self.device = torch.device(device)
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    lib = _lib.load()
    n = lib.st_comm_unique_id_bytes()
    uid = ctypes.create_string_buffer(n)
    if self.rank == 0:
      _lib.call('st_comm_unique_id', uid, n)
    if self.world > 1:
      box = [uid.raw]
      dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
      uid = ctypes.create_string_buffer(box[0], n)
    handle = ctypes.c_void_p()
    with torch.cuda.device(self.device):        # RCCL binds the communicator to the current device
      _lib.call('st_comm_init', uid, n, self.rank, self.world, ctypes.byref(handle))
      from .engine import role_stream
      self.stream = role_stream(self.device, 'collective')
    self._handle = handle
    self._joined = None

  def all_reduce_slices(self, flat, slices, after_stream):
    """SUM-all-reduce ``flat[s:e]`` for every (s, e), in one RCCL group, once ``after_stream`` has
    reached this point."""
    ready = torch.cuda.Event()
    ready.record(after_stream)
    self.stream.wait_event(ready)
    k = len(slices)
    starts = (ctypes.c_size_t * k)(*[s for s, _ in slices])
    counts = (ctypes.c_size_t * k)(*[e - s for s, e in slices])
    self._lib.call('st_allreduce_buckets_f32', self._handle, ctypes.c_void_p(flat.data_ptr()), starts, counts, k,
                   ctypes.c_void_p(self.stream.cuda_stream))
    self._joined = torch.cuda.Event()
    self._joined.record(self.stream)

  def join(self, stream):
    """Make ``stream`` wait for every collective issued so far."""
    if self._joined is not None:
      stream.wait_event(self._joined)
      self._joined = None

  def count(self):
    """Number of ranks in the RCCL communicator itself (ncclCommCount), independent of torch.distributed's view."""
    n = ctypes.c_int(0)
    self._lib.call('st_comm_count', self._handle, ctypes.byref(n))
    return n.value

  def close(self):
    if self._handle:
      self._lib.call('st_comm_destroy', self._handle)
      self._handle = ctypes.c_void_p()


class GradientAllReducer:
  """Sum-all-reduces slices of one flat gradient tensor as back-prop completes them.

  transport 'torch' (default): ``torch.distributed`` async all-reduce (nccl == RCCL on ROCm, gloo on CPU).
  transport 'rccl': the library's own communicator (``st_allreduce_buckets_f32``); env ST_ALLREDUCE=rccl
  selects it without code changes.
  """

  def __init__(self, flat_grads, layer_offsets, group=None, force=False, transport=None, compute_stream=None):
    self.flat = flat_grads
    self.group = group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.transport = transport or os.environ.get('ST_ALLREDUCE', 'torch')
    if self.transport not in ('torch', 'rccl'):
      raise ValueError('transport must be "torch" or "rccl", got {!r}'.format(self.transport))
    # force: exercise the collective on 1 rank
    self.active = self.world > 1 or (force and (dist.is_initialized() or self.transport == 'rccl'))
    sizes = [e - s for s, e in layer_offsets]
    self.buckets = default_buckets(sizes, layer_offsets)
    self._ready_at = {lo: (s, e) for lo, s, e in self.buckets}
    self._pending = []
    self._compute_stream = compute_stream
    # called once per step right after the FIRST bucket (the top layers' slice, which also carries the update gate and the
    # mean-loss slot) has been handed to the all-reduce, with a function `wait(stream)` that makes `stream` wait for that
    # bucket's reduction only: SpeechModel.step reads the reduced slots back from there, long before back-prop ends
    self.after_first_bucket = None
    self.comm = RcclCommunicator(flat_grads.device, group) if (self.active and self.transport == 'rccl') else None

  @property
  def hook_layers(self):
    """The layers whose completion makes a bucket ready (the lowest layer of each bucket): the only ones
    ``engine.backward`` has to call ``on_layer_done`` for."""
    return set(self._ready_at) if self.active else set()

  def readback_stream(self):
    """Where a read-back that must follow a bucket's reduction is enqueued: the library communicator's stream (in order behind
    the all-reduce), or the process's 'collective' role stream for the torch transport -- both on a hardware queue the compute
    stream does not use (engine_streams.role_stream)."""
    if self.comm is not None:
      return self.comm.stream
    if self.flat.device.type != 'cuda':
      return None                      # (host tensors: the CPU tests of the control flow)
    from .engine_streams import role_stream
    return role_stream(self.flat.device, 'collective')

  def _stream(self):
    return self._compute_stream if self._compute_stream is not None else torch.cuda.current_stream(self.flat.device)

  def on_layer_done(self, i):
    if not self.active or i not in self._ready_at:
      return
    s, e = self._ready_at[i]
    if self.comm is not None:
      self.comm.all_reduce_slices(self.flat, [(s, e)], self._stream())
      reduced = self.comm._joined
      wait = lambda stream: stream.wait_event(reduced)      # noqa: E731
    else:
      work = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
      self._pending.append(work)

      def wait(stream, work=work):
        if stream is None:
          work.wait()                # (host tensors)
          return
        with torch.cuda.stream(stream):
          work.wait()                # nccl: a stream-side wait; gloo (tests): blocks the host until the bucket is reduced
    if i == self.buckets[0][0] and self.after_first_bucket is not None:
      self.after_first_bucket(wait)

  def finish(self):
    if self.comm is not None:
      self.comm.join(self._stream())
    for w in self._pending:
      w.wait()
    self._pending = []


def make_reducer(flat_grads, layer_offsets, group=None, transport=None, force=False):
  """The gradient exchange of a data-parallel job, by the rule bench.py and SpeechModel.enable_data_parallel share: the library's
  own RCCL communicator (st_allreduce_buckets_f32) when the job runs on the nccl backend with one GPU per rank -- but only if it
  really spans the job (its ncclCommCount equals the world size on EVERY rank), else every rank falls back to
  torch.distributed together.  ``transport``: 'rccl' / 'torch' to force one (env ST_ALLREDUCE does the same).  Returns
  (reducer, note) -- note says why a fallback happened, or None."""
  world = dist.get_world_size(group) if dist.is_initialized() else 1
  want = transport or os.environ.get('ST_ALLREDUCE')
  if want is None:
    nccl = dist.is_initialized() and dist.get_backend(group) == 'nccl'
    want = 'rccl' if (world > 1 and nccl and not os.environ.get('ST_SHARE_GPU')) else 'torch'
  reducer, note = None, None
  if want == 'rccl':
    ok = 1
    try:
      reducer = GradientAllReducer(flat_grads, layer_offsets, group, force=force, transport='rccl')
      ok = int(reducer.comm is not None and reducer.comm.count() == world)
    except Exception as e:      # noqa: BLE001 -- whatever the set-up raises, the exchange must still happen somehow
      ok, note = 0, 'library communicator failed: %r' % (e,)
    if dist.is_initialized() and world > 1:
      flag = torch.tensor([ok], dtype=torch.int32, device=flat_grads.device)
      dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
      ok = int(flag[0])
    if not ok:
      note = note or 'library communicator does not span the job (count != world) on some rank'
      reducer = None
  if reducer is None:
    reducer = GradientAllReducer(flat_grads, layer_offsets, group, force=force, transport='torch')
  return reducer, note


def all_reduce_mean_scalar(value, device, group=None):
  """Mean of a host scalar over ranks (loss reporting)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return value
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, group=group)
  return float(t[0]) / dist.get_world_size(group)
