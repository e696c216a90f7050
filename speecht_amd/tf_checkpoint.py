"""TensorFlow checkpoint bundles (``speechT.ckpt-N.index`` + ``.data-00000-of-00001``) without TensorFlow.

The reference saves and restores with ``tf.train.Saver(tf.global_variables())`` (speech_model.py:122,251-260;
training.py:86-88) and publishes trained weights in that form (README.md:72-79).  TF >= 0.12 writes the "V2"
tensor bundle, which is two plain formats stacked:

* ``<prefix>.index`` -- a LevelDB-style sorted string table: data blocks of prefix-compressed
  (shared, non_shared, value_len, key suffix, value) entries with a restart array, a meta-index and an index block
  of (last key -> BlockHandle), a 48-byte footer (two handles + magic 0xdb4775248b80fb57); every block is followed
  by a 1-byte compression tag (0 = none, 1 = snappy) and a masked CRC-32C.  Key "" holds ``BundleHeaderProto``
  (num_shards, endianness, version), every other key is a variable name holding ``BundleEntryProto``
  (dtype, shape, shard_id, offset, size, crc32c of the tensor bytes, masked).
* ``<prefix>.data-SSSSS-of-NNNNN`` -- the tensors' raw little-endian bytes at the recorded offsets.

plus the text file ``checkpoint`` (``model_checkpoint_path: "speechT.ckpt-N"``).  This module reads that format
(so that published speechT weights restore into the engine) and writes it (so that the reference can restore what
this package trained).  PARITY UNPINNED: TensorFlow is not installable here, the format is restated from its
published description and pinned by round trips and by the format's own checksums; variable names are the
reference's (speech_model.py:41,65,148-152 + the Adam slot names of tf.train.AdamOptimizer).
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8')}
_DT_OF = {np.dtype('float32'): DT_FLOAT, np.dtype('int32'): DT_INT32, np.dtype('int64'): DT_INT64}
_MASK_DELTA = 0xa282ead8


def crc32c(data, crc=0):
  """CRC-32C of bytes / a contiguous numpy array (the library's host routine, st_host_crc32c)."""
  import ctypes
  from . import _lib
  buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
  buf = np.ascontiguousarray(buf)
  return int(_lib.load().st_host_crc32c(ctypes.c_void_p(buf.ctypes.data), buf.size, crc))


def mask_crc(crc):
  """LevelDB / TF stored form of a CRC: rotate right by 15 and add a constant."""
  return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
  rot = (masked - _MASK_DELTA) & 0xffffffff
  return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints and the two tiny protobuf messages --------------------------------------------------------------
def _put_varint(value):
  out = bytearray()
  while True:
    b = value & 0x7f
    value >>= 7
    out.append(b | (0x80 if value else 0))
    if not value:
      return bytes(out)


def _get_varint(buf, pos):
  result, shift = 0, 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7f) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


def _pb_fields(buf):
  """Yield (field number, wire type, value) of one protobuf message (varint, 64-bit, bytes, 32-bit)."""
  pos = 0
  while pos < len(buf):
    tag, pos = _get_varint(buf, pos)
    field, wire = tag >> 3, tag & 7
    if wire == 0:
      value, pos = _get_varint(buf, pos)
    elif wire == 1:
      value, pos = buf[pos:pos + 8], pos + 8
    elif wire == 2:
      n, pos = _get_varint(buf, pos)
      value, pos = buf[pos:pos + n], pos + n
    elif wire == 5:
      value, pos = buf[pos:pos + 4], pos + 4
    else:
      raise ValueError('unsupported protobuf wire type {}'.format(wire))
    yield field, wire, value


def _pb_varint_field(field, value):
  return _put_varint(field << 3) + _put_varint(value)


def _pb_bytes_field(field, payload):
  return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def _encode_entry(dtype, shape, offset, size, crc_masked):
  """BundleEntryProto: dtype = 1, shape = 2 (TensorShapeProto.dim = 2 {size = 1}), shard_id = 3, offset = 4,
  size = 5, crc32c = 6 (fixed32)."""
  dims = b''.join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)
  out = _pb_varint_field(1, dtype) + _pb_bytes_field(2, dims)
  if offset:
    out += _pb_varint_field(4, offset)            # shard_id 0 and offset 0 are proto3 defaults: omitted
  out += _pb_varint_field(5, size)
  out += _put_varint((6 << 3) | 5) + struct.pack('<I', crc_masked)
  return out


def _decode_entry(buf):
  entry = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
  for field, wire, value in _pb_fields(buf):
    if field == 1:
      entry['dtype'] = value
    elif field == 2:
      for f2, _, dim in _pb_fields(value):
        if f2 == 2:
          size = 0
          for f3, _, v3 in _pb_fields(dim):
            if f3 == 1:
              size = v3
          entry['shape'].append(size)
    elif field == 3:
      entry['shard_id'] = value
    elif field == 4:
      entry['offset'] = value
    elif field == 5:
      entry['size'] = value
    elif field == 6:
      entry['crc32c'] = struct.unpack('<I', value)[0]
    elif field == 7:
      entry['sliced'] = True
  return entry


# ---- snappy (index blocks may be compressed; TF's bundle writer does not, other writers might) -----------------
def _snappy_decompress(buf):
  n, pos = _get_varint(buf, 0)
  out = bytearray()
  while pos < len(buf):
    tag = buf[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:                                   # literal
      length = tag >> 2
      if length >= 60:
        extra = length - 59
        length = int.from_bytes(buf[pos:pos + extra], 'little')
        pos += extra
      length += 1
      out += buf[pos:pos + length]
      pos += length
      continue
    if kind == 1:
      length, offset = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | buf[pos]
      pos += 1
    elif kind == 2:
      length, offset = (tag >> 2) + 1, int.from_bytes(buf[pos:pos + 2], 'little')
      pos += 2
    else:
      length, offset = (tag >> 2) + 1, int.from_bytes(buf[pos:pos + 4], 'little')
      pos += 4
    for _ in range(length):                         # copies may overlap their own output
      out.append(out[-offset])
  if len(out) != n:
    raise ValueError('corrupt snappy block')
  return bytes(out)


# ---- the string table ------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
  body, kind = data[offset:offset + size], data[offset + size]
  stored = struct.unpack('<I', data[offset + size + 1:offset + size + 5])[0]
  if verify and unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
    raise ValueError('checksum mismatch in table block at {}'.format(offset))
  if kind == 1:
    body = _snappy_decompress(body)
  elif kind != 0:
    raise ValueError('unknown block compression {}'.format(kind))
  n_restarts = struct.unpack('<I', body[-4:])[0]
  end = len(body) - 4 - 4 * n_restarts
  pos, key, out = 0, b'', []
  while pos < end:
    shared, pos = _get_varint(body, pos)
    non_shared, pos = _get_varint(body, pos)
    value_len, pos = _get_varint(body, pos)
    key = key[:shared] + body[pos:pos + non_shared]
    pos += non_shared
    out.append((key, body[pos:pos + value_len]))
    pos += value_len
  return out


def read_table(path, verify=True):
  """All (key, value) pairs of a LevelDB-format table file, in key order."""
  with open(path, 'rb') as f:
    data = f.read()
  if len(data) < 48 or struct.unpack('<Q', data[-8:])[0] != TABLE_MAGIC:
    raise ValueError('{} is not a TensorFlow checkpoint index (bad magic)'.format(path))
  footer = data[-48:]
  _, pos = _get_varint(footer, 0)                   # meta-index handle (unused)
  _, pos = _get_varint(footer, pos)
  index_off, pos = _get_varint(footer, pos)
  index_size, pos = _get_varint(footer, pos)
  out = []
  for _, handle in _read_block(data, index_off, index_size, verify):
    off, p = _get_varint(handle, 0)
    size, p = _get_varint(handle, p)
    out.extend(_read_block(data, off, size, verify))
  return out


def _build_block(entries, restart_interval=16):
  body, restarts, last = bytearray(), [], b''
  for i, (key, value) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(body))
    else:
      while shared < min(len(last), len(key)) and last[shared] == key[shared]:
        shared += 1
    body += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
    last = key
  if not restarts:
    restarts = [0]
  for r in restarts:
    body += struct.pack('<I', r)
  body += struct.pack('<I', len(restarts))
  return bytes(body)


def write_table(path, entries, block_size=4096):
  """Write sorted (key, value) pairs as an uncompressed LevelDB-format table (what TF's BundleWriter emits)."""
  entries = sorted(entries)
  out = bytearray()

  def emit(block):
    handle = _put_varint(len(out)) + _put_varint(len(block))
    out.extend(block + b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
    return handle
  index, current, size = [], [], 0
  for key, value in entries:
    current.append((key, value))
    size += len(key) + len(value) + 3
    if size >= block_size:
      index.append((current[-1][0], emit(_build_block(current))))
      current, size = [], 0
  if current or not index:
    index.append((current[-1][0] if current else b'', emit(_build_block(current))))
  meta = emit(_build_block([]))
  idx = emit(_build_block(index, restart_interval=1))
  footer = meta + idx
  out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
  tmp = path + '.tmp'
  with open(tmp, 'wb') as f:
    f.write(out)
  os.replace(tmp, path)


# ---- tensor bundles --------------------------------------------------------------------------------------------
def read_bundle(prefix, names=None, verify=True):
  """{variable name: ndarray} of the checkpoint ``prefix`` (e.g. ``train/run/speechT.ckpt-1000``).  ``names``: an
  optional predicate on the variable name (skip what is not needed: a speechT checkpoint is 290 MB)."""
  entries = read_table(prefix + '.index', verify)
  if not entries or entries[0][0] != b'':
    raise ValueError('{}.index has no bundle header'.format(prefix))
  num_shards, endianness = 1, 0
  for field, _, value in _pb_fields(entries[0][1]):
    if field == 1:
      num_shards = value
    elif field == 2:
      endianness = value
  if endianness != 0:
    raise ValueError('big-endian checkpoints are not supported')
  out, shards = {}, {}
  for key, value in entries[1:]:
    name = key.decode()
    if names is not None and not names(name):
      continue
    e = _decode_entry(value)
    if e['sliced']:
      raise ValueError('partitioned variable {} is not supported'.format(name))
    if e['dtype'] not in _DTYPES:
      raise ValueError('variable {} has unsupported dtype {}'.format(name, e['dtype']))
    if e['shard_id'] not in shards:
      shards[e['shard_id']] = np.memmap('{}.data-{:05d}-of-{:05d}'.format(prefix, e['shard_id'], num_shards), mode='r')
    raw = np.asarray(shards[e['shard_id']][e['offset']:e['offset'] + e['size']])
    dt = _DTYPES[e['dtype']]
    if raw.size != int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize:
      raise ValueError('variable {}: {} bytes for shape {}'.format(name, raw.size, e['shape']))
    if verify and e['crc32c'] is not None and unmask_crc(e['crc32c']) != crc32c(raw):
      raise ValueError('checksum mismatch in variable {}'.format(name))
    out[name] = raw.view(dt).reshape(e['shape']).copy()
  return out


def write_bundle(prefix, tensors):
  """Write {name: ndarray (float32 / int32 / int64)} as a single-shard V2 bundle."""
  data_path = '{}.data-00000-of-00001'.format(prefix)
  entries, offset = [], 0
  # BundleHeaderProto: num_shards = 1 (field 1), endianness LITTLE = 0 (default), version {producer = 1} (field 3)
  entries.append((b'', _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1))))
  with open(data_path + '.tmp', 'wb') as f:
    for name in sorted(tensors):
      a = np.asarray(tensors[name])
      if a.dtype not in _DT_OF:
        raise ValueError('variable {} has unsupported dtype {}'.format(name, a.dtype))
      shape = a.shape                                             # () for the scalars (ascontiguousarray would make it (1,))
      raw = np.ascontiguousarray(a.astype(a.dtype.newbyteorder('<'), copy=False)).reshape(-1).view(np.uint8)
      f.write(raw.tobytes())
      entries.append((name.encode(), _encode_entry(_DT_OF[a.dtype], shape, offset, raw.size, mask_crc(crc32c(raw)))))
      offset += raw.size
  os.replace(data_path + '.tmp', data_path)
  write_table(prefix + '.index', entries)


# ---- the `checkpoint` state file (text CheckpointState proto) ---------------------------------------------------
def read_checkpoint_state(directory):
  """(latest, [all]) checkpoint prefixes named by ``<directory>/checkpoint`` in TF's text format, resolved against
  the directory when relative (tf.train.get_checkpoint_state, speech_model.py:252); None if there is none."""
  path = os.path.join(directory, 'checkpoint')
  if not os.path.exists(path):
    return None
  text = open(path).read()
  latest = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', text, re.M)
  if not latest:
    return None
  resolve = lambda p: p if os.path.isabs(p) else os.path.join(directory, p)
  return resolve(latest.group(1)), [resolve(p) for p in re.findall(r'^all_model_checkpoint_paths:\s*"(.*)"\s*$', text, re.M)]


def write_checkpoint_state(directory, latest, all_paths):
  with open(os.path.join(directory, 'checkpoint.tmp'), 'w') as f:
    f.write('model_checkpoint_path: "{}"\n'.format(latest))
    for p in all_paths:
      f.write('all_model_checkpoint_paths: "{}"\n'.format(p))
  os.replace(os.path.join(directory, 'checkpoint.tmp'), os.path.join(directory, 'checkpoint'))


# ---- the reference's variables <-> the engine --------------------------------------------------------------------
_LAYER_VAR = re.compile(r'^(?:.*/)?convolution_layer_(\d+)/(filters|bias)(?:/(Adam|Adam_1))?$')


def reference_variable_names(n_layers):
  """The names ``tf.train.Saver(tf.global_variables())`` of the reference graph expects (speech_model.py:41,65,72-82,
  122,148-152), as TF 1.x naming rules give them: ``get_variable`` variables and the Adam slots derived from their
  names ignore ``tf.name_scope``; ``tf.Variable``-created ones honour it -- the unnamed ``Variable`` (global_step,
  built outside any scope), ``learning_rate`` (outside), and Adam's non-slot accumulators, which
  ``apply_gradients`` creates inside ``tf.name_scope('training')``: ``training/beta1_power`` / ``training/beta2_power``.
  NOT checked against a TensorFlow installation (none exists here): restated from TF 1.1's optimizer /
  slot_creator naming; ``Saver.restore`` raises NotFound for a missing key, so this set is what ``save_tf`` writes."""
  names = {'Variable', 'learning_rate', 'training/beta1_power', 'training/beta2_power'}
  for i in range(n_layers):
    for var in ('filters', 'bias'):
      base = 'convolution_layer_{}/{}'.format(i, var)
      names.update({base, base + '/Adam', base + '/Adam_1'})
  return names


def split_variables(tensors):
  """Sort a speechT checkpoint's variables: ({layer: {('filters'|'bias', None|'Adam'|'Adam_1'): array}}, scalars).
  Names (speech_model.py): ``convolution_layer_<i>/filters`` [W, Cin, Cout] and ``.../bias`` [Cout] (:148-152), their
  Adam slots ``.../Adam`` (m) and ``.../Adam_1`` (v) under whatever name scope the optimizer was built in (:72-82),
  ``learning_rate`` (:65), the unnamed ``Variable`` = global_step (:41), ``beta1_power`` / ``beta2_power``."""
  layers, scalars = {}, {}
  for name, value in tensors.items():
    m = _LAYER_VAR.match(name)
    if m:
      layers.setdefault(int(m.group(1)), {})[(m.group(2), m.group(3))] = value
    else:
      scalars[name.split('/')[-1]] = value
  return layers, scalars
