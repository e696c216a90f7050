"""TensorFlow V2 checkpoint bundles without TensorFlow (speecht_amd/tf_checkpoint.py): the container formats are
pinned by their published known-answer values (CRC-32C vectors of RFC 3720, the table magic, a hand-assembled
table) and by round trips; the variable naming is the reference's (speech_model.py:41,65,148-152)."""
import os
import struct

import numpy as np
import pytest

from speecht_amd import tf_checkpoint as tfc


def test_crc32c_known_answers_and_masking():
  assert tfc.crc32c(b'123456789') == 0xe3069283
  assert tfc.crc32c(bytes(32)) == 0x8a9136aa                      # RFC 3720 B.4
  assert tfc.crc32c(bytes([0xff] * 32)) == 0x62a8ab43
  assert tfc.crc32c(bytes(range(32))) == 0x46dd794e
  assert tfc.crc32c(bytes(reversed(range(32)))) == 0x113fdb5c
  assert tfc.crc32c(b' world', tfc.crc32c(b'hello')) == tfc.crc32c(b'hello world')     # streaming
  for v in (0, 1, 0xdeadbeef, 0xffffffff):
    assert tfc.unmask_crc(tfc.mask_crc(v)) == v and tfc.mask_crc(v) != v


def test_reader_on_a_hand_assembled_table(tmp_path):
  """A table put together byte by byte from the LevelDB format description (not by write_table): one data block
  with prefix compression across a restart point, an index block, an empty meta-index, footer + magic."""
  def block(entries_bytes, restarts):
    body = entries_bytes + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
    return body, body + b'\x00' + struct.pack('<I', tfc.mask_crc(tfc.crc32c(body + b'\x00')))
  e = bytes([0, 5, 2]) + b'apple' + b'v1'                         # shared 0, non-shared 5, value 2
  e += bytes([3, 4, 2]) + b'rove' + b'v2'                         # "app" + "rove" = approve
  r2 = len(e)
  e += bytes([0, 6, 0]) + b'banana'                               # restart point, empty value
  data_body, data_raw = block(e, [0, r2])
  meta_body, meta_raw = block(b'', [0])
  handle = lambda off, size: tfc._put_varint(off) + tfc._put_varint(size)
  idx_entry = bytes([0, 6, len(handle(0, len(data_body)))]) + b'banana' + handle(0, len(data_body))
  idx_body, idx_raw = block(idx_entry, [0])
  footer = handle(len(data_raw), len(meta_body)) + handle(len(data_raw) + len(meta_raw), len(idx_body))
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
  path = str(tmp_path / 'hand.index')
  open(path, 'wb').write(data_raw + meta_raw + idx_raw + footer)
  assert tfc.read_table(path) == [(b'apple', b'v1'), (b'approve', b'v2'), (b'banana', b'')]
  raw = bytearray(open(path, 'rb').read())
  raw[3] ^= 1
  open(path, 'wb').write(raw)
  with pytest.raises(ValueError, match='checksum'):
    tfc.read_table(path)


def test_table_round_trip_many_blocks(tmp_path):
  rng = np.random.default_rng(0)
  entries = [(('layer_%03d/variable/%d' % (i // 7, i)).encode(), bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8)))
             for i in range(500)] + [(b'', b'header')]
  path = str(tmp_path / 't.index')
  tfc.write_table(path, entries, block_size=512)
  assert tfc.read_table(path) == sorted(entries)


def test_snappy_blocks_are_understood():
  # literal "abcd", copy (offset 4, length 4) twice via a 1-byte-offset copy, long literal header form
  stream = bytes([12]) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4])
  assert tfc._snappy_decompress(stream) == b'abcdabcdabcd'
  lit = bytes(range(70))
  assert tfc._snappy_decompress(bytes([70]) + bytes([60 << 2, 69]) + lit) == lit


def test_bundle_round_trip_and_reference_variable_names(tmp_path):
  rng = np.random.default_rng(1)
  tensors = {'Variable': np.array(1234, dtype=np.int32), 'learning_rate': np.array(1e-4, dtype=np.float32),
             'training/beta1_power': np.array(0.5, dtype=np.float32)}
  for i, (w, cin, cout) in enumerate([(48, 5, 6), (7, 6, 6), (1, 6, 29)]):
    for slot in ('', '/Adam', '/Adam_1'):
      prefix = 'training/' if slot and i == 1 else ''             # optimizer slots may sit under a name scope
      tensors['%sconvolution_layer_%d/filters%s' % (prefix, i, slot)] = rng.standard_normal((w, cin, cout)).astype(np.float32)
      tensors['%sconvolution_layer_%d/bias%s' % (prefix, i, slot)] = rng.standard_normal(cout).astype(np.float32)
  prefix = str(tmp_path / 'speechT.ckpt-1234')
  tfc.write_bundle(prefix, tensors)
  assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(t.nbytes for t in tensors.values())
  got = tfc.read_bundle(prefix)
  assert set(got) == set(tensors)
  for k in tensors:
    assert got[k].dtype == tensors[k].dtype and got[k].shape == tensors[k].shape
    np.testing.assert_array_equal(got[k], tensors[k])
  only = tfc.read_bundle(prefix, names=lambda n: n.endswith('/filters'))
  assert sorted(only) == ['convolution_layer_%d/filters' % i for i in range(3)]
  layers, scalars = tfc.split_variables(got)
  assert sorted(layers) == [0, 1, 2] and set(layers[1]) == {(v, s) for v in ('filters', 'bias') for s in (None, 'Adam', 'Adam_1')}
  assert int(scalars['Variable']) == 1234 and float(scalars['beta1_power']) == 0.5
  # a flipped bit in the tensor data is caught by the entry's checksum
  with open(prefix + '.data-00000-of-00001', 'r+b') as f:
    f.seek(100); b = f.read(1); f.seek(100); f.write(bytes([b[0] ^ 4]))
  with pytest.raises(ValueError, match='checksum mismatch in variable'):
    tfc.read_bundle(prefix)


def test_reference_graph_variable_names():
  """The key set ``save_tf`` writes (speecht_amd/speech_model.py asserts it equals this): one line per variable of the
  reference graph.  get_variable names and their Adam slots carry no name scope (speech_model.py:148-152), the
  tf.Variable-built ones do -- global_step and learning_rate are built outside any scope (:41,:65), Adam's
  beta powers inside tf.name_scope('training') (:72-82)."""
  names = tfc.reference_variable_names(11)
  assert len(names) == 4 + 11 * 6
  assert {'Variable', 'learning_rate', 'training/beta1_power', 'training/beta2_power'} <= names
  assert 'beta1_power' not in names and 'training/learning_rate' not in names
  assert {'convolution_layer_0/filters', 'convolution_layer_10/bias', 'convolution_layer_8/filters/Adam',
          'convolution_layer_8/bias/Adam_1'} <= names
  assert not any(n.startswith('training/convolution_layer') for n in names)
  # and the reader sorts exactly these back into layers + scalars
  layers, scalars = tfc.split_variables({n: np.zeros(1, np.float32) for n in names})
  assert sorted(layers) == list(range(11)) and all(len(v) == 6 for v in layers.values())
  assert set(scalars) == {'Variable', 'learning_rate', 'beta1_power', 'beta2_power'}


def test_checkpoint_state_file_and_latest_checkpoint(tmp_path):
  from speecht_amd.speech_model import latest_checkpoint
  d = tmp_path / 'train'
  d.mkdir()
  assert latest_checkpoint(str(d)) is None
  tfc.write_checkpoint_state(str(d), 'speechT.ckpt-20', ['speechT.ckpt-10', 'speechT.ckpt-20'])
  text = open(d / 'checkpoint').read()
  assert text.splitlines()[0] == 'model_checkpoint_path: "speechT.ckpt-20"'
  latest, every = tfc.read_checkpoint_state(str(d))
  assert latest == os.path.join(str(d), 'speechT.ckpt-20') and len(every) == 2
  assert latest_checkpoint(str(d)) is None                       # named, but the bundle is not there
  tfc.write_bundle(latest, {'Variable': np.array(20, dtype=np.int32)})
  assert latest_checkpoint(str(d)) == latest
  # a state file written on another machine holds an absolute path that no longer exists: the basename resolves
  tfc.write_checkpoint_state(str(d), '/somewhere/else/train/speechT.ckpt-20', [])
  assert latest_checkpoint(str(d)) == latest
