"""CPU: nlt_amd.ckpt reads TensorFlow's tensor-bundle checkpoint format (SSTable index + raw data shard) without TF.
No TensorFlow exists here, so the fixture is produced by the small WRITER below, restated from the same published
layouts (table_format.txt, tensor_bundle.proto) -- a round trip, not a pin against a real TF file (said so in ckpt.py)."""
import os
import struct

import numpy as np
import pytest
import torch

from nlt_amd import ckpt
from oracle import nlt_oracle as O
from test_host_orchestration import make


def _vi(n):
    out = b''
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def _block(entries, restart_every=16):
    """LevelDB block: prefix-compressed entries + restart array (+ 5-byte trailer appended by the caller)."""
    buf, restarts, last = b'', [], b''
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        buf += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        last = k
    for r in restarts or [0]:
        buf += struct.pack('<I', r)
    return buf + struct.pack('<I', max(len(restarts), 1))


def _with_trailer(block):
    return block + b'\x00' + struct.pack('<I', ckpt.masked_crc(block + b'\x00'))


def _pb_varint(field, v):
    return _vi(field << 3) + _vi(v)


def _pb_bytes(field, b):
    return _vi(field << 3 | 2) + _vi(len(b)) + b


def write_bundle(prefix, tensors, extra_string_keys=()):
    """tensors: {key: float32 / int64 ndarray}; sorted keys, one data shard, two data blocks in the index table."""
    enum = {np.dtype(np.float32): 1, np.dtype(np.int64): 9}
    data, entries = b'', [(b'', _pb_varint(1, 1) + _pb_varint(2, 0) + _pb_bytes(3, _pb_varint(1, 1)))]
    items = sorted([(k.encode(), v) for k, v in tensors.items()] + [(k.encode(), None) for k in extra_string_keys])
    for k, arr in items:
        if arr is None:                                              # a DT_STRING entry (e.g. the object graph): readers skip it
            payload = b'\x05hello'
            e = _pb_varint(1, 7) + _pb_bytes(2, b'') + _pb_varint(4, len(data)) + _pb_varint(5, len(payload))
        else:
            payload = np.ascontiguousarray(arr).tobytes()
            shape = b''.join(_pb_bytes(2, _pb_varint(1, d)) for d in arr.shape)
            e = (_pb_varint(1, enum[arr.dtype]) + _pb_bytes(2, shape) + _pb_varint(4, len(data)) + _pb_varint(5, len(payload))
                 + _vi(6 << 3 | 5) + struct.pack('<I', ckpt.masked_crc(payload)))
        entries.append((k, e))
        data += payload
    half = len(entries) // 2
    blocks = [_with_trailer(_block(entries[:half])), _with_trailer(_block(entries[half:]))]
    index_entries, off, out = [], 0, b''
    for blk, last_key in zip(blocks, (entries[half - 1][0], entries[-1][0])):
        index_entries.append((last_key, _vi(off) + _vi(len(blk) - 5)))
        out += blk
        off += len(blk)
    meta = _with_trailer(_block([]))
    meta_off = len(out); out += meta
    idx = _with_trailer(_block(index_entries, restart_every=1))
    idx_off = len(out); out += idx
    foot = _vi(meta_off) + _vi(len(meta) - 5) + _vi(idx_off) + _vi(len(idx) - 5)
    out += foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', ckpt.TABLE_MAGIC)
    open(prefix + '.index', 'wb').write(out)
    open(prefix + '.data-00000-of-00001', 'wb').write(data)


def reference_keys(weights):
    """Oracle weights -> the variable keys tf.train.Checkpoint(net=model) produces for the reference's layer aliases."""
    out = {}
    for net in ('query', 'obs'):
        for li, convs in enumerate(weights[net]):
            for ci, (k, b) in enumerate(convs):
                mid = '' if len(convs) == 1 else 'layer_with_weights-%d/' % ci
                out['net/net_%s_layer%d/%skernel/.ATTRIBUTES/VARIABLE_VALUE' % (net, li, mid)] = np.asarray(k, np.float32)
                out['net/net_%s_layer%d/%sbias/.ATTRIBUTES/VARIABLE_VALUE' % (net, li, mid)] = np.asarray(b, np.float32)
    return out


def test_crc32c_known_answers_and_chunked_form():
    assert ckpt.crc32c(b'123456789') == 0xe3069283                  # the CRC-32C check value
    assert ckpt.crc32c(b'\x00' * 32) == 0x8a9136aa                   # RFC 3720 B.4
    rng = np.random.default_rng(0)
    for n in (4096, 5000, 70001, 300007):                                # the NumPy chunk-parallel path == the byte loop
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert ckpt.crc32c(d) == ckpt._crc_raw(d, 0xffffffff) ^ 0xffffffff


def test_checkpoint_round_trip_into_the_model(tmp_path):
    om2 = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, seed=2)
    w2 = om2.numpy_weights()
    tensors = reference_keys(w2)
    tensors['step/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(1234, np.int64)
    tensors['optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE'] = np.array(77, np.int64)
    tensors['net/net_query_layer0/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE'] = np.zeros((1, 1, 5, 16), np.float32)
    prefix = os.path.join(tmp_path, 'ckpt-100')
    write_bundle(prefix, tensors, extra_string_keys=('_CHECKPOINTABLE_OBJECT_GRAPH',))
    raw = ckpt.read_bundle(prefix)
    assert int(raw['step/.ATTRIBUTES/VARIABLE_VALUE']) == 1234 and '_CHECKPOINTABLE_OBJECT_GRAPH' not in raw
    got = ckpt.reference_weights(prefix)
    assert [len(l) for l in got['query']] == [len(l) for l in w2['query']] and len(got['obs']) == 7
    for net in ('query', 'obs'):
        for lg, lr in zip(got[net], w2[net]):
            for (k, b), (kr, br) in zip(lg, lr):
                assert np.array_equal(k, kr) and np.array_equal(b, br)
    _, pm = make(256, 64, 32)                                        # product model holding seed-1 weights (CPU tensors)
    ckpt.load_reference_checkpoint(pm, prefix)
    for layer, lw in zip(pm.net['query'].layers, w2['query']):
        convs = [layer] if hasattr(layer, 'set_weights') else [c for c, _ in layer.convs()]
        for c, (k, b) in zip(convs, lw):
            assert torch.equal(c.kernel, torch.tensor(k)) and torch.equal(c.bias, torch.tensor(b))


def test_corruption_and_foreign_files_are_rejected(tmp_path):
    prefix = os.path.join(tmp_path, 'ckpt-1')
    write_bundle(prefix, {'net/net_query_layer0/kernel/.ATTRIBUTES/VARIABLE_VALUE': np.ones((1, 1, 5, 16), np.float32),
                          'net/net_query_layer0/bias/.ATTRIBUTES/VARIABLE_VALUE': np.zeros(16, np.float32)})
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[5] ^= 0xff
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    with pytest.raises(ValueError):
        ckpt.read_bundle(prefix)
    assert ckpt.read_bundle(prefix, verify=False)                    # ... unless told not to check
    open(prefix + '.index', 'wb').write(b'not a table' * 10)
    with pytest.raises(ValueError):
        ckpt.read_bundle(prefix)


def test_import_refuses_norm_layer_variables_and_half_written_convs(tmp_path):
    """A checkpoint with norm layers (gamma / beta) would shift the layer_with_weights-N numbering: refused, not misaligned;
    a conv without its bias is named in the error."""
    from nlt_amd import ckpt
    base = {'net/net_query_layer0/kernel/.ATTRIBUTES/VARIABLE_VALUE': np.zeros((1, 1, 5, 16), np.float32),
            'net/net_query_layer0/bias/.ATTRIBUTES/VARIABLE_VALUE': np.zeros(16, np.float32)}
    t = dict(base)
    t['net/net_query_layer1/layer_with_weights-1/gamma/.ATTRIBUTES/VARIABLE_VALUE'] = np.ones(16, np.float32)
    write_bundle(str(tmp_path / 'norm'), t)
    with pytest.raises(NotImplementedError, match='gamma'):
        ckpt.reference_weights(str(tmp_path / 'norm'))
    t = dict(base)
    t['net/net_query_layer1/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE'] = np.zeros((2, 2, 32, 16), np.float32)
    write_bundle(str(tmp_path / 'half'), t)
    with pytest.raises(ValueError, match='net_query_layer1.*bias'):
        ckpt.reference_weights(str(tmp_path / 'half'))


def test_reader_against_a_checkpoint_written_by_tensorflow():
    """tests/golden/make_tf_golden.py (run where tensorflow==2.2.0 exists) saves `tf.train.Checkpoint(step, net)` over the
    reference's attribute names; the reader must find every variable under its key, bit for bit.  Skipped while the files are
    absent (no TensorFlow in the build image): until then the reader has only met this file's own writer."""
    import os
    ckdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_ckpt')
    exp_path = os.path.join(ckdir, 'expected.npz')
    if not os.path.exists(exp_path):
        pytest.skip("tests/golden/tf_ckpt/ not generated (needs tensorflow==2.2.0; see tests/golden/make_tf_golden.py)")
    exp = np.load(exp_path)
    prefix = os.path.join(ckdir, str(exp['prefix']))
    got = ckpt.read_bundle(prefix)
    keys = [k for k in exp.files if k != 'prefix']
    assert keys
    for k in keys:
        key = k.replace('|', '/')
        assert key in got, key
        np.testing.assert_array_equal(got[key], exp[k])
    w = ckpt.reference_weights(prefix)
    assert len(w['query']) == 4 and len(w['obs']) == 2 and len(w['query'][1]) == 2
