"""Reading the reference's TensorFlow checkpoints (`ckpt-N.index` + `ckpt-N.data-00000-of-00001`) without TensorFlow.

The reference saves `tf.train.Checkpoint(step=..., optimizer=..., net=model)` (nlt/trainvali.py:134-141) and restores
`tf.train.Checkpoint(net=model)` for inference (nlt/nlt_test.py:68-73); the model's trainable layers are reachable as
`net_{query,obs}_layer{i}` attributes (nlt/models/base.py:79-101), so the variables are stored under
    net/net_query_layer0/kernel/.ATTRIBUTES/VARIABLE_VALUE                        (plain Conv2D: L0, head)
    net/net_query_layer3/layer_with_weights-1/bias/.ATTRIBUTES/VARIABLE_VALUE      (Keras Sequential blocks)
This module parses the on-disk "tensor bundle" format -- an SSTable index in TensorFlow's copy of the LevelDB table
format whose values are BundleEntryProto messages, plus raw little-endian tensor bytes in the data shard(s) -- and
maps those keys onto `Model.load_weights`' structure (Keras array layouts are kept as they are).

STATUS: format restated from the published layouts (tensorflow/core/lib/io/table_format.txt,
tensorflow/core/protobuf/tensor_bundle.proto); no TensorFlow is installable here, so it is exercised only against
bundles written by the test-side writer in tests/test_host_ckpt.py (round trip), not against a real TF file.
Snappy-compressed index blocks (TF writes the bundle index uncompressed) are rejected loudly.
"""
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_, 14: None, 19: np.float16}   # 14: bfloat16

_CRC_TABLE = None


def _table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    return _CRC_TABLE


def _crc_raw(data, c):
    """The table-driven register update over `data` from register value c (no initial / final inversion)."""
    tab = _table()
    for b in data:
        c = tab[(c ^ b) & 0xff] ^ (c >> 8)
    return c


def _zeros_operator(nbytes):
    """The GF(2)-linear map "feed nbytes zero bytes" on the 32-bit register, as its 32 column images."""
    tab = _table()
    cols = [tab[(1 << i) & 0xff] ^ ((1 << i) >> 8) for i in range(32)]        # one zero byte
    apply_ = lambda m, v: _xor_cols(m, v)
    result = [1 << i for i in range(32)]                                       # identity
    while nbytes:
        if nbytes & 1:
            result = [apply_(cols, v) for v in result]
        cols = [apply_(cols, v) for v in cols]
        nbytes >>= 1
    return result


def _xor_cols(cols, v):
    out, i = 0, 0
    while v:
        if v & 1:
            out ^= cols[i]
        v >>= 1
        i += 1
    return out


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of table blocks and bundle entries.  Large buffers are cut into equal chunks
    whose registers advance together in NumPy (the update is linear over GF(2), so chunk results combine exactly)."""
    data = bytes(data)
    n = len(data)
    if n < (1 << 12) or crc:
        return _crc_raw(data, crc ^ 0xffffffff) ^ 0xffffffff
    chunks = max(64, min(4096, n >> 7))
    m = -(-n // chunks)
    buf = np.zeros(chunks * m, np.uint8)
    buf[chunks * m - n:] = np.frombuffer(data, np.uint8)                       # leading zeros leave a zero register at zero
    cols = buf.reshape(chunks, m)
    tab = np.array(_table(), np.uint32)
    state = np.zeros(chunks, np.uint32)
    for i in range(m):
        state = tab[(state ^ cols[:, i]) & 0xff] ^ (state >> 8)
    zm = _zeros_operator(m)
    total = 0
    for r in state.tolist():
        total = _xor_cols(zm, total) ^ r                                       # register(A || B) = Z_len(B)(register(A)) ^ register(B)
    init = _xor_cols(_zeros_operator(n), 0xffffffff)                          # what the 0xffffffff initial value turns into
    return (total ^ init) ^ 0xffffffff


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block(buf, offset, size, verify=True):
    """Entries (key, value) of one table block at [offset, offset + size); the 5-byte trailer follows it."""
    data = buf[offset:offset + size]
    ctype = buf[offset + size]
    if ctype != 0:
        raise NotImplementedError("compressed table block (type %d): the bundle index is expected uncompressed" % ctype)
    if verify:
        stored, = struct.unpack_from('<I', buf, offset + size + 1)
        if stored != masked_crc(bytes(data) + bytes([ctype])):
            raise ValueError("table block checksum mismatch at offset %d" % offset)
    num_restarts, = struct.unpack_from('<I', data, len(data) - 4)
    end = len(data) - 4 - 4 * num_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _varint(data, pos)
        unshared, pos = _varint(data, pos)
        vlen, pos = _varint(data, pos)
        key = key[:shared] + bytes(data[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(data[pos:pos + vlen])))
        pos += vlen
    return out


def _proto(buf):
    """Minimal protobuf wire decoding: {field number: [values]}; varints as int, length-delimited as bytes."""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _entry(value):
    """BundleEntryProto -> (dtype enum, shape, shard_id, offset, size, crc32c or None)."""
    m = _proto(value)
    shape = []
    for sh in m.get(2, []):                                  # TensorShapeProto
        for dim in _proto(sh).get(2, []):                    # repeated Dim
            shape.append(_proto(dim).get(1, [0])[0])
    if 7 in m:
        raise NotImplementedError("sliced (partitioned) variables")
    return (m.get(1, [0])[0], tuple(shape), m.get(3, [0])[0], m.get(4, [0])[0], m.get(5, [0])[0],
            m[6][0] if 6 in m else None)


def read_bundle(prefix, verify=True):
    """{variable key: numpy array} of every numeric tensor in the checkpoint `prefix` (e.g. '.../checkpoints/ckpt-100')."""
    index = memoryview(open(prefix + '.index', 'rb').read())
    if len(index) < 48 or struct.unpack_from('<Q', index, len(index) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s.index is not a TensorFlow table file" % prefix)
    foot = len(index) - 48
    _, pos = _varint(index, foot)                            # metaindex handle: offset, size (unused)
    _, pos = _varint(index, pos)
    ioff, pos = _varint(index, pos)
    isize, pos = _varint(index, pos)
    entries = []
    for _, handle in _block(index, ioff, isize, verify):
        off, p = _varint(handle, 0)
        size, _ = _varint(handle, p)
        entries += _block(index, off, size, verify)
    header = dict(entries).get(b'')
    num_shards = _proto(header).get(1, [1])[0] if header else 1
    if header and _proto(header).get(2, [0])[0] != 0:
        raise NotImplementedError("big-endian bundle")
    shards = {}
    out = {}
    for key, value in entries:
        if key == b'':
            continue
        dtype, shape, shard, offset, size, crc = _entry(value)
        np_dtype = DTYPES.get(dtype)
        if np_dtype is None:
            continue                                         # strings (the object graph proto), variants, ...
        if shard not in shards:
            shards[shard] = open('%s.data-%05d-of-%05d' % (prefix, shard, num_shards), 'rb').read()
        raw = shards[shard][offset:offset + size]
        if len(raw) != size:
            raise ValueError("%s: data shard too short for %r" % (prefix, key))
        if verify and crc is not None and masked_crc(raw) != crc:
            raise ValueError("%s: checksum mismatch for %r" % (prefix, key))
        out[key.decode()] = np.frombuffer(raw, dtype=np_dtype).reshape(shape).copy()
    return out


_KEY = re.compile(r'^net/net_(query|obs)_layer(\d+)/(?:layer_with_weights-(\d+)/)?(kernel|bias)/\.ATTRIBUTES/VARIABLE_VALUE$')


def reference_weights(prefix, verify=True):
    """Checkpoint -> {'query': [[(kernel, bias), ...] per layer], 'obs': [...]} as `Model.load_weights` takes it
    (Keras layouts: Conv2D (kh,kw,Cin,Cout), Conv2DTranspose (kh,kw,Cout,Cin))."""
    found, other = {}, []
    for key, arr in read_bundle(prefix, verify).items():
        m = _KEY.match(key)
        if m:
            net, layer, conv, what = m.group(1), int(m.group(2)), int(m.group(3) or 0), m.group(4)
            found.setdefault(net, {}).setdefault(layer, {}).setdefault(conv, {})[what] = arr
        elif key.startswith('net/') and key.endswith('/.ATTRIBUTES/VARIABLE_VALUE') and '/.OPTIMIZER_SLOT/' not in key:
            other.append(key)
    if other:
        # norm layers (gamma / beta / moving_*) would shift the layer_with_weights-N numbering: refuse instead of misaligning
        raise NotImplementedError("%s: the checkpoint holds network variables that are not conv kernels / biases (%s%s); only "
                                  "the norm = None branch can be imported" % (prefix, other[0], ', ...' if len(other) > 1 else ''))
    if 'query' not in found:
        raise ValueError("%s: no net/net_query_layer*/... variables (is this an NLT checkpoint?)" % prefix)
    out = {}
    for net, layers in found.items():
        out[net] = []
        for li in range(max(layers) + 1):
            convs = layers.get(li)
            if convs is None:
                raise ValueError("%s: %s layer %d has no variables" % (prefix, net, li))
            for c in sorted(convs):
                missing = [w for w in ('kernel', 'bias') if w not in convs[c]]
                if missing:
                    raise ValueError("%s: net_%s_layer%d (conv %d) has no %s variable" % (prefix, net, li, c, ' / '.join(missing)))
            out[net].append([(convs[c]['kernel'], convs[c]['bias']) for c in sorted(convs)])
    out.setdefault('obs', [])
    return out


def load_reference_checkpoint(model, prefix, verify=True):
    """Loads a reference checkpoint into a built nlt_amd model (same architecture keys in its config)."""
    return model.load_weights(reference_weights(prefix, verify))
