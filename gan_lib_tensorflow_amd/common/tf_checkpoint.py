"""TensorFlow `tf.train.Saver` (V2, "tensor bundle") checkpoints without TensorFlow: reader, writer, and the name map
onto this package's variable store.

Why: the reference publishes trained SNGAN weights as a Saver checkpoint (SNGAN/README.md:75-79) and restores with
`saver.restore` / `optimistic_restore` (SNGAN/gan_cifar_resnet.py:585-590; common/misc.py:275-307).  This package keeps
the reference's variable names, so importing such a checkpoint is a dictionary lookup once the container format is read.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*):
  <prefix>.index                 an SSTable (LevelDB table format, no compression): key "" -> BundleHeaderProto,
                                 key <tensor name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
  <prefix>.data-NNNNN-of-MMMMM   raw little-endian tensor bytes, addressed by (shard_id, offset, size)
SSTable: data blocks of prefix-compressed (shared, unshared, value_len, key_delta, value) entries + a restart array, each
block followed by a 5-byte trailer (compression type, masked CRC32C); an index block of (last key, BlockHandle) entries; a
48-byte footer (metaindex handle, index handle, padding, magic 0xdb4775248b80fb57).

The Inception classifier of the IS harness is NOT covered here: it ships as a frozen GraphDef download
(common/inception/inception_score.py:29-56), which this environment cannot fetch.
"""
import os
import struct
from collections import OrderedDict

import numpy as np

_MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 6: np.dtype('i1'), 9: np.dtype('<i8'),
           10: np.dtype('bool'), 19: np.dtype('<f2'), 14: np.dtype('<u2')}     # 14 = DT_BFLOAT16, returned as raw uint16
_DT_OF = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9, np.dtype('uint8'): 4}


# ---- CRC32C (Castagnoli), masked as LevelDB / TensorFlow store it ------------------------------------------------------
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC = _crc_table()


def _crc32c_bytes(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in data:
        crc = _CRC[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _gf2_times(mat, vec):
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _gf2_square(mat):
    return [_gf2_times(mat, mat[n]) for n in range(32)]


def _zeros_operator(nbytes):
    """the GF(2) matrix that advances a CRC-32C by `nbytes` zero bytes (zlib's crc32_combine construction)"""
    odd = [0x82F63B78] + [1 << (n - 1) for n in range(1, 32)]      # one zero bit
    even = _gf2_square(odd)                                       # two
    odd = _gf2_square(even)                                       # four
    op = [1 << n for n in range(32)]                              # identity
    n = nbytes
    while True:
        even = _gf2_square(odd)
        if n & 1:
            op = [_gf2_times(even, c) for c in op]
        n >>= 1
        if n == 0:
            break
        odd = _gf2_square(even)
        if n & 1:
            op = [_gf2_times(odd, c) for c in op]
        n >>= 1
        if n == 0:
            break
    return op


def crc32c(data):
    """CRC-32C of a bytes-like object.  Large buffers: K equal chunks advance in lock step as one numpy vector (the byte
    loop is sequential per chunk, parallel across chunks), then the chunk CRCs are chained with the zero-advance operator."""
    data = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    n = len(data)
    if n < (1 << 16):
        return _crc32c_bytes(data)
    k = 4096
    clen = n // k
    body = np.frombuffer(data, dtype=np.uint8, count=k * clen).reshape(k, clen)
    tab = np.asarray(_CRC, dtype=np.uint32)
    state = np.full(k, 0xFFFFFFFF, dtype=np.uint32)
    for i in range(clen):
        state = tab[(state ^ body[:, i]) & 0xFF] ^ (state >> 8)
    parts = (state ^ 0xFFFFFFFF).tolist()
    op = _zeros_operator(clen)
    crc = parts[0]
    for c in parts[1:]:
        crc = _gf2_times(op, crc) ^ c
    tail = data[k * clen:]
    if tail:
        crc = _gf2_times(_zeros_operator(len(tail)), crc) ^ _crc32c_bytes(tail)
    return crc


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / minimal protobuf ------------------------------------------------------------------------------------
def _get_varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf):
    """yield (field number, wire type, value) of a serialized message; value = int or bytes"""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _parse_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c)"""
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for fn, wt, v in _pb_fields(buf):
        if fn == 1:
            e['dtype'] = v
        elif fn == 2:                                     # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3
                    e['shape'].append(size)
        elif fn == 3:
            e['shard_id'] = v
        elif fn == 4:
            e['offset'] = v
        elif fn == 5:
            e['size'] = v
        elif fn == 6:
            e['crc32c'] = struct.unpack('<I', v)[0]
        elif fn == 7:
            e['sliced'] = True
    return e


def _ser_entry(dtype, shape, shard_id, offset, size, crc):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(int(s)) for s in shape))
    out = b'\x08' + _put_varint(dtype) + b'\x12' + _put_varint(len(dims)) + dims
    if shard_id:
        out += b'\x18' + _put_varint(shard_id)
    if offset:
        out += b'\x20' + _put_varint(offset)
    out += b'\x28' + _put_varint(size) + b'\x35' + struct.pack('<I', crc)
    return out


# ---- SSTable ----------------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify):
    data = buf[offset:offset + size]
    ctype = buf[offset + size]
    if ctype != 0:
        raise ValueError("compressed SSTable block (type %d): TensorFlow writes checkpoint indices uncompressed" % ctype)
    if verify:
        stored = struct.unpack('<I', buf[offset + size + 1:offset + size + 5])[0]
        if stored != _mask(crc32c(buf[offset:offset + size + 1])):
            raise ValueError("SSTable block checksum mismatch at offset %d" % offset)
    nrestart = struct.unpack('<I', data[-4:])[0]
    end = len(data) - 4 - 4 * nrestart
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _get_varint(data, pos)
        unshared, pos = _get_varint(data, pos)
        vlen, pos = _get_varint(data, pos)
        key = key[:shared] + data[pos:pos + unshared]
        pos += unshared
        out.append((key, data[pos:pos + vlen]))
        pos += vlen
    return out


def _read_table(path, verify=True):
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != _MAGIC:
        raise ValueError(f"{path} is not an SSTable (bad magic)")
    footer = buf[-48:]
    pos = 0
    _, pos = _get_varint(footer, pos)       # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries = []
    for _, handle in _read_block(buf, ioff, isize, verify):
        boff, p = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p)
        entries.extend(_read_block(buf, boff, bsize, verify))
    return entries


def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def _write_table(path, items, block_bytes=4096):
    """items: sorted [(key bytes, value bytes)]"""
    blob, index, cur, cur_size = bytearray(), [], [], 0

    def flush():
        nonlocal cur, cur_size
        if not cur:
            return
        blk = _build_block(cur)
        off = len(blob)
        blob.extend(blk + b'\x00' + struct.pack('<I', _mask(crc32c(blk + b'\x00'))))
        index.append((cur[-1][0], _put_varint(off) + _put_varint(len(blk))))
        cur, cur_size = [], 0
    for k, v in items:
        cur.append((k, v))
        cur_size += len(k) + len(v)
        if cur_size >= block_bytes:
            flush()
    flush()
    meta = _build_block([])
    moff = len(blob)
    blob.extend(meta + b'\x00' + struct.pack('<I', _mask(crc32c(meta + b'\x00'))))
    iblk = _build_block(index, restart_interval=1)
    ioff = len(blob)
    blob.extend(iblk + b'\x00' + struct.pack('<I', _mask(crc32c(iblk + b'\x00'))))
    footer = _put_varint(moff) + _put_varint(len(meta)) + _put_varint(ioff) + _put_varint(len(iblk))
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    blob.extend(footer)
    with open(path, 'wb') as f:
        f.write(bytes(blob))


# ---- public API ---------------------------------------------------------------------------------------------------------
def list_variables(prefix, verify=True):
    """[(name, shape, numpy dtype)] of a checkpoint `<prefix>.index` (tf.train.list_variables)"""
    out = []
    for k, v in _read_table(prefix + '.index', verify):
        if k == b'':
            continue
        e = _parse_entry(v)
        out.append((k.decode(), tuple(e['shape']), _DTYPES.get(e['dtype'])))
    return out


def read_checkpoint(prefix, names=None, verify=True):
    """name -> ndarray for every (or the requested) tensor of the checkpoint `<prefix>.index` + data shards."""
    entries = _read_table(prefix + '.index', verify)
    header = dict(num_shards=1)
    table = OrderedDict()
    for k, v in entries:
        if k == b'':
            for fn, _, val in _pb_fields(v):
                if fn == 1:
                    header['num_shards'] = val
                elif fn == 2 and val != 0:
                    raise ValueError("big-endian checkpoint")
            continue
        table[k.decode()] = _parse_entry(v)
    shards = {}
    out = OrderedDict()
    for name, e in table.items():
        if names is not None and name not in names:
            continue
        if e['sliced']:
            raise ValueError(f"{name}: partitioned variables are not supported")
        dt = _DTYPES.get(e['dtype'])
        if dt is None:
            raise ValueError(f"{name}: unsupported dtype enum {e['dtype']}")
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards']), 'rb').read()
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size'] or e['size'] != int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize:
            raise ValueError(f"{name}: size mismatch")
        if verify and e['crc32c'] is not None and _mask(crc32c(raw)) != e['crc32c']:
            raise ValueError(f"{name}: tensor checksum mismatch")
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
    return out


def write_checkpoint(prefix, tensors):
    """Write name -> ndarray (float32 / float64 / int32 / int64 / uint8) as a single-shard V2 checkpoint."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    data, items, off = bytearray(), [], 0
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        dt = _DT_OF.get(a.dtype)
        if dt is None:
            raise ValueError(f"{name}: dtype {a.dtype} not supported by the writer")
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes(order='C')        # (a scalar keeps its rank-0 shape)
        items.append((name.encode(), _ser_entry(dt, a.shape, 0, off, len(raw), _mask(crc32c(raw)))))
        data.extend(raw)
        off += len(raw)
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'      # num_shards = 1, endianness LITTLE (default), version { producer: 1 }
    _write_table(prefix + '.index', [(b'', header)] + items)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))


# ---- name map onto the variable store ------------------------------------------------------------------------------------
def trainer_state_from_checkpoint(ckpt, g_prefix='Generator', d_prefix='Discriminator', beta2=0.9):
    """A tf.train.Saver checkpoint of the reference's SNGAN script (name -> ndarray, e.g. from read_checkpoint) as the
    state dict `SNGANTrainer.load_state_dict` takes.  Variable names are identical by construction (store.py); what needs
    translating is the optimiser state: TF stores `<var>/Adam`, `<var>/Adam_1` (kept as is) and the scalars
    `beta1_power`, `beta2_power` (generator's optimiser, created first, :521-523) and `beta1_power_1`, `beta2_power_1`
    (critic's, :524-526): beta2_power = beta2 ** t gives the step count the Adam kernel keeps instead."""
    out = OrderedDict()
    for k, v in ckpt.items():
        if k in ('beta1_power', 'beta2_power', 'beta1_power_1', 'beta2_power_1'):
            continue
        out[k] = v
    for key, net in (('beta2_power', g_prefix), ('beta2_power_1', d_prefix)):
        if net + '/adam_t' in ckpt:          # written by checkpoint_from_trainer_state: the exact count
            out[net + '/adam_t'] = np.asarray(int(np.asarray(ckpt[net + '/adam_t']).reshape(-1)[0]), dtype=np.int64)
        elif key in ckpt:
            out[net + '/adam_t'] = np.asarray(adam_t_from_beta2_power(np.asarray(ckpt[key]).reshape(-1)[0], beta2), dtype=np.int64)
    return out


def adam_t_from_beta2_power(p, beta2=0.9):
    """Step count from TF's float32 `beta2_power` = beta2 ** t.  The power underflows to exactly 0.0 after ~980 steps at
    beta2 = 0.9 (under 200 iterations of the critic), and is a denormal with few significant bits before that: there the
    count cannot be recovered, and need not be -- the bias correction sqrt(1 - beta2^t) is 1.0f from t ~ 160 on, which is
    what TF itself computes from the stored 0.0.  Such a power maps to the SATURATED count (the smallest t whose power is
    below the smallest float32 denormal); only p == 1 (a fresh optimiser) maps to 0."""
    p = float(p)
    t_sat = int(np.ceil(np.log(1e-45) / np.log(beta2)))
    if p >= 1.0:
        return 0
    if p <= 1.2e-38:                  # zero, negative (corrupt) or denormal
        return t_sat
    return min(int(round(np.log(p) / np.log(beta2))), t_sat)


def checkpoint_from_trainer_state(state, g_prefix='Generator', d_prefix='Discriminator', beta1=0.0, beta2=0.9):
    """The inverse of trainer_state_from_checkpoint: `SNGANTrainer.state_dict()` as the tensors a tf.train.Saver of the
    reference's script would hold (write_checkpoint takes it from there).  Entries private to this package (iteration
    counter, device RNG state) are dropped: the reference feeds `_iteration` and seeds its RNG from outside the checkpoint."""
    out = OrderedDict()
    for k, v in state.items():
        if k.startswith('_') or k.endswith('/adam_t'):
            continue
        out[k] = np.asarray(v)
    for net, suffix in ((g_prefix, ''), (d_prefix, '_1')):
        if net + '/adam_t' in state:
            t = int(np.asarray(state[net + '/adam_t']))
            out['beta1_power' + suffix] = np.asarray(beta1 ** t if t > 0 else 1.0, dtype=np.float32)
            out['beta2_power' + suffix] = np.asarray(beta2 ** t, dtype=np.float32)       # 0.0f from t ~ 980 on, as in TF
            out[net + '/adam_t'] = np.asarray(t, dtype=np.int64)      # the exact count beside it (read back in preference)
    return out


def optimistic_restore(trainer_or_store, prefix, verify=True):
    """common/misc.py:275-307: restore every variable whose NAME and SHAPE match the checkpoint, ignore the rest.
    Accepts a ParamStore (weights only) or a trainer with load_state_dict (weights + optimiser state).  Returns the list
    of restored names."""
    ckpt = read_checkpoint(prefix, verify=verify)
    store = getattr(trainer_or_store, 'store', trainer_or_store)
    keep = OrderedDict()
    for k, v in trainer_state_from_checkpoint(ckpt).items():
        base = k[:-len('/Adam_1')] if k.endswith('/Adam_1') else k[:-len('/Adam')] if k.endswith('/Adam') else k
        if k.endswith('/adam_t'):
            keep[k] = v
        elif base in store.vars and tuple(store.vars[base].shape) == tuple(np.asarray(v).shape):
            keep[k] = v
    if hasattr(trainer_or_store, 'store'):
        trainer_or_store.load_state_dict(keep, strict=False)
    else:
        store.load_state_dict({k: v for k, v in keep.items() if k in store.vars}, strict=False)
    return [k for k in keep if k in store.vars]
