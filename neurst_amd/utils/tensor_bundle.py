"""TensorFlow "tensor bundle" checkpoints (prefix.index + prefix.data-00000-of-00001) without TensorFlow.

This is the on-disk format of every checkpoint the reference writes (neurst/utils/checkpoints.py:94-183: a
tf.train.Checkpoint whose attributes are the model's variable names) -- reading it is what lets a model trained with
the reference continue here, writing it is the way back.

  prefix.index   a leveldb-format sorted table: key "" -> BundleHeaderProto, every other key -> BundleEntryProto
                 {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (masked CRC-32C of the tensor bytes)}
                 table = data blocks | metaindex block | index block | 48-byte footer (magic 0xdb4775248b80fb57);
                 block = prefix-compressed entries (shared, non_shared, value_len varints) + restart array, followed by
                 a 1-byte compression type and the masked CRC-32C of block + type
  prefix.data-*  the raw little-endian tensor bytes, back to back
  object-based checkpoints name a variable `v` of attribute `a` "<a with '/' -> '.S', '.' -> '..'>/.ATTRIBUTES/VARIABLE_VALUE"
  (neurst/utils/compat.py:152-155 undoes exactly that) and carry one string tensor "_CHECKPOINTABLE_OBJECT_GRAPH".

PARITY UNPINNED: the reference tree holds no TensorFlow-written checkpoint and TensorFlow is not installable here, so
this module is verified by round trips, by the format's own invariants (block / tensor CRCs, footer magic) and against
the public format description only (tests/test_checkpoints.py) -- not against bytes TensorFlow wrote.
"""
import os
import struct

import numpy as np

from neurst_amd.data.tfrecord import _fields, _ld, _read_varint, _write_varint, crc32c

_MASK_DELTA = 0xA282EAD8
_MAGIC = 0xDB4775248B80FB57
_RESTART_INTERVAL = 16
_BLOCK_BYTES = 4096
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"

# tensorflow/core/framework/types.proto
_DT_TO_NP = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
             9: np.dtype("<i8"), 10: np.dtype("bool"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
_NP_TO_DT = {v: k for k, v in _DT_TO_NP.items()}
DT_STRING, DT_BFLOAT16 = 7, 14


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


class BundleError(IOError):
    pass


# ------------------------------------------------------------------------------------------------ sorted table (index file)
def _parse_block(buf, offset, size, verify=True):
    content = bytes(buf[offset:offset + size])
    trailer = bytes(buf[offset + size:offset + size + 5])
    if len(content) < size or len(trailer) < 5:
        raise BundleError("truncated table block")
    if trailer[0] != 0:
        raise BundleError(f"compressed table block (type {trailer[0]}) is not supported")
    if verify and _mask(crc32c(content + trailer[:1])) != struct.unpack("<I", trailer[1:])[0]:
        raise BundleError("table block checksum mismatch")
    (n_restarts,) = struct.unpack("<I", content[-4:])
    end = len(content) - 4 * (n_restarts + 1)
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _read_varint(content, pos)
        non_shared, pos = _read_varint(content, pos)
        vlen, pos = _read_varint(content, pos)
        key = key[:shared] + content[pos:pos + non_shared]
        pos += non_shared
        out.append((key, content[pos:pos + vlen]))
        pos += vlen
    return out


def _read_table(path, verify=True):
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _MAGIC:
        raise BundleError(f"{path}: not a tensor-bundle index (bad footer magic)")
    footer = buf[-48:]
    _, p = _read_varint(footer, 0)        # metaindex handle
    _, p = _read_varint(footer, p)
    idx_off, p = _read_varint(footer, p)
    idx_size, p = _read_varint(footer, p)
    entries = []
    for _, handle in _parse_block(buf, idx_off, idx_size, verify):
        off, q = _read_varint(handle, 0)
        size, q = _read_varint(handle, q)
        entries.extend(_parse_block(buf, off, size, verify))
    return entries


def _build_block(items):
    out, restarts, prev = bytearray(), [], b""
    for i, (key, value) in enumerate(items):
        shared = 0
        if i % _RESTART_INTERVAL == 0:
            restarts.append(len(out))
        else:
            n = min(len(prev), len(key))
            while shared < n and prev[shared] == key[shared]:
                shared += 1
        out += _write_varint(shared) + _write_varint(len(key) - shared) + _write_varint(len(value)) + key[shared:] + value
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _write_table(path, items):
    """items: [(key bytes, value bytes)] in ascending key order."""
    out, index = bytearray(), []

    def emit(block):
        handle = _write_varint(len(out)) + _write_varint(len(block))
        out.extend(block + b"\x00" + struct.pack("<I", _mask(crc32c(block + b"\x00"))))
        return handle
    cur, cur_bytes = [], 0
    for key, value in items:
        cur.append((key, value))
        cur_bytes += len(key) + len(value) + 3
        if cur_bytes >= _BLOCK_BYTES:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index.append((cur[-1][0], emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index))
    footer = meta + idx
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))
    with open(path, "wb") as fp:
        fp.write(bytes(out))


# ------------------------------------------------------------------------------------------------ entries
def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
    for fno, wt, val in _fields(memoryview(buf)):
        if fno == 1 and wt == 0:
            e["dtype"] = val
        elif fno == 2 and wt == 2:
            for f2, w2, dim in _fields(val):
                if f2 == 2 and w2 == 2:
                    size = 0
                    for f3, w3, v3 in _fields(dim):
                        if f3 == 1 and w3 == 0:
                            size = v3
                    e["shape"].append(size)
        elif fno == 3 and wt == 0:
            e["shard_id"] = val
        elif fno == 4 and wt == 0:
            e["offset"] = val
        elif fno == 5 and wt == 0:
            e["size"] = val
        elif fno == 6 and wt == 5:
            e["crc32c"] = struct.unpack("<I", val)[0]
    return e


def _vint(fno, v):
    return _write_varint(fno << 3) + _write_varint(v)


def _encode_entry(dtype, shape, offset, size, crc):
    shp = b"".join(_ld(2, _vint(1, int(d)) if d else b"") for d in shape)
    out = _vint(1, dtype) + _ld(2, shp)
    if offset:
        out += _vint(4, offset)
    out += _vint(5, size) + _write_varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


def _decode_strings(raw, count):
    lens, pos = [], 0
    for _ in range(count):
        n, pos = _read_varint(raw, pos)
        lens.append(n)
    pos += 4  # masked checksum of the lengths
    out = []
    for n in lens:
        out.append(bytes(raw[pos:pos + n]))
        pos += n
    return out


def _encode_strings(values):
    """[varint64 length]* | fixed32 masked CRC of the lengths (as uint32 / uint64 words) | bytes.  Returns (payload, crc of
    the whole payload as tensor_bundle.cc accumulates it)."""
    head, crc = b"", 0
    for v in values:
        head += _write_varint(len(v))
        crc = crc32c(struct.pack("<I", len(v)) if len(v) <= 0xFFFFFFFF else struct.pack("<Q", len(v)), crc)
    cks = struct.pack("<I", _mask(crc))
    crc = crc32c(cks, crc)
    body = b"".join(values)
    crc = crc32c(body, crc)
    return head + cks + body, crc


def read_bundle(prefix, verify=True):
    """-> {key: numpy array (bfloat16 widened to float32) | list of bytes (string tensors)} of a tensor bundle."""
    entries = _read_table(prefix + ".index", verify)
    if not entries or entries[0][0] != b"":
        raise BundleError(f"{prefix}.index: missing bundle header")
    num_shards = 1
    for fno, wt, val in _fields(memoryview(entries[0][1])):
        if fno == 1 and wt == 0:
            num_shards = val
        if fno == 2 and wt == 0 and val != 0:
            raise BundleError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, value in entries[1:]:
        e = _parse_entry(value)
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = open(f"{prefix}.data-{sid:05d}-of-{num_shards:05d}", "rb")
        fp = shards[sid]
        fp.seek(e["offset"])
        raw = fp.read(e["size"])
        if len(raw) < e["size"]:
            raise BundleError(f"{prefix}: tensor {key!r} truncated")
        name = key.decode("utf-8")
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if e["dtype"] == DT_STRING:
            out[name] = _decode_strings(raw, count)
            continue
        if verify and e["crc32c"] is not None and _mask(crc32c(raw)) != e["crc32c"]:
            raise BundleError(f"{prefix}: tensor {name} checksum mismatch")
        if e["dtype"] == DT_BFLOAT16:
            arr = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
        elif e["dtype"] in _DT_TO_NP:
            arr = np.frombuffer(raw, dtype=_DT_TO_NP[e["dtype"]])
        else:
            raise BundleError(f"{prefix}: tensor {name} has unsupported dtype {e['dtype']}")
        out[name] = arr.reshape(e["shape"]).copy()
    for fp in shards.values():
        fp.close()
    return out


def write_bundle(prefix, tensors):
    """tensors: {key: numpy array | list of bytes (string tensor, stored as a vector; one element = scalar)}."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    header = _vint(1, 1) + _ld(3, _vint(1, 1))       # num_shards = 1, little endian (default), version.producer = 1
    items = [(b"", header)]
    offset = 0
    with open(f"{prefix}.data-00000-of-00001", "wb") as fp:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            val = tensors[name]
            if isinstance(val, (list, tuple)) and (not val or isinstance(val[0], (bytes, str))):
                vals = [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in val]
                raw, crc = _encode_strings(vals)
                entry = _encode_entry(DT_STRING, [] if len(vals) == 1 else [len(vals)], offset, len(raw), _mask(crc))
            else:
                arr = np.asarray(val)  # tobytes() below emits C order whatever the strides
                dt = arr.dtype.newbyteorder("<") if arr.dtype.byteorder == ">" else arr.dtype
                if np.dtype(dt) not in _NP_TO_DT:
                    raise TypeError(f"{name}: dtype {arr.dtype} cannot be stored")
                raw = arr.astype(dt, copy=False).tobytes()
                entry = _encode_entry(_NP_TO_DT[np.dtype(dt)], list(arr.shape), offset, len(raw), _mask(crc32c(raw)))
            fp.write(raw)
            items.append((name.encode("utf-8"), entry))
            offset += len(raw)
    _write_table(prefix + ".index", items)


# ------------------------------------------------------------------------------------------------ object-based naming
def escape_name(name):
    return name.replace(".", "..").replace("/", ".S")


def checkpoint_key(var_name):
    return escape_name(var_name) + _SUFFIX


def variable_name(key):
    """neurst/utils/compat.py:152-155 `wrapper_var_name` (plus the '..' -> '.' half of TensorFlow's escaping)."""
    name = key[:-len(_SUFFIX)] if key.endswith(_SUFFIX) else key
    out, i = "", 0
    while i < len(name):
        if name.startswith(".S", i):
            out, i = out + "/", i + 2
        elif name.startswith("..", i):
            out, i = out + ".", i + 2
        else:
            out, i = out + name[i], i + 1
    return out


def object_graph_proto(var_names):
    """TrackableObjectGraph of tf.train.Checkpoint(**{name: variable}): node 0 = the root with one child per variable
    (local_name = the variable's name), node i = a leaf with the single attribute VARIABLE_VALUE."""
    names = sorted(var_names)
    root = b"".join(_ld(1, _vint(1, i + 1) + _ld(2, n.encode("utf-8"))) for i, n in enumerate(names))
    nodes = _ld(1, root)
    for n in names:
        attr = _ld(1, b"VARIABLE_VALUE") + _ld(2, n.encode("utf-8")) + _ld(3, checkpoint_key(n).encode("utf-8"))
        nodes += _ld(1, _ld(2, attr))
    return nodes
