"""TFRecord files and tf.train.Example records without TensorFlow.

The reference feeds training from TFRecord shards (neurst/data/dataset_utils.py:256-326 `load_tfrecords`, :550-568
`take_one_record`; written by neurst/cli/create_tfrecords.py).  On-disk format (stable, public):

  record  := uint64 length (LE) | uint32 masked_crc32c(length bytes) | data | uint32 masked_crc32c(data)
  masked  := ((crc >> 15) | (crc << 17)) + 0xa282ead8   (mod 2**32), crc = CRC-32C (Castagnoli)

  tf.train.Example { Features features = 1 }      Features { map<string, Feature> feature = 1 }
  Feature { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 } }
  BytesList { repeated bytes value = 1 }   FloatList { repeated float value = 1 [packed] }
  Int64List { repeated int64 value = 1 [packed] }

The codec below is a hand-written protobuf wire reader / writer for exactly these messages (packed and unpacked
repeated scalars are both accepted, as protobuf requires).  tests/test_data_feed.py pins it on records TensorFlow itself
wrote (tests/golden/tfrecord_seq2seq_head.bin, cut from the reference's tests/examples/train.tfrecords-*) and against
google.protobuf's own serialisation of the same schema.
"""
import glob
import os
import struct

import numpy as np

_MASK_DELTA = 0xA282EAD8


def _make_table():
    poly = 0x82F63B78  # reflected Castagnoli polynomial
    table = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table[i] = c
    return table


_TABLE = _make_table()
_TABLE_LIST = [int(x) for x in _TABLE]


def crc32c_py(data, crc=0):
    """CRC-32C (iSCSI / RFC 3720 polynomial 0x1EDC6F41, reflected), byte-at-a-time table walk (~2 MB/s)."""
    c = crc ^ 0xFFFFFFFF
    t = _TABLE_LIST
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


_native_crc = None


def crc32c(data, crc=0):
    """CRC-32C through libneurst_hip.so's host routine nst_crc32c (slicing-by-8) when the library is built -- a 900-frame
    utterance record is 288 KB --, else the Python table walk."""
    global _native_crc
    if _native_crc is None:
        try:
            from neurst_amd import _lib
            _native_crc = _lib.lib.nst_crc32c
        except Exception:  # library not built: host-only use of the codec still works
            _native_crc = False
    if _native_crc:
        data = bytes(data)
        return int(_native_crc(data, len(data), crc))
    return crc32c_py(data, crc)


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


class TFRecordError(IOError):
    pass


def read_records(path, check_crc=True):
    """Yields the payload bytes of every record of one TFRecord file, in file order."""
    with open(path, "rb") as fp:
        while True:
            head = fp.read(12)
            if not head:
                return
            if len(head) < 12:
                raise TFRecordError(f"{path}: truncated record header")
            (length,), (len_crc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if check_crc and masked_crc32c(head[:8]) != len_crc:
                raise TFRecordError(f"{path}: corrupted record length")
            data = fp.read(length)
            tail = fp.read(4)
            if len(data) < length or len(tail) < 4:
                raise TFRecordError(f"{path}: truncated record")
            if check_crc and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise TFRecordError(f"{path}: corrupted record data")
            yield data


def frame_record(data):
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data))


def write_records(path, records):
    with open(path, "wb") as fp:
        for r in records:
            fp.write(frame_record(bytes(r)))


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _read_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _write_varint(v):
    v &= 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    """Yields (field_number, wire_type, value) of one message; value = int (varint / fixed) or memoryview slice (len-delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fno, wt, val


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_feature(buf):
    """-> ("bytes", [bytes...]) | ("float", float32 array) | ("int64", int64 array) | (None, [])"""
    for fno, wt, val in _fields(buf):
        if wt != 2:
            continue
        if fno == 1:
            return "bytes", [bytes(v) for f, w, v in _fields(val) if f == 1 and w == 2]
        if fno == 2:
            parts = []
            for f, w, v in _fields(val):
                if f != 1:
                    continue
                parts.append(np.frombuffer(bytes(v), dtype="<f4"))  # packed run or one unpacked fixed32
            return "float", (np.concatenate(parts) if parts else np.zeros(0, np.float32))
        if fno == 3:
            vals = []
            for f, w, v in _fields(val):
                if f != 1:
                    continue
                if w == 0:
                    vals.append(_signed64(v))
                else:
                    p, n = 0, len(v)
                    while p < n:
                        x, p = _read_varint(v, p)
                        vals.append(_signed64(x))
            return "int64", np.asarray(vals, dtype=np.int64)
    return None, []


def parse_example(record):
    """tf.train.Example bytes -> {feature name: ("bytes"|"float"|"int64", values)}."""
    out = {}
    buf = memoryview(record)
    for fno, wt, features in _fields(buf):
        if fno != 1 or wt != 2:
            continue
        for f2, w2, entry in _fields(features):
            if f2 != 1 or w2 != 2:
                continue
            key, value = None, (None, [])
            for f3, w3, v3 in _fields(entry):  # map entry: key = 1, value = 2
                if f3 == 1 and w3 == 2:
                    key = bytes(v3).decode("utf-8")
                elif f3 == 2 and w3 == 2:
                    value = _parse_feature(v3)
            if key is not None:
                out[key] = value
    return out


def _ld(fno, payload):
    return _write_varint((fno << 3) | 2) + _write_varint(len(payload)) + payload


def encode_example(features):
    """{name: list of bytes/str | float array | int array} -> tf.train.Example bytes (map entries in sorted key order,
    packed numeric lists, as TensorFlow's python protobuf writes them)."""
    body = b""
    for key in sorted(features):
        val = features[key]
        if isinstance(val, (bytes, str)):
            val = [val]
        arr = val if isinstance(val, (list, tuple)) and val and isinstance(val[0], (bytes, str)) else np.asarray(val)
        if isinstance(arr, (list, tuple)):
            inner = b"".join(_ld(1, v.encode("utf-8") if isinstance(v, str) else bytes(v)) for v in arr)
            feat = _ld(1, inner)
        elif arr.dtype.kind == "f":
            feat = _ld(2, _ld(1, arr.astype("<f4").tobytes()) if arr.size else b"")
        elif arr.dtype.kind in "iub":
            feat = _ld(3, _ld(1, b"".join(_write_varint(int(x)) for x in arr.reshape(-1))) if arr.size else b"")
        else:
            raise TypeError(f"feature {key}: unsupported value type {arr.dtype}")
        body += _ld(1, _ld(1, key.encode("utf-8")) + _ld(2, feat))
    return _ld(1, body)


# ------------------------------------------------------------------------------------------------ file sets
def flatten_string_list(x):
    """neurst/utils/misc.py flatten_string_list: 'a,b' / ['a', 'b,c'] -> ['a', 'b', 'c']."""
    if x is None:
        return None
    if isinstance(x, str):
        return [s.strip() for s in x.strip().split(",") if s.strip()]
    out = []
    for y in x:
        out.extend(flatten_string_list(y))
    return out


def list_record_files(file_path):
    """File set of `load_tfrecords` (dataset_utils.py:284-294): a directory stands for dir/*train*, an existing file for
    itself, anything else for the prefix pattern path*; every pattern lists in sorted order (Dataset.list_files,
    shuffle=False)."""
    files = []
    for f in flatten_string_list(file_path):
        if os.path.isdir(f):
            pattern = os.path.join(f, "*train*")
        elif os.path.exists(f):
            pattern = f
        else:
            pattern = f + "*"
        files.extend(sorted(glob.glob(pattern)))
    return files


def interleave_records(files, cycle_length=10, check_crc=True):
    """Deterministic order of Dataset.interleave(TFRecordDataset, cycle_length=10, block_length=1): one record from each of
    the (up to) `cycle_length` open files in turn; an exhausted file's slot is refilled with the next unopened file."""
    pending = list(files)
    slots = []
    while pending and len(slots) < cycle_length:
        slots.append(read_records(pending.pop(0), check_crc))
    i = 0
    while slots:
        if i >= len(slots):
            i = 0
        try:
            yield next(slots[i])
            i += 1
        except StopIteration:
            if pending:
                slots[i] = read_records(pending.pop(0), check_crc)
            else:
                slots.pop(i)
