"""ORACLE — test infrastructure only (see oracle/tf_ops.py header).

Pure-Python restatement of the on-disk format the reference's input pipelines read
(Input_Pipeline/cifar10Dataset.py:33-67, svhnDataset.py:33-70, mnistDataset.py:30-66):

  * a TFRecord file (tf.data.TFRecordDataset, cifar10Dataset.py:38) = a sequence of
        uint64 length | uint32 masked_crc32c(length bytes) | payload[length] | uint32 masked_crc32c(payload)
    little-endian, crc32c = CRC-32/Castagnoli (poly 0x1EDC6F41 reflected 0x82F63B78), and
        mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8   (mod 2^32);
  * each payload = a serialized tf.Example proto whose features map holds
        'image': bytes_list (raw uint8 HWC), 'label' / 'height' / 'width': int64_list   (cifar10Dataset.py:44-50);
  * parser: decode_raw uint8 -> reshape [height, width, C] -> float32 -> x/255*2-1 (MNIST: x/255, mnistDataset.py:65),
    label -> one_hot(NUM_CLASSES)   (cifar10Dataset.py:52-62).

TensorFlow itself (the library that defines this format) is not under /root/reference and not importable here; its
version is unpinned.  The format above is TensorFlow's published record / Example wire format.  PINNING: the CRC is
checked against the RFC 3720 B.4 known-answer vectors (tests/test_tfrecord.py); the proto encoding follows the
protobuf wire format (field numbers of example.proto / feature.proto cited below).  No TFRecord file written by
TensorFlow exists in the reference repository -> beyond those vectors this restatement is PARITY UNPINNED.
"""
import struct

import numpy as np

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)

MASK_DELTA = 0xa282ead8


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + MASK_DELTA) & 0xFFFFFFFF


# ---------------------------------------------------------------- protobuf wire format (varint / length-delimited)

def _varint(v):
    v &= (1 << 64) - 1                         # int64 negatives are 10-byte two's complement varints
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """features: dict name -> bytes (bytes_list of one value) or int (int64_list of one value), keys written in the given
    order.  example.proto: Example{features=1}; Features{map<string,Feature> feature=1} (map entry: key=1, value=2);
    Feature{bytes_list=1, float_list=2, int64_list=3}; BytesList{repeated bytes value=1}; Int64List{repeated int64 value=1
    [packed]}."""
    entries = b''
    for k, v in features.items():
        if isinstance(v, (bytes, bytearray)):
            feat = _ld(1, _ld(1, bytes(v)))
        else:
            feat = _ld(3, _ld(1, _varint(int(v))))               # packed int64
        entries += _ld(1, _ld(1, k.encode()) + _ld(2, feat))
    return _ld(1, entries)


def _fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield f, wt, v


def decode_example(buf):
    """-> dict name -> bytes | list of int | list of float (first-level decoding of every feature)."""
    out = {}
    for f, wt, feats in _fields(bytes(buf)):
        if f != 1:
            continue
        for f2, _, entry in _fields(feats):
            if f2 != 1:
                continue
            key, feat = None, b''
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = v.decode()
                elif f3 == 2:
                    feat = v
            for kind, _, lst in _fields(feat):
                if kind == 1:                                    # BytesList
                    vals = [v for f4, _, v in _fields(lst) if f4 == 1]
                    out[key] = vals[0] if len(vals) == 1 else vals
                elif kind == 3:                                  # Int64List: packed (wire type 2) or repeated varints
                    vals = []
                    for f4, wt4, v in _fields(lst):
                        if f4 != 1:
                            continue
                        if wt4 == 2:
                            p = 0
                            while p < len(v):
                                x, p = _read_varint(v, p)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
                    out[key] = vals
                elif kind == 2:                                  # FloatList (packed fixed32)
                    vals = []
                    for f4, wt4, v in _fields(lst):
                        if f4 == 1 and wt4 == 2:
                            vals += list(struct.unpack('<%df' % (len(v) // 4), v))
                        elif f4 == 1:
                            vals.append(struct.unpack('<f', v)[0])
                    out[key] = vals
    return out


# ---------------------------------------------------------------- record framing

def frame(payload):
    head = struct.pack('<Q', len(payload))
    return head + struct.pack('<I', masked_crc32c(head)) + payload + struct.pack('<I', masked_crc32c(payload))


def write_tfrecord(path, images_u8, labels):
    """images_u8 [N,H,W,C] uint8, labels [N] int -> the reference's file layout (one Example per image)."""
    images_u8 = np.ascontiguousarray(images_u8, np.uint8)
    n, h, w, _ = images_u8.shape
    with open(path, 'wb') as f:
        for i in range(n):
            f.write(frame(encode_example({'image': images_u8[i].tobytes(), 'label': int(labels[i]), 'height': h, 'width': w})))


def read_tfrecord(path):
    """-> list of payload bytes; raises ValueError on a CRC mismatch or a truncated record."""
    out = []
    with open(path, 'rb') as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        if pos + 12 > len(data):
            raise ValueError("truncated record header at byte %d" % pos)
        (n,) = struct.unpack_from('<Q', data, pos)
        (c,) = struct.unpack_from('<I', data, pos + 8)
        if c != masked_crc32c(data[pos:pos + 8]):
            raise ValueError("length CRC mismatch at byte %d" % pos)
        if pos + 12 + n + 4 > len(data):
            raise ValueError("truncated record payload at byte %d" % pos)
        payload = data[pos + 12:pos + 12 + n]
        (c,) = struct.unpack_from('<I', data, pos + 12 + n)
        if c != masked_crc32c(payload):
            raise ValueError("payload CRC mismatch at byte %d" % pos)
        out.append(payload)
        pos += 16 + n
    return out


def parse(payload, channels, num_classes, unit_range=False):
    """Input_Pipeline/cifar10Dataset.py:41-67 (unit_range: mnistDataset.py:65 x/255) -> (image float32 [H,W,C], one-hot)."""
    ex = decode_example(payload)
    h, w = ex['height'][0], ex['width'][0]
    img = np.frombuffer(ex['image'], np.uint8).reshape(h, w, channels).astype(np.float32)
    img = img / 255 if unit_range else img / 255 * 2 - 1
    return img, np.eye(num_classes, dtype=np.float32)[ex['label'][0]]
