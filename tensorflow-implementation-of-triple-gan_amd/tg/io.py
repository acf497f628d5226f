"""ctypes binding of the host-side input pipeline entry points (include/tg_io.h): TFRecord files of tf.Example
records as the reference's Input_Pipeline/*Dataset.py read them.  Host only — usable without a GPU."""
import ctypes as C

import numpy as np

from . import lib

_typed = False


def _lib():
    global _typed
    l = lib.load()
    if not _typed:
        u8p, i64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        l.tg_crc32c.argtypes = [C.c_void_p, C.c_int64]
        l.tg_crc32c.restype = C.c_uint32
        l.tg_crc32c_masked.argtypes = [C.c_void_p, C.c_int64]
        l.tg_crc32c_masked.restype = C.c_uint32
        l.tg_tfrecord_write.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
        l.tg_record_append.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int]
        l.tg_example_parse.argtypes = [C.c_void_p, C.c_int64, C.POINTER(u8p), i64p, i64p, i64p, i64p]
        l.tg_ds_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        l.tg_ds_size.argtypes = [C.c_void_p]
        l.tg_ds_size.restype = C.c_int64
        l.tg_ds_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.tg_ds_record.argtypes = [C.c_void_p, C.c_int64, C.POINTER(u8p), i64p]
        l.tg_ds_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        l.tg_ds_close.argtypes = [C.c_void_p]
        for n in ('tg_tfrecord_write', 'tg_record_append', 'tg_example_parse', 'tg_ds_open', 'tg_ds_shape', 'tg_ds_record', 'tg_ds_gather', 'tg_ds_close'):
            getattr(l, n).restype = C.c_int
        _typed = True
    return l


def _check(rc, what):
    if rc != 0:
        raise lib.TgError("%s failed (%d): %s" % (what, rc, _lib().tg_last_error_string().decode()))


def crc32c(data):
    b = bytes(data)
    return int(_lib().tg_crc32c(b, len(b)))


def masked_crc32c(data):
    b = bytes(data)
    return int(_lib().tg_crc32c_masked(b, len(b)))


def write_tfrecord(path, images_u8, labels, append=False):
    """images_u8 [N,H,W,C] uint8, labels [N] -> one tf.Example {image,label,height,width} per record."""
    img = np.ascontiguousarray(images_u8, np.uint8)
    lab = np.ascontiguousarray(labels, np.int64)
    n, h, w, c = img.shape
    assert lab.shape == (n,)
    _check(_lib().tg_tfrecord_write(str(path).encode(), img.ctypes.data, lab.ctypes.data, n, h, w, c, int(append)), 'tg_tfrecord_write')


def append_record(path, payload, append=True):
    """frame one payload as a TFRecord record and append it to `path`."""
    b = bytes(payload)
    _check(_lib().tg_record_append(str(path).encode(), b, len(b), int(append)), 'tg_record_append')


def parse_example(payload):
    """serialized tf.Example -> (image bytes, label, height, width)  [tf.parse_single_example of the reference's parser]."""
    b = bytes(payload)
    img, il = C.POINTER(C.c_uint8)(), C.c_int64()
    lab, h, w = C.c_int64(), C.c_int64(), C.c_int64()
    _check(_lib().tg_example_parse(b, len(b), C.byref(img), C.byref(il), C.byref(lab), C.byref(h), C.byref(w)), 'tg_example_parse')
    return C.string_at(img, il.value), lab.value, h.value, w.value


class RecordFile(object):
    """tf.data.TFRecordDataset(name): memory-mapped, indexed, CRC-verified at open."""

    def __init__(self, path):
        self.path = str(path)
        self._h = C.c_void_p()
        _check(_lib().tg_ds_open(self.path.encode(), C.byref(self._h)), 'tg_ds_open')
        self.size = int(_lib().tg_ds_size(self._h))
        self.shape = None
        if self.size:
            h, w, c = C.c_int(), C.c_int(), C.c_int()
            _check(_lib().tg_ds_shape(self._h, C.byref(h), C.byref(w), C.byref(c)), 'tg_ds_shape')
            self.shape = (h.value, w.value, c.value)

    def __len__(self):
        return self.size

    def record(self, i):
        p, n = C.POINTER(C.c_uint8)(), C.c_int64()
        _check(_lib().tg_ds_record(self._h, int(i), C.byref(p), C.byref(n)), 'tg_ds_record')
        return C.string_at(p, n.value)

    def gather(self, idx, images_out=None, labels_out=None, n_threads=4):
        """records idx -> (uint8 [n,H,W,C], int32 [n]); the output arrays may be caller-provided (e.g. views of pinned memory)."""
        idx = np.ascontiguousarray(idx, np.int64)
        n = idx.size
        h, w, c = self.shape
        if images_out is None:
            images_out = np.empty((n, h, w, c), np.uint8)
        if labels_out is None:
            labels_out = np.empty(n, np.int32)
        assert images_out.dtype == np.uint8 and images_out.size >= n * h * w * c and images_out.flags['C_CONTIGUOUS']
        assert labels_out.dtype == np.int32 and labels_out.size >= n
        _check(_lib().tg_ds_gather(self._h, idx.ctypes.data, n, images_out.ctypes.data, labels_out.ctypes.data, int(n_threads)), 'tg_ds_gather')
        return images_out, labels_out

    def close(self):
        if self._h:
            _lib().tg_ds_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
