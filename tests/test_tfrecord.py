"""Input pipeline: the C++ TFRecord / tf.Example reader and writer behind include/tg_io.h against the
oracle's pure-Python restatement of the format (oracle/tfrecord.py) — bit-exact both ways — the CRC-32C known-answer
vectors of RFC 3720 B.4, corruption / truncation / ragged cases, and the host logic of the shuffle / repeat / batch
streams of Input_Pipeline/tfrecordDataset.py.  Host only (no GPU needed: the entry points never touch the device)."""
import os
import struct
import sys

import numpy as np
import pytest

from oracle import tfrecord as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-implementation-of-triple-gan_amd'))

RFC3720 = [(bytes(32), 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C),
           (b'123456789', 0xE3069283), (b'', 0)]


def tgio():
    from tg import io
    return io


def synth(n, h, w, c, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, h, w, c), dtype=np.uint8), rng.integers(0, 10, n).astype(np.int64)


def test_crc32c_known_answers():
    io = tgio()
    for data, want in RFC3720:
        assert O.crc32c(data) == want
        assert io.crc32c(data) == want
    rng = np.random.default_rng(1)
    for n in (1, 7, 8, 9, 63, 64, 65, 1000, 4097):
        for off in (0, 1, 3):                       # unaligned starts exercise the byte-wise head of the slicing-by-8 loop
            b = rng.integers(0, 256, n + off, dtype=np.uint8).tobytes()[off:]
            assert io.crc32c(b) == O.crc32c(b)
            assert io.masked_crc32c(b) == O.masked_crc32c(b)
    assert O.masked_crc32c(b'') == O.MASK_DELTA


def test_writer_and_oracle_produce_identical_files(tmp_path):
    io = tgio()
    img, lab = synth(9, 5, 7, 3)
    lab[3] = -1                                     # int64 negatives are ten-byte varints
    a, b = tmp_path / 'a.tfrecords', tmp_path / 'b.tfrecords'
    io.write_tfrecord(a, img, lab)
    O.write_tfrecord(b, img, lab)
    assert a.read_bytes() == b.read_bytes()
    io.write_tfrecord(a, img[:2], lab[:2], append=True)
    assert len(O.read_tfrecord(a)) == 11


@pytest.mark.parametrize("shape", [(6, 32, 32, 3), (4, 28, 28, 1), (1, 1, 1, 1)])
def test_reader_matches_oracle(tmp_path, shape):
    io = tgio()
    img, lab = synth(*shape, seed=3)
    p = tmp_path / 'x.tfrecords'
    O.write_tfrecord(p, img, lab)                   # written by the oracle, read by the C++ side
    rec = io.RecordFile(p)
    assert len(rec) == shape[0] and rec.shape == shape[1:]
    payloads = O.read_tfrecord(p)
    for i, pl in enumerate(payloads):
        assert rec.record(i) == pl
        ib, l, h, w = io.parse_example(pl)
        ex = O.decode_example(pl)
        assert ib == ex['image'] == img[i].tobytes() and [l] == ex['label'] and [h] == ex['height'] and [w] == ex['width']
    idx = np.array([shape[0] - 1, 0, 0] + list(range(shape[0])), np.int64)
    for nt in (1, 3):
        x, y = rec.gather(idx, n_threads=nt)
        np.testing.assert_array_equal(x, img[idx])
        np.testing.assert_array_equal(y, lab[idx])
    rec.close()


def test_example_variants_tensorflow_may_emit():
    """map entries in any order, unknown features, non-packed int64 lists, a float feature: all legal encodings of the same Example."""
    io = tgio()
    img = bytes(range(12))
    nonpacked = O._ld(3, O._varint((1 << 3) | 0) + O._varint(7))                       # Int64List.value as a plain varint field
    feats = [('width', O._ld(3, O._ld(1, O._varint(2)))), ('extra', O._ld(2, O._ld(1, struct.pack('<2f', 1.5, 2.5)))),
             ('label', nonpacked), ('image', O._ld(1, O._ld(1, img))), ('height', O._ld(3, O._ld(1, O._varint(2))))]
    entries = b''.join(O._ld(1, O._ld(1, k.encode()) + O._ld(2, f)) for k, f in feats)
    payload = O._ld(1, entries)
    assert io.parse_example(payload) == (img, 7, 2, 2)
    ex = O.decode_example(payload)
    assert ex['label'] == [7] and ex['extra'] == [1.5, 2.5] and ex['image'] == img
    from tg.lib import TgError
    missing = O._ld(1, b''.join(O._ld(1, O._ld(1, k.encode()) + O._ld(2, f)) for k, f in feats if k != 'label'))
    with pytest.raises(TgError, match="'label' is missing"):
        io.parse_example(missing)
    with pytest.raises(TgError, match="malformed"):
        io.parse_example(payload[:-3])


def test_corruption_truncation_and_empty(tmp_path):
    io = tgio()
    from tg.lib import TgError
    img, lab = synth(3, 4, 4, 3)
    p = tmp_path / 'x.tfrecords'
    O.write_tfrecord(p, img, lab)
    good = p.read_bytes()
    rec_len = len(good) // 3
    cases = {'payload CRC mismatch': (rec_len, good[:rec_len + 40] + bytes([good[rec_len + 40] ^ 1]) + good[rec_len + 41:]),
             'length CRC mismatch': (rec_len, good[:rec_len] + bytes([good[rec_len] ^ 4]) + good[rec_len + 1:]),
             'truncated record payload': (2 * rec_len, good[:-5]), 'truncated record header': (rec_len, good[:rec_len + 6])}
    for what, (at, data) in cases.items():
        q = tmp_path / 'bad.tfrecords'
        q.write_bytes(data)
        with pytest.raises(TgError, match=what + " at byte %d" % at):
            io.RecordFile(q)
        with pytest.raises(ValueError, match=what + " at byte %d" % at):
            O.read_tfrecord(q)
    e = tmp_path / 'empty.tfrecords'
    e.write_bytes(b'')
    r = io.RecordFile(e)
    assert len(r) == 0 and r.shape is None and O.read_tfrecord(e) == []
    with pytest.raises(TgError, match="cannot open"):
        io.RecordFile(tmp_path / 'nope.tfrecords')
    # ragged geometry: record 1 has another size than record 0
    O.write_tfrecord(p, img, lab)
    with open(p, 'ab') as f:
        f.write(O.frame(O.encode_example({'image': bytes(12), 'label': 1, 'height': 2, 'width': 2})))
    r = io.RecordFile(p)
    assert len(r) == 4
    with pytest.raises(TgError, match="record 3: geometry 2x2"):
        r.gather([0, 3])
    with pytest.raises(TgError, match="out of range"):
        r.gather([4])
    # crafted geometry in record 0: height * width wraps to 0 in 64 bits (must be rejected, not divided by), or exceeds the image
    for hh, ww in ((1 << 32, 1 << 32), (1 << 40, 1), (4, 5), (0, 4), (1 << 16, 1 << 16)):
        q = tmp_path / 'geom.tfrecords'
        q.write_bytes(O.frame(O.encode_example({'image': bytes(12), 'label': 1, 'height': hh, 'width': ww})))
        with pytest.raises(TgError, match="image bytes do not match height\\*width"):
            io.RecordFile(q)


def test_shuffle_repeat_batch_streams():
    from Input_Pipeline.tfrecordDataset import ShuffleStream, _batches
    n, buf = 57, 10
    draws = list(ShuffleStream(n, buf, 3, seed=5))
    assert len(draws) == 3 * n
    for e in range(3):
        ep = draws[e * n:(e + 1) * n]
        assert sorted(ep) == list(range(n))                        # every epoch is a permutation
        assert all(ep[i] < buf + i for i in range(n))              # element i can only come from the first buf+i records
    assert draws[:n] != draws[n:2 * n]
    assert list(ShuffleStream(n, buf, 3, seed=5)) == draws        # deterministic per seed
    assert list(ShuffleStream(5, 1, 2, seed=0)) == [0, 1, 2, 3, 4] * 2   # buffer 1 = file order
    assert list(ShuffleStream(0, 4, -1, seed=0)) == []
    it = iter(ShuffleStream(3, 8, -1, seed=1))                    # endless
    assert sorted(next(it) for _ in range(3)) == [0, 1, 2] and next(it) in (0, 1, 2)
    bs = list(_batches(ShuffleStream(n, buf, 2, seed=2), 25))      # batches run across the epoch boundary; only the last is short
    assert [len(b) for b in bs] == [25, 25, 25, 25, 14]


def _write_split(tmp_path, Dataset, cfg, n_lab, n_unl, n_test, hw, ch):
    """files named as the reference expects; pixel (0,0,0) of every image carries label*20 + 5 so batches can be checked."""
    d = tmp_path / 'Tfrecord'
    d.mkdir()
    rng = np.random.default_rng(0)

    def mk(n):
        lab = rng.integers(0, 10, n).astype(np.int64)
        img = rng.integers(0, 256, (n, hw, hw, ch), dtype=np.uint8)
        img[:, 0, 0, 0] = lab * 20 + 5
        return img, lab
    Dataset.TRAIN_SIZE = n_lab + n_unl
    tr = Dataset(str(tmp_path), cfg, n_lab, 'train', True)
    te = Dataset(str(tmp_path), cfg, n_lab, 'test', False)
    names = tr.get_filenames() + te.get_filenames()
    for name, n in zip(names, (n_lab, n_unl, n_test)):
        O.write_tfrecord(name, *mk(n))
    return tr, te, [os.path.basename(x) for x in names]


def test_dataset_protocol_on_host(tmp_path):
    from Input_Pipeline.cifar10Dataset import cifar10Dataset
    from config import Config

    class Cfg(Config):
        DATA_NAME = 'cifar10'
        BATCH_SIZE = BATCH_SIZE_G = 8
        BATCH_SIZE_L_C, BATCH_SIZE_U_C, BATCH_SIZE_L_D, BATCH_SIZE_U_D = 4, 4, 2, 6
        IMAGE_HEIGHT = IMAGE_WIDTH = 32
        CHANNEL = 3
        NUM_CLASSES = 10
        REPEAT = -1
        PIPELINE_DEVICE = False
    cfg = Cfg()
    train_size = cifar10Dataset.TRAIN_SIZE
    try:
        tr, te, names = _write_split(tmp_path, cifar10Dataset, cfg, 40, 110, 21, 32, 3)
        assert names == ['cifar10_train_000040.tfrecords', 'cifar10_train_000110.tfrecords', 'cifar10_test.tfrecords']
        init_train, init_val, nnio = tr.inputpipline_train_val(te)
        init_train()
        for _ in range(30):                                           # > one pass over both labelled and unlabelled files
            b = nnio.next()
            assert b['x_l_c'].shape == (4, 32, 32, 3) and b['y_l_c'].shape == (4, 10) and b['x_l_d'].shape == (2, 32, 32, 3)
            assert b['x_u_d'].shape == (6, 32, 32, 3) and b['x_u_c'].shape == (4, 32, 32, 3)
            for x, y in ((b['x_l_c'], b['y_l_c']), (b['x_l_d'], b['y_l_d'])):
                assert x.dtype == np.float32 and -1.0 <= x.min() and x.max() <= 1.0
                lab = np.rint(((x[:, 0, 0, 0] + 1) / 2 * 255 - 5) / 20).astype(int)    # undo x/255*2-1
                np.testing.assert_array_equal(y.argmax(1), lab)
        init_val()
        vb = list(nnio.val_batches())
        assert [x.shape[0] for x, _ in vb] == [8, 8, 5]                # the test split exactly once, last batch short
        # per-record parser (the reference's map function) agrees with the batched path and with the oracle's parser
        rec = tr._files[0]
        img, onehot = tr.parser(rec.record(3))
        oi, oo = O.parse(rec.record(3), 3, 10)
        np.testing.assert_array_equal(img, oi)
        np.testing.assert_array_equal(onehot, oo)
        xs, ys = tr._to_host(*rec.gather([3]))
        np.testing.assert_array_equal(xs[0], img)
        np.testing.assert_array_equal(ys[0], onehot)
    finally:
        cifar10Dataset.TRAIN_SIZE = train_size


def test_mnist_unit_range_and_names(tmp_path):
    from Input_Pipeline.mnistDataset import mnistDataset
    from Input_Pipeline.svhnDataset import svhnDataset
    from config import Config

    class Cfg(Config):
        DATA_NAME = 'mnist'
        BATCH_SIZE = 4
        NUM_CLASSES = 10
        IMAGE_HEIGHT = IMAGE_WIDTH = 28
        CHANNEL = 1
    ds = mnistDataset('/data', Cfg(), 100, 'train')
    assert [os.path.basename(n) for n in ds.get_filenames()] == ['mnist_train_000100.tfrecords', 'mnist_train_059900.tfrecords']
    assert os.path.basename(svhnDataset('/data', Cfg(), 1000, 'train').get_filenames()[1]) == 'svhn_train_072257.tfrecords'
    assert svhnDataset('/data', Cfg(), 1000, 'test').get_filenames() == ['/data/Tfrecord/svhn_test.tfrecords']
    img = np.arange(28 * 28, dtype=np.uint8).reshape(1, 28, 28, 1)
    p = tmp_path / 'm.tfrecords'
    O.write_tfrecord(p, img, [7])
    x, y = ds.parser(O.read_tfrecord(p)[0])
    np.testing.assert_array_equal(x, img[0].astype(np.float32) / 255)        # mnistDataset.py:65
    assert y.argmax() == 7
