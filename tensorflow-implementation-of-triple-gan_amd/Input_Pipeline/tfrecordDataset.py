"""TFRecord input pipeline — MI355X-native counterpart of the reference's Input_Pipeline/{cifar10,svhn,mnist}Dataset.py
(same constructor, method names and file naming; line references are to Input_Pipeline/svhnDataset.py unless noted).

  reference (TensorFlow)                                   here
  tf.data.TFRecordDataset(name)             :33-42         tg.io.RecordFile: mmap + index + CRC check in C++ (csrc/tfrecord.cpp)
  .map(self.parser, num_parallel_calls)     :44-70         tg_ds_gather (threaded tf.Example decode into pinned uint8 staging),
                                                           then ON THE DEVICE: x/255*2-1 (MNIST x/255) and one_hot (tg_u8_affine_f32,
                                                           tg_onehot_i32_f32) — 4x fewer bytes over PCIe than float32 batches
  .shuffle(MIN_QUEUE_EXAMPLES + 30*BATCH_SIZE).repeat(n)   :76-82   ShuffleStream: the same streaming-buffer algorithm on record indices
  .batch(bs).prefetch(5*bs)                 :84-87         Batcher + a background thread that stays `PREFETCH` batches ahead
  iterators / initializers                  :96-137        init_op_train() / init_op_val() restart the streams

Batch protocol (SURVEY §8a T1 — the reference's three pipelines disagree with its placeholders, the build fixes one):
three training streams — labelled for C (L_C), labelled for D (L_D), unlabelled (U_D + U_C, sliced as x_u[:U_D], x_u[U_D:]
Training/Train_goodGAN.py:255-256) — and the test split in batches of BATCH_SIZE.
"""
import os
import queue
import threading

import numpy as np

from tg import io as tgio


class ShuffleStream(object):
    """dataset.shuffle(buffer_size).repeat(count) on record indices 0..n-1: a buffer is filled with the next `buffer_size`
    indices in file order; every draw takes a uniformly random slot and refills it from the file; at the end of an epoch the
    buffer drains; `count` epochs (-1: endless)."""

    def __init__(self, n, buffer_size, repeat, seed):
        self.n, self.buffer_size, self.repeat = int(n), max(1, int(buffer_size)), repeat
        self.rng = np.random.default_rng(seed)

    def __iter__(self):
        epoch = 0
        while self.n > 0 and (self.repeat is None or self.repeat < 0 or epoch < self.repeat):
            nxt = min(self.buffer_size, self.n)
            buf = list(range(nxt))
            while buf:
                j = int(self.rng.integers(len(buf)))
                yield buf[j]
                if nxt < self.n:
                    buf[j] = nxt
                    nxt += 1
                else:
                    buf[j] = buf[-1]
                    buf.pop()
            epoch += 1


def _batches(stream, batch_size):
    """dataset.batch(batch_size): consecutive groups; only the very last one may be short."""
    cur = []
    for i in stream:
        cur.append(i)
        if len(cur) == batch_size:
            yield np.asarray(cur, np.int64)
            cur = []
    if cur:
        yield np.asarray(cur, np.int64)


class _Slot(object):
    """one staging buffer of the prefetch ring: uint8 images + int32 labels in PINNED host memory when a GPU is present (so the
    host-to-device copy is asynchronous), and the event after which the buffer may be refilled."""

    def __init__(self, max_n, shape, pinned):
        h, w, c = shape
        self.t_img = self.t_lab = None
        if pinned:
            import torch
            self.t_img = torch.empty(max_n * h * w * c, dtype=torch.uint8).pin_memory()
            self.t_lab = torch.empty(max_n, dtype=torch.int32).pin_memory()
            self.img, self.lab = self.t_img.numpy(), self.t_lab.numpy()
        else:
            self.img, self.lab = np.empty(max_n * h * w * c, np.uint8), np.empty(max_n, np.int32)
        self.shape, self.n, self.event = shape, 0, None

    @property
    def images(self):
        h, w, c = self.shape
        return self.img[:self.n * h * w * c].reshape(self.n, h, w, c)

    @property
    def labels(self):
        return self.lab[:self.n]


class _Prefetcher(object):
    """background thread: index batches -> decoded uint8 / int32 batches, `depth` ahead (dataset.prefetch).  The ring has
    depth + 2 slots: up to `depth` queued, one being filled, one in the consumer's hands; a slot whose copy to the device is still
    in flight (the host runs ahead of the GPU) is not refilled before that copy's event has completed."""

    def __init__(self, rec, index_batches, depth, n_threads, max_batch, pinned):
        depth = max(1, depth)
        self.q = queue.Queue(maxsize=depth)
        self.stop = False
        ring = [_Slot(max_batch, rec.shape, pinned) for _ in range(depth + 2)] if len(rec) else []

        def run():
            try:
                k = 0
                for idx in index_batches:
                    if self.stop:
                        return
                    slot = ring[k % len(ring)]
                    k += 1
                    if slot.event is not None:
                        slot.event.synchronize()
                        slot.event = None
                    slot.n = len(idx)
                    rec.gather(idx, slot.img, slot.lab, n_threads=n_threads)
                    self.q.put(slot)
                self.q.put(None)
            except Exception as e:          # surfaced in the consumer
                self.q.put(e)
        self.t = threading.Thread(target=run, daemon=True)
        self.t.start()

    def get(self):
        item = self.q.get()
        if isinstance(item, Exception):
            raise item
        return item

    def close(self):
        self.stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass


class _NNIO(object):
    """what Training/Train_goodGAN.py consumes: next() = the feed of one iteration, val_batches() = the test split."""

    def __init__(self, train, val):
        self.train, self.val = train, val

    def next(self):
        return self.train._next()

    def val_batches(self):
        return self.val._val_batches()


class tfrecordDataset(object):
    PREFIX = None
    TRAIN_SIZE = None
    CHANNELS = 3
    UNIT_RANGE = False          # True: x/255 (mnistDataset.py:65); False: x/255*2-1
    PREFETCH = 5                # batches (the reference prefetches batch_size*5 elements, :86)
    DECODE_THREADS = 1          # one thread decodes ~0.9 M CIFAR records/s; spawning helpers only pays for large batches

    def __init__(self, data_dir, config, num_label=None, subset='train', use_augmentation=False, seed=None):
        self.data_dir = os.path.join(data_dir, "Tfrecord")                       # :16
        self.subset = subset
        self.use_augmentation = use_augmentation
        self.config = config
        self.num_label = num_label
        self.train_size = self.TRAIN_SIZE
        self.seed = (1234 if seed is None else seed) + 1000 * getattr(config, 'RANK', 0)
        self._streams = None

    # ---- files ----------------------------------------------------------------------------------
    def get_filenames(self):                                                     # :23-31
        assert self.subset in ['train', 'test'], 'Invalid data subset "%s"' % self.subset
        if self.subset == 'train':
            return [os.path.join(self.data_dir, '%s_%s_%s.tfrecords' % (self.PREFIX, self.subset, str(self.num_label).zfill(6))),
                    os.path.join(self.data_dir, '%s_%s_%s.tfrecords' % (self.PREFIX, self.subset,
                                                                       str(self.train_size - self.num_label).zfill(6)))]
        return [os.path.join(self.data_dir, '%s_%s.tfrecords' % (self.PREFIX, self.subset))]

    def input_from_tfrecord_filename(self):                                      # :33-42
        names = self.get_filenames()
        if self.subset == 'test':
            return [tgio.RecordFile(n) for n in names]
        lab = tgio.RecordFile(names[0])
        return [lab, lab, tgio.RecordFile(names[1])]                              # labelled for D, labelled for C, unlabelled

    # ---- per-record parser (API parity and tests; the batched path decodes in C++ and scales on the device) -------------
    def parser(self, serialized_example):                                        # :44-70
        img, label, h, w = tgio.parse_example(serialized_example)
        image = np.frombuffer(img, np.uint8).reshape(h, w, self.CHANNELS).astype(np.float32)
        image = image / 255 if self.UNIT_RANGE else image / 255 * 2 - 1
        onehot = np.eye(self.config.NUM_CLASSES, dtype=np.float32)[label]
        if self.use_augmentation:
            image, onehot = self.pre_processing(image, onehot)
        return image, onehot

    def pre_processing(self, image, label):                                      # :72-74 (a no-op in the reference)
        return image, label

    def shuffle_and_repeat(self, dataset, repeat=1, seed=0):                     # :76-82
        return ShuffleStream(len(dataset), self.config.MIN_QUEUE_EXAMPLES + 30 * self.config.BATCH_SIZE, repeat, self.seed + seed)

    def batch(self, dataset, batch_size):                                        # :84-87
        return _batches(dataset, batch_size)

    # ---- device tail ----------------------------------------------------------------------------
    def _to_device(self, images_u8, labels_i32, want_labels=True, slot=None):
        """uint8 [n,H,W,C] / int32 [n] host arrays -> (Act float32 scaled, Act one-hot) on the GPU."""
        import torch
        from tg import lib
        from tg.runtime import Act, ctx
        cx = ctx()
        n, h, w, c = images_u8.shape
        xu = torch.from_numpy(images_u8).to(cx.device, non_blocking=True)
        x = torch.empty(n * h * w * c, dtype=torch.float32, device=cx.device)
        scale, shift = (1.0, 0.0) if self.UNIT_RANGE else (2.0, -1.0)
        lib.call('tg_u8_affine_f32', lib.ptr(xu), lib.ptr(x), xu.numel(), scale, shift, cx.stream)
        xa, ya = Act(x, n, h, w, c, c), None
        if want_labels:
            k = self.config.NUM_CLASSES
            lu = torch.from_numpy(labels_i32).to(cx.device, non_blocking=True)
            y = torch.empty(n * k, dtype=torch.float32, device=cx.device)
            lib.call('tg_onehot_i32_f32', lib.ptr(lu), lib.ptr(y), n, k, cx.stream)
            ya = Act(y, n, 1, 1, k, k)
        if slot is not None:                 # the staging buffer may be refilled once these copies have run
            slot.event = torch.cuda.Event()
            slot.event.record(torch.cuda.current_stream(cx.device))
        return xa, ya

    def _to_host(self, images_u8, labels_i32):
        """the same tail in NumPy (no GPU: tests, tools)."""
        x = images_u8.astype(np.float32)
        x = x / 255 if self.UNIT_RANGE else x / 255 * 2 - 1
        return x, np.eye(self.config.NUM_CLASSES, dtype=np.float32)[labels_i32]

    def _on_device(self):
        import torch
        return bool(getattr(self.config, 'PIPELINE_DEVICE', True)) and torch.cuda.is_available()

    def _finish(self, slot, want_labels=True):
        if self._on_device():
            return self._to_device(slot.images, slot.labels, want_labels, slot)
        return self._to_host(slot.images.copy(), slot.labels.copy())

    # ---- pipelines ------------------------------------------------------------------------------
    def _start_train(self):
        c = self.config
        lab_d, lab_c, unl = self._files
        if self._streams:
            for s in self._streams:
                s.close()
        mk = lambda rec, bs, repeat, seed: _Prefetcher(rec, self.batch(self.shuffle_and_repeat(rec, repeat, seed), bs), self.PREFETCH,
                                                       self.DECODE_THREADS, bs, self._on_device())
        self._streams = [mk(lab_c, c.BATCH_SIZE_L_C, -1, 1), mk(lab_d, c.BATCH_SIZE_L_D, -1, 2),
                         mk(unl, c.BATCH_SIZE_U_D + c.BATCH_SIZE_U_C, getattr(c, 'REPEAT', -1) or -1, 3)]
        self._epoch = getattr(self, '_epoch', 0) + 1

    def _next(self):
        c = self.config
        if self._streams is None:
            self._start_train()
        got = [s.get() for s in self._streams]
        if any(g is None for g in got) or got[2].n < c.BATCH_SIZE_U_D + c.BATCH_SIZE_U_C:
            raise StopIteration("unlabelled stream exhausted (config.REPEAT epochs)")
        b = {}
        b['x_l_c'], b['y_l_c'] = self._finish(got[0])
        b['x_l_d'], b['y_l_d'] = self._finish(got[1])
        xu, _ = self._finish(got[2], want_labels=False)
        if isinstance(xu, np.ndarray):
            b['x_u_d'], b['x_u_c'] = xu[:c.BATCH_SIZE_U_D], xu[c.BATCH_SIZE_U_D:]
        else:
            b['x_u_d'], b['x_u_c'] = xu.view_rows(0, c.BATCH_SIZE_U_D), xu.view_rows(c.BATCH_SIZE_U_D, xu.n)
        return b

    def _val_batches(self):
        rec = self._files[0]
        stream = self.batch(self.shuffle_and_repeat(rec, 1, 99), self.config.BATCH_SIZE)
        pf = _Prefetcher(rec, stream, self.PREFETCH, self.DECODE_THREADS, self.config.BATCH_SIZE, False)
        while True:
            got = pf.get()
            if got is None:
                return
            yield self._to_host(got.images.copy(), got.labels.copy())          # Train.evaluate stages host arrays itself

    def inputpipline_train_val(self, other):                                     # :89-137
        self._files = self.input_from_tfrecord_filename()
        other._files = other.input_from_tfrecord_filename()

        def init_op_train():
            self._start_train()

        def init_op_val():
            pass
        return init_op_train, init_op_val, _NNIO(self, other)

    def inputpipline_testSet(self):                                              # :139-160 (cifar10Dataset.py)
        self._files = self.input_from_tfrecord_filename()
        return (lambda: None), self._val_batches()
