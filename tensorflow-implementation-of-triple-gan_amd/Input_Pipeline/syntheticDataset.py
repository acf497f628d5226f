"""Synthetic CIFAR-10 / SVHN / MNIST-shaped data source with the reference's Dataset protocol
(Input_Pipeline/svhnDataset.py:89-137): `Dataset(data_dir, config, num_label, subset, use_augmentation)
.inputpipline_train_val(other) -> (init_op_train, init_op_val, NNIO)`.

The reference's TFRecord files are not part of its repository, so every config is measured on synthetic,
seeded class-prototype images in the dataset's value range.  The batch protocol is the consistent
one the build defines: three streams of sizes L_C, L_D and U_D+U_C per iteration, the unlabelled
one sliced as x_u[:U_D], x_u[U_D:U_D+U_C] (Training/Train_goodGAN.py:255-256).
"""
import numpy as np


class _NNIO(object):
    def __init__(self, train, val):
        self.train, self.val = train, val

    def next(self):
        return self.train._next()

    def val_batches(self):
        return self.val._val_batches()


class syntheticDataset(object):
    def __init__(self, data_dir, config, num_label, subset, use_augmentation=False, seed=1234, val_size=1000):
        self.config, self.subset, self.seed = config, subset, seed
        c = config
        lo = 0.0 if c.DATA_NAME == 'mnist' else -1.0
        self.lo = lo
        self.proto = np.random.default_rng(seed).uniform(lo, 1.0, (c.NUM_CLASSES,) + tuple(c.IMAGE_DIM)).astype(np.float32)
        self.val_size = val_size
        self.rng = np.random.default_rng(seed + (1 if subset == 'train' else 2) + 1000 * getattr(config, 'RANK', 0))

    def _imgs(self, n, rng=None):
        rng = self.rng if rng is None else rng
        y = rng.integers(0, self.config.NUM_CLASSES, n)
        x = np.clip(self.proto[y] + 0.25 * rng.standard_normal((n,) + tuple(self.config.IMAGE_DIM)), self.lo, 1.0)
        return x.astype(np.float32), np.eye(self.config.NUM_CLASSES, dtype=np.float32)[y]

    def _next(self):
        c = self.config
        b = {}
        b['x_l_c'], b['y_l_c'] = self._imgs(c.BATCH_SIZE_L_C)
        b['x_l_d'], b['y_l_d'] = self._imgs(c.BATCH_SIZE_L_D)
        xu, _ = self._imgs(c.BATCH_SIZE_U_D + c.BATCH_SIZE_U_C)
        b['x_u_d'], b['x_u_c'] = xu[:c.BATCH_SIZE_U_D], xu[c.BATCH_SIZE_U_D:c.BATCH_SIZE_U_D + c.BATCH_SIZE_U_C]
        return b

    def _val_batches(self):
        rng = np.random.default_rng(self.seed + 99)
        bs = self.config.BATCH_SIZE
        for _ in range(max(1, self.val_size // bs)):
            yield self._imgs(bs, rng)

    def inputpipline_train_val(self, other):
        def init_op_train():
            pass

        def init_op_val():
            pass
        return init_op_train, init_op_val, _NNIO(self, other)
