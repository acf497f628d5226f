#!/usr/bin/env python3
"""Input-pipeline throughput: TFRecord open (mmap + index + CRC-32C of every record), threaded tf.Example
decode into pinned staging, host-to-device copy + on-device scaling / one-hot.  Synthetic CIFAR-10-shaped records written
with tg_tfrecord_write.  Prints one JSON line.  The step consumes 200 real images per 100 nominal images: at the bench
rate (5 600 images/s/GPU, 8 GPUs) a node needs ~90 000 decoded images/s."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd")
for p in (ROOT, PKG):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def main():
    import torch
    from tg import io as tgio
    from config import Config
    from Input_Pipeline.cifar10Dataset import cifar10Dataset
    n_lab, n_unl, n_test = 4000, 46000, 1000
    rng = np.random.default_rng(0)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'Tfrecord'))

        class Cfg(Config):
            DATA_NAME = 'cifar10'
            BATCH_SIZE = BATCH_SIZE_G = 100
            BATCH_SIZE_L_C, BATCH_SIZE_U_C, BATCH_SIZE_L_D, BATCH_SIZE_U_D = 50, 50, 20, 80
            IMAGE_HEIGHT = IMAGE_WIDTH = 32
            CHANNEL = 3
            NUM_CLASSES = 10
            REPEAT = -1
            Z_DIM = 100
            NUM_LABEL = 4000
            FAKE_G_LAMBDA = 0.3
            CLA_LEARNINIG_RATE = 3e-3
            TRAIN_SIZE = 46000
            EPOCHS = 1
        cfg = Cfg()
        tr = cifar10Dataset(d, cfg, n_lab, 'train', True)
        te = cifar10Dataset(d, cfg, n_lab, 'test', False)
        t0 = time.perf_counter()
        for name, n in zip(tr.get_filenames() + te.get_filenames(), (n_lab, n_unl, n_test)):
            tgio.write_tfrecord(name, rng.integers(0, 256, (n, 32, 32, 3), dtype=np.uint8), rng.integers(0, 10, n))
        out['write_images_per_s'] = round((n_lab + n_unl + n_test) / (time.perf_counter() - t0))
        big = tr.get_filenames()[1]
        t0 = time.perf_counter()
        rec = tgio.RecordFile(big)
        dt = time.perf_counter() - t0
        out['open_index_crc'] = dict(records=len(rec), mbytes=round(os.path.getsize(big) / 1e6, 1), seconds=round(dt, 4),
                                     mb_per_s=round(os.path.getsize(big) / 1e6 / dt))
        idx = rng.integers(0, len(rec), 130 * 200)
        out['decode_images_per_s'] = {}
        for nt in (1, 4, 8):
            t0 = time.perf_counter()
            for i in range(200):
                rec.gather(idx[i * 130:(i + 1) * 130], n_threads=nt)
            out['decode_images_per_s']['threads_%d' % nt] = round(130 * 200 / (time.perf_counter() - t0))
        if torch.cuda.is_available():
            from tg.runtime import Context, set_context
            set_context(Context('cuda:0'))
            init_train, _, nnio = tr.inputpipline_train_val(te)
            init_train()
            for _ in range(5):
                nnio.next()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters = 300
            for _ in range(iters):
                nnio.next()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out['feeds_per_s'] = round(iters / dt, 1)
            out['real_images_per_s_into_hbm'] = round(iters * 200 / dt)
            out['nominal_images_per_s_sustained'] = round(iters * 100 / dt)
            # the training step fed from the files every iteration (PCIe-inclusive rate of the bench.py workload)
            from Training.Train_goodGAN import Train
            from Model.Good_GAN_cifar10 import Good_GAN_cifar10
            q, _ = np.linalg.qr(np.random.default_rng(4321).standard_normal((3072, 3072)))
            cfg.ZCA = (np.zeros(3072, np.float32), q.astype(np.float32))
            cfg.SUMMARY = False
            trn = Train(cfg, None, None)
            trn._build_train_graph(Good_GAN_cifar10)
            trn.set_hyper(3e-4, 3e-3, 0.3, 0.5)
            for _ in range(Train.AUTO_ITERS + 3 if getattr(cfg, 'EXEC_MODE', 'auto') == 'auto' else 3):      # past the execution-mode decision
                trn.feed(nnio.next()); trn.sample_latent(); trn.train_iteration()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 100
            for _ in range(steps):
                trn.feed(nnio.next()); trn.sample_latent(); trn.train_iteration()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out['train_from_tfrecords'] = dict(images_per_sec=round(steps * 100 / dt, 1), ms_per_step=round(dt / steps * 1e3, 3),
                                               exec_mode_chosen=trn.exec_mode_chosen()[0],
                                               note='CIFAR-10 config, fp32; every iteration decoded from TFRecord files, copied over PCIe (uint8) and scaled on the device')
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
