"""Shared scaffolding of the -m gpu parity tests: one Context per process, a CIFAR-10 config with the synthetic
ZCA of SURVEY §8d, and helpers that map the oracle's per-application randomness onto the batched HIP path."""
import numpy as np

from oracle import step_cifar10 as S

_STATE = {}


def make_config(sizes=None, **over):
    from config import Config
    s = dict(S.SIZES, **(sizes or {}))

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = "cifar10"
        DATA_DIR = "/nonexistent"
        NUM_LABEL = 4000
        BATCH_SIZE_G = s['B_G']
        BATCH_SIZE_L_C = s['L_C']
        BATCH_SIZE_U_C = s['U_C']
        BATCH_SIZE_L_D = s['L_D']
        BATCH_SIZE_U_D = s['U_D']
        BATCH_SIZE = s['B_G']
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = 32, 32, 3
        FAKE_G_LAMBDA = 0.3
        Z_DIM = 100
        NUM_CLASSES = 10
        LEARNING_RATE = 3e-4
        CLA_LEARNINIG_RATE = 3e-3
        EPOCHS = 1
        TRAIN_SIZE = 1000
        SUMMARY = False
        USE_HIP_GRAPH = False
        ZCA = zca()

    c = TempConfig()
    for k, v in over.items():
        setattr(c, k, v)
    return c


def zca():
    if 'zca' not in _STATE:
        _STATE['zca'] = S.synth_zca()
    return _STATE['zca']


def make_config_goodgan(data, sizes, **over):
    """MNIST / SVHN experiment configs of Training/Train_goodGAN.py:484-530,641-685 at the given batch sizes."""
    from config import Config
    hw, ch = (28, 1) if data == 'mnist' else (32, 3)

    class TempConfig(Config):
        NAME = "Good_GAN"
        DATA_NAME = data
        DATA_DIR = "/nonexistent"
        NUM_LABEL = 100 if data == 'mnist' else 1000
        BATCH_SIZE_G = sizes['B_G']
        BATCH_SIZE_L_C = sizes['L_C']
        BATCH_SIZE_U_C = sizes['U_C']
        BATCH_SIZE_L_D = sizes['L_D']
        BATCH_SIZE_U_D = sizes['U_D']
        BATCH_SIZE = sizes['B_G']
        IMAGE_HEIGHT, IMAGE_WIDTH, CHANNEL = hw, hw, ch
        FAKE_G_LAMBDA = 0.1
        Z_DIM = 100
        NUM_CLASSES = 10
        MINIBATCH_DIS = False
        LEARNING_RATE = 1e-3 if data == 'mnist' else 3e-4
        CLA_LEARNINIG_RATE = 3e-4
        EPOCHS = 1
        TRAIN_SIZE = 1000
        SUMMARY = False
        USE_HIP_GRAPH = False

    c = TempConfig()
    for k, v in over.items():
        setattr(c, k, v)
    return c


def fresh_trainer(config, params=None, Model=None):
    """new Context + Train + model; optionally load an oracle parameter dict."""
    import torch
    from tg import runtime
    from Training.Train_goodGAN import Train
    if Model is None:
        from Model.Good_GAN_cifar10 import Good_GAN_cifar10 as Model
    runtime.set_context(None)
    torch.cuda.empty_cache()
    tr = Train(config, None, None)
    tr._build_train_graph(Model)
    if params is not None:
        for st in tr.cx.stores.values():
            st.load_dict(params)
        tr.cx.stores['classifier'].ema.copy_(tr.cx.stores['classifier'].p)
    return tr


def cat_rnd(*parts):
    """concatenate the per-application rnd dicts of the oracle along the batch axis."""
    return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}


def injected_arrays(rnd):
    """oracle rnd of one iteration -> '<rng scope>/<name>' arrays in the HIP path's batching order."""
    out = {}
    for k, v in cat_rnd(rnd['D']['C_unl'], rnd['D']['C_unl_d']).items():
        out['D/C/' + k] = v
    for k, v in cat_rnd(rnd['D']['D_real'], rnd['D']['D_fake'], rnd['D']['D_unl']).items():
        out['D/D/' + k] = v
    for k, v in rnd['G']['D_fake'].items():
        out['G/D/' + k] = v
    for k, v in cat_rnd(rnd['C']['C_real'], rnd['C']['C_unl'], rnd['C']['C_unl_rep'], rnd['C']['C_fake']).items():
        out['C/C/' + k] = v
    for k, v in rnd['C']['D_unl'].items():
        out['C/D/' + k] = v
    return out


def rel_err(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30)


def injected_arrays_goodgan(rnd):
    """oracle rnd of one Good_GAN iteration -> '<rng scope>/<name>' arrays in the HIP path's batching order."""
    out = {}
    for k, v in cat_rnd(rnd['D']['C_unl'], rnd['D']['C_unl_d']).items():
        out['D/C/' + k] = v
    for k, v in cat_rnd(rnd['D']['D_real'], rnd['D']['D_fake'], rnd['D']['D_unl']).items():
        out['D/D/' + k] = v
    for k, v in rnd['G']['D_fake'].items():
        out['G/D/' + k] = v
    for k, v in cat_rnd(rnd['C']['C_real'], rnd['C']['C_unl'], rnd['C']['C_fake']).items():
        out['C/C/' + k] = v
    for k, v in rnd['C']['D_unl'].items():
        out['C/D/' + k] = v
    return out
