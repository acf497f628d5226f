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


def fresh_trainer(config, params=None):
    """new Context + Train + model; optionally load an oracle parameter dict."""
    import torch
    from tg import runtime
    from Training.Train_goodGAN import Train
    from Model.Good_GAN_cifar10 import Good_GAN_cifar10
    runtime.set_context(None)
    torch.cuda.empty_cache()
    tr = Train(config, None, None)
    tr._build_train_graph(Good_GAN_cifar10)
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
