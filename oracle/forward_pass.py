"""ORACLE — test infrastructure only (see oracle/tf_ops.py header; PARITY UNPINNED).

CPU restatement of the reference's whole-graph route — the one its `_build_train_graph` builds (Training/Train_goodGAN.py:400-426):

    G, D, C = Model.forward_pass(z_g, y_g, x_l_c, y_l_c, x_l_d, y_l_d, x_u_d, x_u_c, train)      Model/Good_GAN_cifar10.py:204-278
                                                                                                  Model/Good_GAN.py:428-472
    d_loss, g_loss, c_loss = Train_base._loss_GAN(D, C, [y_g, y_l_c], Lambda)                     Training/train_base.py:113-154

and of the helper methods `_loss_GAN` is written with (train_base.py:43-57,75-84).  Every network application gets its own dropout /
noise draw (`rnd[<application>]`), as separate TF ops would; running statistics are updated in call-site order.
"""
import numpy as np

from . import nets_cifar10 as N
from . import nets_goodgan as NG
from . import tf_ops as T


def forward_pass_cifar10(P, b, rnd, zca, train=True, moving=None):
    """Model/Good_GAN_cifar10.py:204-278.  rnd keys: C_real, C_unl, C_unl_rep, C_unl_d, C_fake (classifier applications, call-site order
    :228-240), D_real, D_fake, D_unl.  Returns ([G, D(6), C(5)], pop_mean updates) — D = [sigmoid, logits] x (real, fake, unl).
    moving: dict whose generator batch-norm moving statistics this execution updates (nets_cifar10.generator_fwd)."""
    G, _ = N.generator_fwd(P, b['z_g'], b['y_g'], moving=moving)                                    # :216-217
    pops = {}
    x_u_c_z = N.zca_apply(b['x_u_c'], *zca)                                                       # :221-225
    C_real, _, _ = N.classifier_fwd(P, N.zca_apply(b['x_l_c'], *zca), train, rnd['C_real'], pops)   # :228
    C_unl, _, _ = N.classifier_fwd(P, x_u_c_z, train, rnd['C_unl'], pops)                           # :231
    C_rep, _, _ = N.classifier_fwd(P, x_u_c_z, train, rnd['C_unl_rep'], pops)                       # :233
    C_unl_d, _, _ = N.classifier_fwd(P, N.zca_apply(b['x_u_d'], *zca), train, rnd['C_unl_d'], pops)  # :236
    C_fake, _, _ = N.classifier_fwd(P, N.zca_apply(G, *zca), train, rnd['C_fake'], pops)            # :240
    X_P = np.concatenate([b['x_l_d'], b['x_u_d']], axis=0)                                        # :258-259
    Y_P = np.concatenate([b['y_l_d'], T.argmax_onehot(C_unl_d)], axis=0)
    D_real_l, _ = N.discriminator_fwd(P, X_P, Y_P, rnd['D_real'])                                 # :264
    D_fake_l, _ = N.discriminator_fwd(P, G, b['y_g'], rnd['D_fake'])                              # :267
    D_unl_l, _ = N.discriminator_fwd(P, b['x_u_c'], T.argmax_onehot(C_unl), rnd['D_unl'])         # :270-271
    D = [T.sigmoid(D_real_l), D_real_l, T.sigmoid(D_fake_l), D_fake_l, T.sigmoid(D_unl_l), D_unl_l]
    return [G, D, [C_real, C_unl, C_unl_d, C_fake, C_rep]], pops


def forward_pass_goodgan(P, data, b, rnd, train=True):
    """Model/Good_GAN.py:428-472 (MNIST / SVHN): no ZCA, four classifier outputs (C_real, C_unl, C_unl_d, C_fake)."""
    CL, DL = NG.classifier_layers(data), NG.discriminator_layers(data)
    bnu = {}
    G, _, _ = NG.seq_fwd(P, NG.generator_layers(data), b['z_g'], b['y_g'], {}, True, bnu)
    Gimg = G.reshape((-1,) + NG.image_shape(data))
    C_real, _, _ = NG.seq_fwd(P, CL, b['x_l_c'], None, rnd['C_real'], train, bnu)
    C_unl, _, _ = NG.seq_fwd(P, CL, b['x_u_c'], None, rnd['C_unl'], train, bnu)
    C_unl_d, _, _ = NG.seq_fwd(P, CL, b['x_u_d'], None, rnd['C_unl_d'], train, bnu)
    C_fake, _, _ = NG.seq_fwd(P, CL, Gimg, None, rnd['C_fake'], train, bnu)
    X_P = np.concatenate([b['x_l_d'], b['x_u_d']], axis=0)
    Y_P = np.concatenate([b['y_l_d'], T.argmax_onehot(C_unl_d)], axis=0)
    D_real_l, _, _ = NG.seq_fwd(P, DL, X_P, Y_P, rnd['D_real'], True)
    D_fake_l, _, _ = NG.seq_fwd(P, DL, Gimg, b['y_g'], rnd['D_fake'], True)
    D_unl_l, _, _ = NG.seq_fwd(P, DL, b['x_u_c'], T.argmax_onehot(C_unl), rnd['D_unl'], True)
    D = [T.sigmoid(D_real_l), D_real_l, T.sigmoid(D_fake_l), D_fake_l, T.sigmoid(D_unl_l), D_unl_l]
    return [G, D, [C_real, C_unl, C_unl_d, C_fake]], bnu


# ---- Train_base helpers (train_base.py:43-57,75-84) and _loss_GAN (:113-154), values only -----------------------------------------

def entropy(logits):
    return T.entropy(logits)[0]


def balance_entropy(logits):
    return T.balance_entropy(logits)[0]


def softmax_cross_entropy_loss_w_logits(labels, logits):
    return T.softmax_ce_mean(logits, labels)[0]


def sigmoid_cross_entopy_w_logits(labels, logits):
    return T.bce_mean(logits, labels)[0]


def loss_gan(D, C, Y, Lambda, cifar10):
    """train_base.py:113-154, term for term."""
    _, D_real_l, _, D_fake_l, _, D_unl_l = D
    C_real, C_unl, _C_unl_d, C_fake = C[:4]
    y_g, y_l_c = Y
    d_loss = (sigmoid_cross_entopy_w_logits(np.ones_like(D_real_l), D_real_l) + 0.5 * sigmoid_cross_entopy_w_logits(np.zeros_like(D_fake_l), D_fake_l)
              + 0.5 * sigmoid_cross_entopy_w_logits(np.zeros_like(D_unl_l), D_unl_l))                 # :123-126
    g_loss = 0.5 * sigmoid_cross_entopy_w_logits(np.ones_like(D_fake_l), D_fake_l)                   # :128
    c_loss_real = softmax_cross_entropy_loss_w_logits(y_l_c, C_real)                                # :130
    c_loss_fake = softmax_cross_entropy_loss_w_logits(y_g, C_fake)                                  # :131
    c_loss_unl = T.c_unl_loss(C_unl, D_unl_l)[0]                                                    # :134-138
    c_loss_real = c_loss_real + 1e-6 * entropy(C_unl) + 1e-3 * balance_entropy(C_unl)               # :141-145
    c_loss = 0.01 * 0.5 * c_loss_unl + c_loss_real + Lambda[0] * c_loss_fake                        # :148
    if cifar10:
        c_loss += Lambda[1] * T.mse_mean(C_unl, C[4])[0]                                            # :118,149-152
    return float(d_loss), float(g_loss), float(c_loss)


def training_statistics_cifar10(P, b, rnd, zca, Lambda):
    """The end-of-epoch statistics run of Training/Train_goodGAN.py:280-285: ONE sess.run of [merged_summary_train, d_loss, g_loss, c_loss]
    on the epoch's last feed with train_ph = True — the whole forward graph once (one draw per random op, shared by the three losses), no
    solver.  Side effects committed to P, as the session leaves them: pop_mean of every classifier application in call-site order, the
    generator's batch-norm moving statistics.  Returns the three losses — the numbers the reference logs (:287-288) and summarises."""
    (G, D, C), pops = forward_pass_cifar10(P, b, rnd, zca, True, moving=P)
    for p, v in pops.items():
        P[p + 'meanOnlyBatchNormalization/pop_mean'] = v
    return loss_gan(D, C, [b['y_g'], b['y_l_c']], Lambda, True)
