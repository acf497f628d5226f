"""ORACLE — test infrastructure only (see oracle/tf_ops.py header; PARITY UNPINNED).

The set of variables `tf.train.Saver()` would store for the reference's training graph (Training/Saver.py:30-32 saves every global
variable), derived from the scopes of Model/*.py and the optimiser / EMA construction of Training/Train_goodGAN.py:79-103:

  * model variables — `tf.variable_scope(net)` > layer scope > variable.  tf.layers.* called inside `with tf.variable_scope(name)` with
    `name=name` double the scope (Model/modle_base.py:39-47,160-167,248-257: 'good_generator/gg_h0_lin/gg_h0_lin/kernel');
    nn.NiN_WN opens `name` and passes it on to dense_WN (Model/nn.py:581-588: 'classifier/NiN1/NiN1/V'); tf.contrib.layers.batch_norm
    with scope=name creates beta, gamma, moving_mean, moving_variance (modle_base.py:229-237); conv2d_WN / dense_WN create V, b,
    meanOnlyBatchNormalization/pop_mean, g (nn.py:476-492,529-542).
  * optimiser slots — AdamOptimizer(name='Adam_optimizer') (train_base.py:91-97) creates `<var>/Adam_optimizer` (m) and
    `<var>/Adam_optimizer_1` (v) for every variable of its var_list; three optimisers with the same name on disjoint variable lists
    (Train_goodGAN.py:81-91).  [UNVERIFIED-TF: slot naming as tf.train.Optimizer's slot_creator does it]
  * non-slot optimiser state — beta1_power / beta2_power, one pair per optimiser, uniquified by TensorFlow in creation order
    (d, g, c) inside name_scope('Train').  The package stores the step count t instead (beta^t is a function of it).
  * EMA shadows — `ema.apply(c_vars)` creates `<var>/ExponentialMovingAverage` for every classifier variable (Train_goodGAN.py:101-103).
"""
from . import nets_cifar10 as N
from . import nets_goodgan as NG

ADAM_M, ADAM_V, EMA = '/Adam_optimizer', '/Adam_optimizer_1', '/ExponentialMovingAverage'
NON_SLOT = ['Train/beta1_power', 'Train/beta2_power', 'Train/beta1_power_1', 'Train/beta2_power_1', 'Train/beta1_power_2', 'Train/beta2_power_2']


def model_variables(data):
    """[(name, trainable)] of the three networks for DATA_NAME `data` ('cifar10' -> Model/Good_GAN_cifar10.py, else Model/Good_GAN.py)."""
    out = []
    if data == 'cifar10':
        for name, _ in N.generator_param_shapes():
            out.append((name, True))
            if name.endswith('/gamma'):                            # contrib batch_norm: the moving statistics next to beta / gamma
                scope = name[:-len('gamma')]
                out += [(scope + 'moving_mean', False), (scope + 'moving_variance', False)]
        out += [(name, True) for name, _ in N.discriminator_param_shapes()]
        out += [(name, 'pop_mean' not in name) for name, _ in N.classifier_param_shapes()]
    else:
        out += [(name, 'moving_' not in name) for name, _, _ in NG.param_shapes(data)]
    return out


def saver_variables(data):
    """every name tf.train.Saver() writes for the training graph (model variables, Adam slots, beta powers, classifier EMA shadows)."""
    names = []
    for name, trainable in model_variables(data):
        names.append(name)
        if trainable:
            names += [name + ADAM_M, name + ADAM_V]
            if name.startswith('classifier/'):
                names.append(name + EMA)
    return names + NON_SLOT
