"""Good_GAN_stress64 — the build-defined 64x64x3 stress configuration (BASELINE.json configs[4]).

NOT in the reference: the CIFAR-10 networks of Model/Good_GAN_cifar10.py grown by one resolution stage each, so that the
activation passes (statistics, activations, pooling, dropout, concat) dominate and the HBM side of the step can be
profiled:
  G: dense -> 4x4x512 -> four 5x5 stride-2 transposed convs 256/128/64/3          (4 -> 8 -> 16 -> 32 -> 64)
  D: four stride-1/stride-2 3x3 conv pairs 32/64/128 + 256,256 at 8x8              (64 -> 32 -> 16 -> 8), avg-pool, dense
  C: conv triples 64 / 128 / 256 with max-pool + dropout after each, 3x3 VALID 512, NiN 256, NiN 128, global max-pool
No whitening (a 12288^2 ZCA matrix is not part of the stress definition).  No parity target: roofline capture only.
"""
from Model.Good_GAN_cifar10 import Good_GAN_cifar10


class Good_GAN_stress64(Good_GAN_cifar10):
    C_CONVS = [('conv0_1', 64, 'SAME', False), ('conv0_2', 64, 'SAME', False), ('conv0_3', 64, 'SAME', True),
               ('conv1_1', 128, 'SAME', False), ('conv1_2', 128, 'SAME', False), ('conv1_3', 128, 'SAME', True),
               ('conv2_1', 256, 'SAME', False), ('conv2_2', 256, 'SAME', False), ('conv2_3', 256, 'SAME', True),
               ('conv3', 512, 'VALID', False)]
    D_CONVS = [('conv2d_00', 32, 1, False), ('conv2d_01', 32, 2, True), ('conv2d_10', 64, 1, False), ('conv2d_11', 64, 2, True),
               ('conv2d_20', 128, 1, False), ('conv2d_21', 128, 2, True), ('conv2d_30', 256, 1, False), ('conv2d_31', 256, 1, False)]
    G_DECONVS = [('gg_dconv0', 256), ('gg_dconv1', 128), ('gg_dconv2', 64), ('gg_dconv3', 3)]
    CONSISTENCY = True
    GRAD_BUCKETS = dict(Good_GAN_cifar10.GRAD_BUCKETS, classifier=['classifier/conv1_1/V'])        # the first block here is conv0_*

    def zca(self):
        return None
