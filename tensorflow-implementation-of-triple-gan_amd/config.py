"""Base configuration class — same attribute names as the reference's config.py:9-93
(including its spellings, e.g. CLA_LEARNINIG_RATE is set by the experiment configs)."""


class Config(object):
    NAME = None
    ## Input pipeline
    DATA_NAME = None
    DATA_DIR = None
    NUM_LABEL = None
    BATCH_SIZE = None
    BATCH_SIZE_L_D = None
    SAMPLE_SIZE = 64

    IMAGE_HEIGHT = None
    IMAGE_WIDTH = None
    CHANNEL = None
    REPEAT = None

    ## Model architecture
    Z_DIM = None
    NUM_CLASSES = None
    MINIBATCH_DIS = False

    ## Training settings
    RESTORE = False
    RUN = None
    RESTORE_EPOCH = None
    BATCH_NORM_DECAY = 0.9
    BATCH_NORM_EPSILON = 1e-5
    LEARNING_RATE = 3e-4
    BETA1 = 0.5

    PRE_TRAIN = False
    EPOCHS = None
    TRAIN_SIZE = None
    VAL_STEP = None
    SAVE_PER_EPOCH = 1
    SUMMARY = True
    SUMMARY_GRAPH = True
    SUMMARY_SCALAR = True
    SUMMARY_IMAGE = False
    SUMMARY_HISTOGRAM = False

    SAMPLE_DIR = None
    LOG_DIR = None
    WEIGHT_DIR = None
    DEBUG = False

    ## additions of the MI355X build (not in the reference)
    SEED = 0                 # initial weights + Philox stream
    USE_HIP_GRAPH = None     # True: replay the solver runs as captured hipGraphs; False: launch eagerly; None: EXEC_MODE decides
    EXEC_MODE = 'auto'       # 'auto': times 'plan' and 'graph' over the first iterations and keeps the faster (Train._auto_mode);
                             # 'overlap': eager launches, filter gradients and independent forward passes on a second HIP stream beside the
                             # input-gradient chain (measured fastest on MI355X / ROCm 7.2: 14.6 ms against 15.0 ms per CIFAR-10 step);
                             # 'plan': that two-stream launch sequence recorded once per solver run and re-issued natively (tg_plan_replay,
                             # include/tg_plan.h) - no interpreter on the launch path;
                             # 'graph': hipGraph replay (single chain); 'eager': eager launches on one stream
    ZCA = None               # (mean, mat) arrays when DATA_DIR holds no cifar10_zca_*.npy
    MFMA_DTYPE = 'f32'       # 'bf16': conv/deconv/dense operands rounded to bf16 inside the MFMA kernels (fp32 accumulate)

    def __init__(self):
        """Set values of computed attributes (config.py:70-73)."""
        self.MIN_QUEUE_EXAMPLES = self.BATCH_SIZE * 3
        self.IMAGE_DIM = [self.IMAGE_HEIGHT, self.IMAGE_WIDTH, self.CHANNEL]

    def display(self):
        print("\nConfigurations:")
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)):
                v = getattr(self, a)
                print("{:30} {}".format(a, v if not isinstance(v, tuple) else '<arrays>'))
        print("\n")

    def config_str(self):
        s = "\nConfigurations:\n"
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)):
                v = getattr(self, a)
                s += "{:30} {}".format(a, v if not isinstance(v, tuple) else '<arrays>')
                s += "\n"
        return s
