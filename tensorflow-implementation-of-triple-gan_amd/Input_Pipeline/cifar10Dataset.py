"""Input_Pipeline/cifar10Dataset.py of the reference (class cifar10Dataset, :11-21): file naming
'cifar10_<subset>_<count:06d>.tfrecords' under <data_dir>/Tfrecord, train_size 50000, 3 channel(s),
pixel scaling x/255*2-1 (:60-63).  The pipeline itself is Input_Pipeline/tfrecordDataset.py."""
from Input_Pipeline.tfrecordDataset import tfrecordDataset


class cifar10Dataset(tfrecordDataset):
    PREFIX = 'cifar10'
    TRAIN_SIZE = 50000
    CHANNELS = 3
    UNIT_RANGE = False
