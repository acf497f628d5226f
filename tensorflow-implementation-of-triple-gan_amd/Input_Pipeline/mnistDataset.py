"""Input_Pipeline/mnistDataset.py of the reference (class mnistDataset, :11-21): file naming
'mnist_<subset>_<count:06d>.tfrecords' under <data_dir>/Tfrecord, train_size 60000, 1 channel(s),
pixel scaling x/255 (:65).  The pipeline itself is Input_Pipeline/tfrecordDataset.py."""
from Input_Pipeline.tfrecordDataset import tfrecordDataset


class mnistDataset(tfrecordDataset):
    PREFIX = 'mnist'
    TRAIN_SIZE = 60000
    CHANNELS = 1
    UNIT_RANGE = True
