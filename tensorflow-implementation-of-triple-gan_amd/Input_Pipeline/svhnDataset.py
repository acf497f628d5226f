"""Input_Pipeline/svhnDataset.py of the reference (class svhnDataset, :11-21): file naming
'svhn_<subset>_<count:06d>.tfrecords' under <data_dir>/Tfrecord, train_size 73257, 3 channel(s),
pixel scaling x/255*2-1 (:60-63).  The pipeline itself is Input_Pipeline/tfrecordDataset.py."""
from Input_Pipeline.tfrecordDataset import tfrecordDataset


class svhnDataset(tfrecordDataset):
    PREFIX = 'svhn'
    TRAIN_SIZE = 73257
    CHANNELS = 3
    UNIT_RANGE = False
