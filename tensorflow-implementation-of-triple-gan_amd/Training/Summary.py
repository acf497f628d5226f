"""Scalar summaries — counterpart of the reference's Training/Summary.py (:14-71): <log_dir>/<log_type>/Run_<timestamp>/ with
Comments.txt and a TensorBoard event file.  tf.summary.FileWriter writes a TFRecord file of `Event` protos; the same bytes
are produced here (record framing by the C++ side, tg_record_append; the three tiny protos encoded below):

    event.proto    Event   { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
    summary.proto  Summary { repeated Value value = 1; }   Value { string tag = 1; float simple_value = 2; }

Only scalars are written by the reference's training loop (SUMMARY_SCALAR; image / histogram summaries are off in every
config, config.py:42-46); a `history.csv` with the same numbers is kept next to the event file."""
import os
import socket
import struct
import time

from Training.Saver import _eastern_now


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_event(wall_time, step=None, file_version=None, scalars=None):
    ev = _varint((1 << 3) | 1) + struct.pack('<d', wall_time)
    if step is not None:
        ev += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if scalars:
        vals = b''.join(_ld(1, _ld(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack('<f', float(v))) for tag, v in scalars.items())
        ev += _ld(5, vals)
    return ev


class _FileWriter(object):
    """tf.summary.FileWriter(log_dir): events.out.tfevents.<unix time>.<hostname>, first record = file_version 'brain.Event:2'."""

    def __init__(self, log_dir):
        from tg import io as tgio
        self._io = tgio
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._io.append_record(self.path, encode_event(time.time(), file_version='brain.Event:2'), append=False)

    def add_summary(self, scalars, global_step=None):
        self._io.append_record(self.path, encode_event(time.time(), step=global_step, scalars=scalars))

    def flush(self):
        pass                                                   # every record is written and closed immediately

    def close(self):
        pass


class Summary(object):
    def __init__(self, log_dir, config, **kwargs):                                    # :14-30
        self.config = config
        self.comments = kwargs.get('log_comments', '')
        if 'log_type' in kwargs:
            log_dir = os.path.join(log_dir, kwargs.get('log_type'))
        log_dir = os.path.join(log_dir, 'Run_' + _eastern_now().strftime("%Y-%m-%d_%H_%M_%S"))
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self.summary_writer = _FileWriter(log_dir)
        self._write_comments()
        self._tags = []

    def add_summary(self, summary_dict):                                              # :32-44 -> the "merged summary": tags to evaluate
        self._tags = list(summary_dict.get('scalar', {}).keys()) if 'scalar' in summary_dict else []
        return self._tags

    def write(self, values, step):
        """summary_writer.add_summary(sess.run(merged_summary), step) of Train_goodGAN.py:293,346."""
        scalars = {k: values[k] for k in (self._tags or values) if k in values}
        self.summary_writer.add_summary(scalars, step)
        csv = os.path.join(self.log_dir, 'history.csv')
        new = not os.path.exists(csv)
        with open(csv, 'a') as f:
            if new:
                f.write('step,' + ','.join(scalars) + '\n')
            f.write('%d,' % step + ','.join('%.6g' % float(v) for v in scalars.values()) + '\n')

    def _write_comments(self):                                                        # :63-65
        with open(os.path.join(self.log_dir, 'Comments.txt'), 'w') as txt_file:
            txt_file.write(self.comments)
