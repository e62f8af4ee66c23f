"""Metrics sink of the legacy runner — utils/runner.py:46-49 `log_monitors`, :101 `SummaryWriter(./exp/<algo>_<env>_<time>)`,
:145-158 (train/<metric> at agent.learn_step after every update, train/reward + train/step per finished episode,
eval/reward every eval_freq episodes).

The reference writes TensorBoard event files through `torch.utils.tensorboard`; tensorboard is not a dependency of this
package, so `ScalarWriter` writes the same files itself — TFRecord framing (length, masked CRC-32C, payload, masked
CRC-32C) around hand-encoded `Event{wall_time, step, summary{value{tag, simple_value}}}` protobufs, readable by
`tensorboard --logdir` — plus a `scalars.csv` next to them.  `add_scalar` / `close` are SummaryWriter's.
"""
import math
import os
import socket
import struct
import time

_CRC_TABLE = []


def _crc32c(data):
    """CRC-32C (Castagnoli), the checksum of the TFRecord format."""
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            _CRC_TABLE.append(c)
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data):
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _bytes_field(num, b):
    return _field(num, 2, _varint(len(b)) + b)


def encode_event(wall_time, step=None, tag=None, value=None, file_version=None):
    """tensorflow.Event: 1 wall_time (double), 2 step (int64), 3 file_version (string), 5 summary (Summary)."""
    ev = _field(1, 1, struct.pack("<d", wall_time))
    if step is not None:
        ev += _field(2, 0, _varint(int(step)))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if tag is not None:
        val = _bytes_field(1, tag.encode()) + _field(2, 5, struct.pack("<f", float(value)))     # Summary.Value
        ev += _bytes_field(5, _bytes_field(1, val))                                                # Summary{value}
    return ev


def tfrecord(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", _masked_crc(head)) + payload + struct.pack("<I", _masked_crc(payload))


class ScalarWriter:
    """`SummaryWriter(logdir)` for scalars: a TensorBoard event file + scalars.csv in `logdir`."""

    _serial = -1

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        ScalarWriter._serial += 1             # (SummaryWriter's own suffix: unique per writer of this process)
        name = f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}.{ScalarWriter._serial}"
        self._ev = open(os.path.join(logdir, name), "wb")
        self._ev.write(tfrecord(encode_event(time.time(), file_version="brain.Event:2")))
        self._csv = open(os.path.join(logdir, "scalars.csv"), "w")
        self._csv.write("wall_time,tag,step,value\n")
        self.count = 0

    def add_scalar(self, tag, scalar_value, global_step=None, walltime=None):
        t = time.time() if walltime is None else walltime
        step = 0 if global_step is None else int(global_step)
        self._ev.write(tfrecord(encode_event(t, step, tag, float(scalar_value))))
        self._csv.write(f"{t:.3f},{tag},{step},{float(scalar_value)!r}\n")
        self.count += 1

    def flush(self):
        self._ev.flush()
        self._csv.flush()

    def close(self):
        if not self._ev.closed:
            self.flush()
            self._ev.close()
            self._csv.close()


def log_monitors(writer, monitors, agent, phase, step):
    """utils/runner.py:46-49: one scalar per monitor under `<phase>/<key>`, NaN values skipped."""
    if writer is None or not monitors:
        return
    for key, value in monitors.items():
        value = float(value)
        if not math.isnan(value):
            writer.add_scalar(f"{phase}/{key}", value, global_step=step)


def read_events(path):
    """Parse an event file written by ScalarWriter back into [(wall_time, step, tag, value)] — checks both CRCs of every
    record (tests; also a minimal reader for environments without tensorboard)."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0

    def varint(buf, p):
        n, shift = 0, 0
        while True:
            b = buf[p]
            p += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, p

    def fields(buf):
        p = 0
        while p < len(buf):
            key, p = varint(buf, p)
            num, wire = key >> 3, key & 7
            if wire == 0:
                v, p = varint(buf, p)
            elif wire == 1:
                v, p = buf[p:p + 8], p + 8
            elif wire == 5:
                v, p = buf[p:p + 4], p + 4
            else:
                n, p = varint(buf, p)
                v, p = buf[p:p + n], p + n
            yield num, v
    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        assert struct.unpack("<I", data[pos + 8:pos + 12])[0] == _masked_crc(head), "length CRC"
        payload = data[pos + 12:pos + 12 + n]
        assert struct.unpack("<I", data[pos + 12 + n:pos + 16 + n])[0] == _masked_crc(payload), "payload CRC"
        pos += 16 + n
        wall, step, tag, val = 0.0, 0, None, None
        for num, v in fields(payload):
            if num == 1:
                (wall,) = struct.unpack("<d", v)
            elif num == 2:
                step = v
            elif num == 5:
                for _, value in fields(v):
                    for n2, v2 in fields(value):
                        if n2 == 1:
                            tag = v2.decode()
                        elif n2 == 2:
                            (val,) = struct.unpack("<f", v2)
        if tag is not None:
            out.append((wall, step, tag, val))
    return out
