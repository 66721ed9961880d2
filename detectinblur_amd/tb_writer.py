"""`SummaryWriter` for `--tensorboard_path` (reference train.py:109-120, evaluate.py:143-151).

`torch.utils.tensorboard` needs the `tensorboard` package; when it imports it is used as is.  Otherwise
`EventFileWriter` below writes the same on-disk format -- a TFRecord stream of `Event` protobufs
(`events.out.tfevents.<time>.<host>`), scalars only, which is all the reference logs -- so the directory
opens in TensorBoard either way.  The protobuf messages are encoded by hand (three nested messages, five
fields), the TFRecord framing is length + masked CRC-32C of length + payload + masked CRC-32C of payload.
"""
import os
import socket
import struct
import time


def _crc32c_table():
    poly, table = 0x82F63B78, []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_TABLE = _crc32c_table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag, value, step, wall_time):
    """Event{wall_time=1 (double), step=2 (int64), summary=5{value=1{tag=1 (string), simple_value=2 (float)}}}"""
    val = _field_bytes(1, tag.encode("utf-8")) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
    summary = _field_bytes(1, val)
    return (_varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(int(step) & (2 ** 64 - 1)) +
            _field_bytes(5, summary))


def encode_version_event(wall_time):
    """Event{wall_time=1, file_version=3 "brain.Event:2"}: the first record of every event file."""
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _field_bytes(3, b"brain.Event:2")


class EventFileWriter(object):
    """add_scalar / flush / close of torch.utils.tensorboard.SummaryWriter, scalars only."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self.path = os.path.join(log_dir, "events.out.tfevents.%010d.%s.%d" % (int(time.time()), socket.gethostname(), os.getpid()))
        self._f = open(self.path, "wb")
        self._record(encode_version_event(time.time()))

    def _record(self, data):
        header = struct.pack("<Q", len(data))
        self._f.write(header + struct.pack("<I", _masked_crc(header)) + data + struct.pack("<I", _masked_crc(data)))

    def add_scalar(self, tag, scalar_value, global_step=None, walltime=None):
        if hasattr(scalar_value, "item"):
            scalar_value = scalar_value.item()
        self._record(encode_scalar_event(tag, scalar_value, 0 if global_step is None else global_step,
                                         time.time() if walltime is None else walltime))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()


def read_scalars(path):
    """[(tag, step, value)] of an event file written by either writer -- used by the tests."""
    out = []
    with open(path, "rb") as f:
        blob = f.read()
    pos = 0

    def varint(buf, p):
        n = shift = 0
        while True:
            b = buf[p]
            p += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, p

    def fields(buf):
        p, res = 0, []
        while p < len(buf):
            key, p = varint(buf, p)
            num, wt = key >> 3, key & 7
            if wt == 0:
                v, p = varint(buf, p)
            elif wt == 1:
                v, p = buf[p:p + 8], p + 8
            elif wt == 5:
                v, p = buf[p:p + 4], p + 4
            else:
                n, p = varint(buf, p)
                v, p = buf[p:p + n], p + n
            res.append((num, v))
        return res

    while pos < len(blob):
        (n,) = struct.unpack("<Q", blob[pos:pos + 8])
        assert struct.unpack("<I", blob[pos + 8:pos + 12])[0] == _masked_crc(blob[pos:pos + 8])
        data = blob[pos + 12:pos + 12 + n]
        assert struct.unpack("<I", blob[pos + 12 + n:pos + 16 + n])[0] == _masked_crc(data)
        pos += 16 + n
        ev = dict(fields(data))
        if 5 in ev:
            for num, val in fields(ev[5]):
                v = dict(fields(val))
                out.append((v[1].decode("utf-8"), ev.get(2, 0), struct.unpack("<f", v[2])[0]))
    return out


def make_writer(path):
    """reference train.py:109-120: earlier logs under `path` are cleared, then a writer is created.  The reference
    removes the whole directory tree; since its defaults point --tensorboard_path and --output_dir at the same
    "debug" directory that also deletes checkpoints, so only event files are removed here."""
    if os.path.isdir(path):
        for name in os.listdir(path):
            if name.startswith("events.out.tfevents."):
                os.remove(os.path.join(path, name))
    print("Creating a writer!")
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(path)
    except ImportError:
        return EventFileWriter(path)
