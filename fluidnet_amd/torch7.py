"""Minimal reader for torch7 binary serialisation (enough for FluidNet model files such as
data/models/myModel2D and *_mconf.bin). Host-side plumbing on the "data formats either side of the
path" (SURVEY.md 8f-3); format notes in SURVEY.md Appendix A.

Little-endian stream of typed objects: 0 nil, 1 number(f64), 2 string, 3 table, 4 torch object,
5 boolean, 6/7/8 function. Tables and torch objects are memoised by an int32 index.
"""
import struct

import numpy as np

_STORAGE_DTYPES = {
    "torch.FloatStorage": np.float32, "torch.CudaStorage": np.float32,
    "torch.DoubleStorage": np.float64, "torch.CudaDoubleStorage": np.float64,
    "torch.LongStorage": np.int64, "torch.CudaLongStorage": np.int64,
    "torch.IntStorage": np.int32, "torch.ByteStorage": np.uint8, "torch.CharStorage": np.int8,
    "torch.ShortStorage": np.int16, "torch.HalfStorage": np.float16,
    "torch.CudaHalfStorage": np.float16,
}


class _ObjKey:
    """Hashable wrapper for table-valued table keys."""

    def __init__(self, obj):
        self.obj = obj


class TorchObject:
    """A non-tensor torch class instance (nn.*, cudnn.*, nngraph.Node, tfluids.*): its field table."""

    def __init__(self, typename, fields):
        self.typename = typename
        self.fields = fields

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, default=None):
        return self.fields.get(k, default) if isinstance(self.fields, dict) else default

    def __repr__(self):
        return "TorchObject(%s)" % self.typename


class _Reader:
    def __init__(self, data):
        self.b = data
        self.o = 0
        self.memo = {}

    def _unpack(self, fmt, n):
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += n
        return v[0]

    def i32(self):
        return self._unpack("<i", 4)

    def i64(self):
        return self._unpack("<q", 8)

    def f64(self):
        return self._unpack("<d", 8)

    def string(self):
        n = self.i32()
        s = self.b[self.o:self.o + n]
        self.o += n
        return s.decode("latin-1")

    def obj(self):
        t = self.i32()
        if t == 0:
            return None
        if t == 1:
            v = self.f64()
            return int(v) if (v == v and abs(v) < 2 ** 53 and v == int(v)) else v
        if t == 2:
            return self.string()
        if t == 5:
            return self.i32() == 1
        if t == 3:
            idx = self.i32()
            if idx in self.memo:
                return self.memo[idx]
            out = {}
            self.memo[idx] = out
            n = self.i32()
            for _ in range(n):
                k = self.obj()
                if isinstance(k, (dict, list, np.ndarray)):  # nngraph keys tables by node
                    k = _ObjKey(k)
                out[k] = self.obj()
            return out
        if t == 4:
            idx = self.i32()
            if idx in self.memo:
                return self.memo[idx]
            version = self.string()
            cls = self.string() if version.startswith("V ") else version
            if cls.endswith("Tensor"):
                nd = self.i32()
                size = [self.i64() for _ in range(nd)]
                stride = [self.i64() for _ in range(nd)]
                off = self.i64() - 1
                holder = [None]
                self.memo[idx] = holder  # placeholder (tensors are never self-referential)
                storage = self.obj()
                if storage is None or nd == 0:
                    arr = np.zeros(size, np.float32)
                else:
                    arr = np.lib.stride_tricks.as_strided(
                        storage[off:], shape=size,
                        strides=[s * storage.itemsize for s in stride]).copy()
                self.memo[idx] = arr
                return arr
            if cls.endswith("Storage"):
                n = self.i64()
                dt = np.dtype(_STORAGE_DTYPES[cls])
                arr = np.frombuffer(self.b, dt, n, self.o).copy()
                self.o += n * dt.itemsize
                self.memo[idx] = arr
                return arr
            ob = TorchObject(cls, None)
            self.memo[idx] = ob
            ob.fields = self.obj()
            return ob
        if t in (6, 7, 8):  # functions: (index, bytecode, upvalues) -- skipped
            idx = self.i32()
            if idx in self.memo:
                return self.memo[idx]
            n = self.i32()
            self.o += n
            self.memo[idx] = "<function>"
            self.obj()
            return "<function>"
        raise ValueError("unknown torch7 type tag %d at offset %d" % (t, self.o))


def load(path):
    import sys
    with open(path, "rb") as f:
        data = f.read()
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 20000))  # nngraph models nest deeply
    try:
        return _Reader(data).obj()
    finally:
        sys.setrecursionlimit(old)


def conv_layers(model):
    """Walk an nn.gModule (lib/model.lua) in forward order and return its convolution layers as
    [(weight[nOut, nIn, k...], bias[nOut])] -- `Bank 1: conv stage N` nodes, then the output conv."""
    nodes = model["forwardnodes"]
    out = []
    for i in sorted(nodes):
        data = nodes[i]["data"]
        mod = data.get("module") if isinstance(data, dict) else None
        if isinstance(mod, TorchObject) and "Convolution" in mod.typename:
            w = np.asarray(mod["weight"], np.float32)
            b = np.asarray(mod["bias"], np.float32)
            nout, nin = int(mod["nOutputPlane"]), int(mod["nInputPlane"])
            if "Volumetric" in mod.typename:
                w = w.reshape(nout, nin, int(mod["kT"]), int(mod["kH"]), int(mod["kW"]))
            else:
                w = w.reshape(nout, nin, int(mod["kH"]), int(mod["kW"]))
            out.append((w, b))
    return out
