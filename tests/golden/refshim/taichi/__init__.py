"""Stand-in for the `taichi` package, used ONLY by tests/golden/gen_golden.py in the
authoring container to execute the reference's own kernel bodies as plain Python over
numpy.  Decorators are identities, dtypes are numpy scalar types.
Never imported by the product (kmap_amd/) or shipped to the GPU box as a dependency."""
import numpy as np

u8, u16, u32, u64 = np.uint8, np.uint16, np.uint32, np.uint64
i8, i16, i32, i64 = np.int8, np.int16, np.int32, np.int64
f32, f64 = np.float32, np.float64
cpu, cuda, gpu = "cpu", "cuda", "gpu"
ERROR = "error"


def kernel(fn):
    return fn


def func(fn):
    return fn


def cast(x, t):
    return t(x)


log = np.log
exp = np.exp
sqrt = np.sqrt


def init(*a, **k):
    return None


def set_logging_level(*a, **k):
    return None


class _Cfg:
    arch = "cpu"


cfg = _Cfg()


class _Types:
    u8, u16, u32, u64 = u8, u16, u32, u64
    i8, i16, i32, i64 = i8, i16, i32, i64
    f32, f64 = f32, f64

    @staticmethod
    def ndarray(*a, **k):
        return np.ndarray


types = _Types()


def field(dtype, shape):
    raise NotImplementedError("taichi.field is on the reference's dead GPU_MODE path")
