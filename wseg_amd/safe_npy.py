"""Reading the reference's pickled-dict `.npy` files without executing anything from them.

`voc12/cls_labels.npy` (voc12/data.py:40-44) and the `<name>.npy` CAM dictionaries contrast_infer.py:82-90 writes are
`np.save`d Python dicts, i.e. a pickle stream behind the `.npy` header.  `np.load(allow_pickle=True)` would let such a file
import and call any global; this loader unpickles with a `find_class` that admits only what a dict of numeric numpy arrays
needs (numpy's array reconstructor, `ndarray`, `dtype`, numpy scalars) and refuses every other global.
"""
import pickle

import numpy as np

_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("_codecs", "encode"),          # protocol-2 files written by Python 2 (the reference's cls_labels.npy) carry array bytes as latin-1 text
}


class _NumericUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            if module in ("numpy.core.multiarray", "numpy._core.multiarray"):
                import numpy._core.multiarray as ma
                return getattr(ma, name)
            if module == "_codecs":
                import _codecs
                return _codecs.encode
            return getattr(np, name)
        raise pickle.UnpicklingError(f"refusing global {module}.{name}: only numeric numpy containers are loaded")


def load_pickled_npy(path):
    """The object stored by `np.save(path, obj)` for a dict / list of numeric arrays (restricted unpickling); plain numeric
    `.npy` arrays are returned through np.load(allow_pickle=False)."""
    with open(path, "rb") as f:
        version = np.lib.format.read_magic(f)
        if version == (1, 0):
            shape, fortran, dtype = np.lib.format.read_array_header_1_0(f)
        else:
            shape, fortran, dtype = np.lib.format.read_array_header_2_0(f)
        if not dtype.hasobject:
            f.seek(0)
            return np.load(f, allow_pickle=False)
        arr = _NumericUnpickler(f).load()
    if isinstance(arr, np.ndarray) and arr.shape == ():
        return arr.item()
    return arr
