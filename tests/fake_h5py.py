"""In-memory stand-in for the slice of h5py the replay-buffer HDF5 code uses (File / Group / Dataset, attrs,
create_group, create_dataset, item access, iteration): h5py itself is not installed in this image.  Files live in
a module-level dict keyed by path, so "w" then "r" round-trips inside one process."""
from __future__ import annotations

import numpy as np

_FILES: dict[str, "Group"] = {}


class Dataset:
    def __init__(self, data, compression=None):
        arr = np.asarray(data)
        if arr.dtype == object:
            raise TypeError("Object dtype dtype('O') has no native HDF5 equivalent")
        self._data = arr.copy()
        self.compression = compression
        self.attrs: dict = {}

    def __array__(self, dtype=None, copy=None):
        return self._data if dtype is None else self._data.astype(dtype)

    def __getitem__(self, key):
        return self._data[key]

    @property
    def shape(self):
        return self._data.shape

    @property
    def dtype(self):
        return self._data.dtype

    def __len__(self):
        return len(self._data)


class Group:
    def __init__(self):
        self._children: dict = {}
        self.attrs: dict = {}

    def create_group(self, name):
        g = Group()
        self._children[name] = g
        return g

    def create_dataset(self, name, data=None, compression=None):
        d = Dataset(data, compression)
        self._children[name] = d
        return d

    def __getitem__(self, name):
        return self._children[name]

    def __contains__(self, name):
        return name in self._children

    def items(self):
        return self._children.items()

    def keys(self):
        return self._children.keys()


class File(Group):
    def __init__(self, path, mode="r"):
        if mode == "w":
            super().__init__()
            _FILES[str(path)] = self
        else:
            src = _FILES[str(path)]
            self._children, self.attrs = src._children, src.attrs

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def tree(group, prefix=""):
    """Flat description of a file: {path: ("group", attrs) | ("dataset", dtype, shape, attrs)} (layout comparisons)."""
    out = {prefix or "/": ("group", dict(group.attrs))}
    for k, v in group.items():
        p = f"{prefix}/{k}"
        if isinstance(v, Dataset):
            out[p] = ("dataset", str(v.dtype), tuple(v.shape), dict(v.attrs))
        else:
            out.update(tree(v, p))
    return out
