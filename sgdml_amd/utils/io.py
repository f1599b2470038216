"""Minimal .npz helpers needed by the hot path's callers (subset of sgdml/utils/io.py)."""
import hashlib

import numpy as np


def dataset_md5(dataset):
    """Fingerprint of a dataset, byte-identical to the reference's (sgdml/utils/io.py:208-230):
    MD5 over the MD5 digests of the raveled 'z', 'R', ['E'], 'F' arrays, hex, utf-8 bytes."""
    outer = hashlib.md5()
    for key in ('z', 'R', 'E', 'F'):
        if key == 'E' and 'E' not in dataset:
            continue
        arr = dataset[key]
        outer.update(hashlib.md5(arr.ravel() if isinstance(arr, np.ndarray) else arr).digest())
    return outer.hexdigest().encode('utf-8')


def is_file_type(arg, type):
    """Load an .npz and check its 'type' field ('d', 't' or 'm')  (io.py:327)."""
    try:
        file = np.load(arg, allow_pickle=True)
    except Exception:
        raise ValueError('{0} is not a valid {1} file.'.format(arg, type))
    if 'type' not in file or file['type'].astype(str) != type[0]:
        raise ValueError('{0} is not a valid {1} file.'.format(arg, type))
    return arg, file
