"""
sgdml_amd -- MI355X-native implementation of sGDML's kernel linear-algebra hot path
(kernel-matrix assembly, analytic / iterative solvers, batched force prediction) behind the
reference's ``GDMLTrain`` / ``GDMLPredict`` API (sgdml/train.py:305, sgdml/predict.py:248).

All numerics run in hand-written HIP kernels (``libgdml_hip.so``, C ABI in include/gdml_hip.h)
called through ctypes.  There is no CPU fallback: importing works anywhere, but creating a
context without the library or without a GPU raises.
"""

__version__ = '1.0.3'  # model/task files carry the reference's code_version (sgdml/__init__.py)

MAX_PRINT_WIDTH = 100
LOG_LEVELNAME_WIDTH = 7

# progress-callback protocol constants (sgdml/__init__.py:31-32)
DONE = 1
NOT_DONE = 0
