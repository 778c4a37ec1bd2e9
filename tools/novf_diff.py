"""Bit-wise comparison of two result dumps of tools/soak_parity.py (SOAK_DUMP): the product build against the cross-check build without
-amdgpu-mfma-vgpr-form (make -C ratilqr.jl_amd/csrc novf).  Same sources, same arithmetic, different register allocation: any difference is a
compiler problem (or a race).   python tools/novf_diff.py a.npz b.npz"""
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
assert sorted(a.files) == sorted(b.files), "different problem sets"
diff = []
for k in a.files:
    x, y = a[k], b[k]
    same = x.shape == y.shape and (np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y))
    if same and x.dtype.kind == "f":
        same = np.array_equal(x.view(np.uint64), y.view(np.uint64)) or np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(x[~np.isnan(x)].view(np.uint64), y[~np.isnan(y)].view(np.uint64))
    if not same:
        diff.append(k)
print(f"{len(a.files)} arrays compared bit by bit: {len(diff)} differ" + (f" ({diff[:12]})" if diff else ""))
sys.exit(1 if diff else 0)
