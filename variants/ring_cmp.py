import numpy as np
a = np.load("gpurun_out/dbg_ring.npy"); b = np.load("gpurun_out/dbg_reg.npy")
d = np.abs(a - b)
print("max", d.max(), "ref max", np.abs(b).max())
print("per row max:", np.round(d.max(axis=(0, 1, 3)), 3))
print("per col max:", np.round(d.max(axis=(0, 1, 2)), 3))
print("per chan max:", np.round(d.max(axis=(0, 2, 3)), 3))
