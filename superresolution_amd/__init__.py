"""MI355X-native Restormer TransformerBlock path (HIP kernels behind a C ABI).

``superresolution_amd.model.Restromer`` mirrors the reference module surface;
``superresolution_amd.ops`` holds the tensor-level operators; ``_lib`` binds
``librestormer_hip.so``.
"""
