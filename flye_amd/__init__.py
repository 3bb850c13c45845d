"""flye_amd: MI355X-native overlap hot path of Flye 2.8.1 behind a C ABI.

Only what the hot path needs lives here: ``csrc/`` holds the HIP
kernels and the C-ABI shim (``include/flye_gpu.h``), ``gpu.py`` is the ctypes
binding mirroring the reference's VertexIndex / OverlapDetector /
OverlapContainer interface, ``config.py`` the per-read-type parameter presets
and ``synth.py`` the seeded read simulator used by tests and bench.
"""
