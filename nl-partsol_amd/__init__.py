"""nl-partsol_amd — MI355X-native particle<->grid + stress-update path of NL-PartSol.

The product is the C-ABI library built from csrc/ (see include/nlps_gpu.h); this package holds the
build recipe, the ctypes binding used by tests and bench, and the synthetic-input generator.
"""
