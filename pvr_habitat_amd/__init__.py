"""pvr_habitat_amd: MI355X-native PVR embedding + behavioural-cloning hot path.

Python host side mirrors the reference call surface (sparisi/pvr_habitat src/embeddings.py,
src/models.py, behavioral_cloning/save_embedded_obs.py, main_bc_*.py); the arithmetic lives in
hand-written HIP kernels behind the C-ABI of include/pvr_hip.h (libpvr_hip.so).
"""
__version__ = '0.1.0'
