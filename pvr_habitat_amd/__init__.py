"""pvr_habitat_amd: MI355X-native PVR embedding + behavioural-cloning hot path.

Python host side mirrors the reference call surface (sparisi/pvr_habitat src/embeddings.py,
src/models.py, behavioral_cloning/save_embedded_obs.py, main_bc_*.py); the arithmetic lives in
hand-written HIP kernels behind the C-ABI of include/pvr_hip.h (libpvr_hip.so).
"""
import os as _os

# The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This package keeps five streams busy
# (torch's current stream, two compute lanes, H2D, D2H): with four queues two of them share one and serialise - stream_embed ran at
# 53 k frames/s instead of 75 k inside bench.py (profiles/experiments/r02_stream_creation_order.txt).  The variable is read when the
# runtime initialises (the process's first GPU call), so setting it at import time is early enough; an explicit setting wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
