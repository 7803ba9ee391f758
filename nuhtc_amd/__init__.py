"""nuhtc_amd — MI355X-native engine for NuHTC's htc_lite_swin tile-inference path.

Host mirror of the reference API (`init_detector`, `inference_detector`) over the C-ABI HIP library
libnuhtc_hip.so (include/nuhtc_hip.h).  PyTorch-ROCm is used only for device memory, streams and
torch.distributed."""
__version__ = '0.1.0'

import os as _os

# The streaming path keeps several batches in flight, each on its own HIP stream plus the engine's side stream.  The ROCm runtime
# maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with more streams than queues, unrelated streams serialise
# behind each other.  8 queues: +2.5 % with four batches in flight (1617 -> 1655 tiles/s).  Only effective when set before the
# process's first HIP call, and never overrides a value the caller has set.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
