"""nuhtc_amd — MI355X-native engine for NuHTC's htc_lite_swin tile-inference path.

Host mirror of the reference API (`init_detector`, `inference_detector`) over the C-ABI HIP library
libnuhtc_hip.so (include/nuhtc_hip.h).  PyTorch-ROCm is used only for device memory, streams and
torch.distributed."""
__version__ = '0.1.0'
