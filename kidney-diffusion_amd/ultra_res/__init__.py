"""Ultra-res outpainting grid driver (reference: sample_ultra_res.py / outpainting.py) re-designed
for one-process-per-GPU scheduling with torch.distributed (RCCL on the MI355X node)."""
from . import distributed, grid  # noqa: F401
