"""kaolin.metrics.render (call sites /root/reference/networks.py:377, trainer.py:793,933)."""
from .._mm import ops

mask_iou = ops.mask_iou
