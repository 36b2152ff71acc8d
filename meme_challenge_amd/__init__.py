"""MI355X-native UNITER fine-tuning hot path (gfx950 HIP kernels behind a C ABI).

The package mirrors the reference's module surface for this path only
(model/model.py, model/layer.py, model/meme_uniter.py, the step semantics of
train_template.py and the CLI of train_uniter.py).  All device arithmetic is
done by libuniter_hip.so; there is no CPU or PyTorch-eager fallback: importing
the compute modules without the built library raises.
"""
import os as _os

# The step overlaps the weight-gradient GEMMs and the optimizer (side streams) with the dgrad chain (main stream).
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); once torch.distributed / RCCL has created its
# own streams, the main and side streams of this package end up on ONE queue and their kernels serialise: the
# single-process step stays at 14.4 ms, the same step inside an initialised process group takes 15.6-15.9 ms
# (tests/tools/dp_queue_probe.sh; 14.6 ms with 8 queues, no effect on the single-process run).  The runtime reads
# the variable when it initialises, i.e. at the first HIP call -- later than this import.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
