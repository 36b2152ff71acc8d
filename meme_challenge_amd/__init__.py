"""MI355X-native UNITER fine-tuning hot path (gfx950 HIP kernels behind a C ABI).

The package mirrors the reference's module surface for this path only
(model/model.py, model/layer.py, model/meme_uniter.py, the step semantics of
train_template.py and the CLI of train_uniter.py).  All device arithmetic is
done by libuniter_hip.so; there is no CPU or PyTorch-eager fallback: importing
the compute modules without the built library raises.
"""
__version__ = '0.1.0'
