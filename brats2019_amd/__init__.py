"""brats2019_amd -- MI355X-native (gfx950) ResUNet hot path of lachinov/brats2019.

`model`, `loss`, `train` mirror the reference modules of the same names (drop-in surface);
`engine` / `ops` are the host side of the C-ABI library `lib/libresunet_hip.so` (include/resunet_hip.h);
`parallel` is the one-process-per-GPU data-parallel step over RCCL.  Build the library with
`python -m brats2019_amd.build`.  The path is HIP-only: there is no CPU fallback."""

__all__ = ["model", "loss", "train", "engine", "ops", "parallel", "tiling", "compat"]
