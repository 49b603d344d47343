"""cd360: MI355X-native implementation of custom-diffusion360's pose-conditioned denoising hot path.

Python here is plumbing (device memory, streams, module/state_dict surface); the arithmetic of the path runs in
hand-written HIP kernels behind the C ABI of include/cd360_hip.h (custom-diffusion360_amd/csrc)."""
__all__ = ["cameras", "ops", "nerf", "synth"]
