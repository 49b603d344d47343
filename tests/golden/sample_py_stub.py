"""What sample.py does to the UNet before sampling (sample.py:247-281), with STAND-IN forwards: `register(model, choices)` rebinds
`forward` on every SpatialTransformer / BasicTransformerBlock instance, by class name, to module-level functions named like sample.py's
(`customforward`, `_customforward`) which raise when executed.  Data-free test infrastructure: the package recognises sample.py's
rebinding by a fingerprint of the function's source (cd360/sample_py_patch.py) and serves it natively (sgm/modules/attention.py,
`_sample_py_patch_kind`); the reference's own functions cannot travel to the GPU box (their outputs do:
tests/golden/customforward_cfg3.npz; their fingerprints do: tests/golden/sample_py_fingerprints.json), so `register(...)` DECLARES these
stand-ins equivalent first (`sample_py_patch.trust`) -- a stand-in that RUNS then means the recognition failed.  `register(...,
declare=False)` skips the declaration: the stand-ins are then just functions named like sample.py's and must be installed and run."""

choices = None  # the driver's global (sample.py:274-278), read by the block forward at every call (sample.py:91)


def customforward(self, x, xr, context=None, contextr=None, pose=None, mask_ref=None, prev_weights=None, timesteps=None):
    _ = (self.norm, self.use_linear, self.proj_in, self.transformer_blocks, self.image_cross, self.poscontrol_interval, self.proj_out)
    raise AssertionError("sample.py-style SpatialTransformer forward was executed: the rebinding was not recognised")


def _customforward(self, x, context=None, context_ref=None, pose=None, mask_ref=None, prev_weights=None, additional_tokens=None,
                   n_times_crossframe_attn_in_self=0):
    _ = (self.references, choices, self.attn1, self.norm1, self.attn2, self.norm2, self.rendered_feat, self.pose_emb_layers,
         self.reference_attn, self.ff, self.norm3)
    raise AssertionError("sample.py-style block forward was executed: the rebinding was not recognised")


def register(model, view_choices, declare=True):
    """sample.py:247-278: rebind by class NAME over the whole module tree, then set the global `choices`."""
    global choices
    if declare:
        from cd360 import sample_py_patch
        sample_py_patch.trust("st", customforward)
        sample_py_patch.trust("block", _customforward)
    for m in model.modules():
        if m.__class__.__name__ == "SpatialTransformer":
            setattr(m, "forward", customforward.__get__(m, m.__class__))
        elif m.__class__.__name__ == "BasicTransformerBlock":
            setattr(m, "forward", _customforward.__get__(m, m.__class__))
    choices = [int(c) for c in view_choices]
