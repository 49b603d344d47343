"""UNet wrapper (reference sgm/modules/diffusionmodules/wrappers.py:8-34): maps the conditioning dict onto UNetModel.forward."""
import torch
import torch.nn as nn

OPENAIUNETWRAPPER = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapper"


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        if compile_model:
            raise NotImplementedError("torch.compile is not part of the HIP path (hand-written kernels, no tracing compiler)")
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, **kwargs):
        if "concat" in c:
            x = torch.cat((x, c["concat"].type_as(x)), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)
