import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd")]
import torch, bench
from cd360 import synth, ops
from cd360.job import Sampler
from sgm.modules.diffusionmodules.util import timestep_embedding, conv_image
dev = torch.device("cuda", 0)
latent, refs = 32, 6
net = bench.build_model(latent, refs, 50, dev)
pose = [synth.pose_batch(1, refs, seed=100, n_train=50)[0]] * 3
g = torch.Generator(device=dev).manual_seed(7)
ctx = torch.randn(3, 77, 2048, generator=g, device=dev).to(torch.bfloat16)
y = torch.randn(3, 2816, generator=g, device=dev).to(torch.bfloat16)
x = torch.randn(1, 4, latent, latent, generator=g, device=dev)
rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
with torch.no_grad():
    smp = Sampler(net, pose, ctx, y, 50, use_graph=True)
    smp.use_graph = False
    i = 1
    # un-staged pieces
    x1 = smp.step(x.clone(), 0)
    ref = smp.step(x1.clone(), 1)
    smp.gx = x1.clone(); smp._build_stage(x1)
    smp.gi.copy_(smp._iota[i:i+1])
    ops.unet_stage_in(smp.gx, smp.step_tab, smp.gi, smp.w36, smp.b_in, smp.temb_tab, smp.lab, smp.h0, smp.emb_act)
    x3 = smp.gx.expand(3, -1, -1, -1)
    x_in, c_noise, _, _, _ = smp.denoiser.network_inputs(x3, smp.sigmas[i].expand(3), {})
    dt = net.dtype
    emb = net.time_embed(timestep_embedding(c_noise, net.model_channels).to(dt)) + net.label_emb(smp.y.to(dt))
    print("emb_act", rel(smp.emb_act, torch.nn.functional.silu(emb)))
    h = conv_image(net.input_blocks[0][0], x_in.to(dt).contiguous(memory_format=torch.channels_last))
    print("h0", rel(smp.h0.reshape(3, latent, latent, -1).permute(0, 3, 1, 2), h))
    print("tab", smp.step_tab[i], smp.sigmas[i], smp.sigmas[i+1], c_noise)
    eps_ref = net(x_in, timesteps=c_noise, context=smp.ctx, y=smp.y, pose=smp.pose)[0]
    eps_cl = net.forward_staged(smp.h0, smp.emb_act, smp.ctx, smp.pose, latent, latent)
    print("eps", rel(eps_cl.reshape(3, latent, latent, 4).permute(0, 3, 1, 2), eps_ref), eps_cl.shape, eps_cl.stride())
    out = ops.cfg_euler_step_cl(smp.gx, eps_cl, smp.step_tab, smp.gi, smp.scale, smp.scale_im)
    print("x'", rel(out, ref))
with torch.no_grad():
    h_tok = h.permute(0, 2, 3, 1).reshape(3, latent * latent, -1).contiguous()
    eps2 = net.forward_staged(h_tok, smp.emb_act, smp.ctx, smp.pose, latent, latent)
    print("eps with the module's own h0:", rel(eps2.reshape(3, latent, latent, 4).permute(0, 3, 1, 2), eps_ref), torch.equal(eps2.reshape(3, latent, latent, 4).permute(0, 3, 1, 2).float(), eps_ref.float()))
