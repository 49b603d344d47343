import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
import weights as W
from test_kernels_gpu import nerf_weights, cams_for, bf
from cd360 import nerf, ops, _lib
DEV = "cuda"
C, r, n, S, b = 64, 8, 2, 4, 2
w = nerf_weights(C, seed=C + n)
cams = cams_for(b, n, seed=C).to(DEV)
xref = bf(W.tensor("xref", (b, n, r * r, C), seed=C)).to(DEV, torch.bfloat16)
fw = nerf.FusedNerfWeights(*(w[k].to(DEV) for k in ("plane_coefs.0.weight", "plane_coefs.0.bias", "plane_coefs.2.weight", "plane_coefs.2.bias", "nviews.weight", "nviews.bias", "decoder.weight")))
xs = nerf.patch_positions(r, DEV)
t, _ = nerf.depth_samples(S, 2.0, 0.0, DEV, r * r)
Y, lv = nerf.reference_tables(fw, xref)
g_ = torch.Generator().manual_seed(C)
zP = bf(torch.randn(b * n, r * r, C, generator=g_)).to(DEV, torch.bfloat16)
cview = nerf.view_constants(fw, cams)
lib = _lib.load()
hw = r * r; npts = hw * S
nb = int(lib.cd360_nerf_ws_bytes(b, n, r, S))
ws = torch.full((nb,), 0x7f, dtype=torch.uint8, device=DEV)
g = torch.empty(b, npts, C, dtype=torch.bfloat16, device=DEV)
logits = torch.empty(b, n, npts, device=DEV); lse = torch.empty(b, npts, 2, device=DEV)
P = lambda x: None if x is None else x.data_ptr()
rc = lib.cd360_nerf_mlp_aggregate_ws(P(cams), P(xs), P(xs), P(t.contiguous()), 0 if t.dim() == 1 else S, P(Y), P(zP), P(lv), P(cview), P(fw.Wk), None, P(g), P(logits), P(lse), b, n, r, S, C, b * n, P(ws), nb, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("rc", rc, "t", t.shape)
rec = ws[: b * n * npts * 32].view(torch.float32).reshape(b, n, npts, 8)
ml = ws[b * n * npts * 32:].view(torch.float32).reshape(b, npts, 2)
print("rec finite", torch.isfinite(rec[..., :4]).all().item(), rec[0, 0, :3], rec[1, 1, -2:])
print("ml", ml[0, :4], ml[1, -3:], torch.isfinite(ml).all().item(), (ml[..., 1] > 0).all().item())
print("lse vs ml", (lse[..., 1] - ml[..., 1]).abs().max().item())
print("g finite frac", torch.isfinite(g.float()).float().mean().item(), torch.isfinite(g.float()).reshape(b, hw, S, C).all(-1).all(-1))

f = lambda a, b_: float((torch.nan_to_num(a.float(), nan=1e9) - b_.float()).abs().max())
def direct(ws_):
    g2 = torch.empty_like(g)
    rc = lib.cd360_nerf_mlp_aggregate_ws(P(cams), P(xs), P(xs), P(t.contiguous()), 0 if t.dim() == 1 else S, P(Y), P(zP), P(lv), P(cview), P(fw.Wk), None, P(g2), None, None, b, n, r, S, C, b * n, P(ws_), nb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return g2
with _lib.tuning(nerf_kernel=1):
    one = ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk)[0]
print("tuning now", _lib.get_tuning()["nerf_kernel"])
wsA = torch.full((nb,), 0x7f, dtype=torch.uint8, device=DEV)
gA1 = direct(wsA); gA2 = direct(wsA); gA3 = direct(wsA)
print("0x7f ws: first", f(gA1, one), "second", f(gA2, one), "third", f(gA3, one))
if "dbg16" in os.environ.get("CD360_LIB", ""):
    for smp in (0, 2, 40):
        print("sample", smp, "kernel sees (w0..3, lg, m, l, q0, pix0..3) of the LAST view:", gA1[0, smp, :12].float().tolist())
        print("   memory: rec", rec[0, n - 1, smp].tolist(), "ml", ml[0, smp].tolist())
