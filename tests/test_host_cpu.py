"""Host-side logic of the product package, on CPU: drop-in surface (class / argument / state_dict parity with the reference),
the weight re-layouts feeding the fused kernel (validated through a test-only torch emulation of the kernel's arithmetic
against the oracle), the C-ABI export list, and loud failure without a GPU."""
import ctypes
import gzip
import json
import os
import re

import numpy as np
import pytest
import torch

import weights as W
from oracle import pose_path as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def keys(name):
    with gzip.open(os.path.join(GOLD, name + ".keys.json.gz"), "rt") as f:
        return json.load(f)


def shapes(mod):
    return {k: list(v.shape) for k, v in mod.state_dict().items()}


# ------------------------------------------------------------------------------------------------ drop-in surface
def test_block_and_st_state_dict_match_reference():
    from sgm.modules.attention import BasicTransformerBlock, SpatialTransformer
    blk = BasicTransformerBlock(64, 1, 64, context_dim=32, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2,
                                num_samples=4, rgb_predict=True, mode="feature-nerf", stratified=True)
    assert shapes(blk) == keys("block")
    st = SpatialTransformer(128, 2, 64, depth=5, context_dim=32, use_linear=True, attn_type="softmax-xformers", use_checkpoint=False,
                            image_cross=True, rgb_predict=True, far=2, num_samples=4, mode="feature-nerf", stratified=True)
    assert shapes(st) == keys("st")
    pose_blocks = [d for d, b in enumerate(st.transformer_blocks) if hasattr(b, "pose_emb_layers")]
    assert pose_blocks == [0, 4]  # d % poscontrol_interval == 0 (attention.py:772)
    w = st.transformer_blocks[0].pose_emb_layers.weight
    assert torch.equal(w, torch.cat([torch.eye(128), torch.zeros(128, 128)], 1))  # [I, 0] init (attention.py:515-516)
    assert torch.all(st.transformer_blocks[0].pose_featurenerf.model.decoder.weight == 0) and torch.all(st.proj_out.weight == 0)


def test_unet_tiny_state_dict_matches_reference():
    from make_golden_params import UNET_TINY
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    net = UNetModel(**UNET_TINY)
    assert shapes(net) == keys("unet_tiny")


def test_sdxl_config_instantiates_with_reference_keys():
    """configs/train_co3d_concept.yaml:27-54 (network_config) -> UNetModel via instantiate_from_config, on the meta device:
    1836 tensors / 2634.1 M parameters, names and shapes identical to the reference's."""
    from make_golden_params import SDXL_NETWORK_CONFIG
    from sgm.util import instantiate_from_config
    with torch.device("meta"):
        net = instantiate_from_config(SDXL_NETWORK_CONFIG)
    ref = keys("unet_sdxl")
    mine = shapes(net)
    assert mine == ref
    assert abs(sum(int(np.prod(s)) for s in mine.values()) / 1e6 - 2634.12) < 0.5
    from cd360.sampling import pose_blocks
    names = [n for n, _ in pose_blocks(net)]
    assert len(names) == 12 and names[0] == "input_blocks.4.1.transformer_blocks.0" and "middle_block.1.transformer_blocks.8" in names
    trainable = [k for k, _ in net.named_parameters() if "pose" in k]
    assert abs(sum(int(np.prod(mine[k])) for k in trainable) / 1e6 - 66.7) < 0.2  # the params main.py fine-tunes (diffusion.py:139-144)


def test_attention_modes_and_signatures():
    import inspect
    from sgm.modules import attention as A
    assert set(A.BasicTransformerBlock.ATTENTION_MODES) == {"softmax", "softmax-xformers"}
    sig = inspect.signature(A.BasicTransformerBlock.forward)
    assert list(sig.parameters)[1:] == ["x", "context", "context_ref", "pose", "mask_ref", "prev_weights", "additional_tokens",
                                        "n_times_crossframe_attn_in_self"]
    assert list(inspect.signature(A.SpatialTransformer.forward).parameters)[1:] == ["x", "xr", "context", "contextr", "pose", "mask_ref", "prev_weights"]
    assert list(inspect.signature(A.MemoryEfficientCrossAttention.forward).parameters)[1:] == ["x", "context", "mask", "additional_tokens",
                                                                                                "n_times_crossframe_attn_in_self"]
    from sgm.modules.nerfsd_pytorch3d import NerfSDModule, VolRender
    assert list(inspect.signature(NerfSDModule.forward).parameters)[1:] == ["pose", "xref", "mask_ref", "prev_weights", "imp_sample_next_step"]
    assert list(inspect.signature(VolRender.forward).parameters)[1:] == ["features", "densities", "dists", "return_weight", "densities_uniform",
                                                                         "dists_uniform", "return_weights_uniform", "rgb"]


def test_sample_py_rebinding_is_recognised_and_served_natively():
    """sample.py:247-262 rebinds `forward` on every SpatialTransformer / BasicTransformerBlock instance.  Those two functions ARE the
    sampling mode these classes carry natively, so the assignment is recorded instead of installed (no instance-level `forward`: the fused
    routes stay available), `choices` is read from the rebound function's own globals at call time, and anything else assigned to `forward`
    is installed as usual.  (Equivalence with sample.py's own functions: tests/golden/customforward_cfg3.npz, GPU tests.)"""
    import sample_py_stub as SP
    from cd360 import routes
    from sgm.modules import attention as A
    st = A.SpatialTransformer(128, 2, 64, depth=5, context_dim=32, use_linear=True, attn_type="softmax-xformers", use_checkpoint=False,
                              image_cross=True, rgb_predict=True, far=2, num_samples=4, mode="feature-nerf", stratified=True)
    SP.register(st, [0, 2])
    blocks = list(st.transformer_blocks)
    assert "forward" not in st.__dict__ and all("forward" not in b.__dict__ for b in blocks)
    assert "_sample_py" in st.__dict__ and all("_sample_py" in b.__dict__ for b in blocks)
    assert not A._watched(st), "a recognised rebinding is not an observer: the fused route stays open"
    pose_blk, plain_blk = blocks[0], blocks[1]
    assert pose_blk.image_cross and not plain_blk.image_cross and pose_blk.reference_choices is None
    pose_blk.rendered_feat = torch.zeros(1)
    A._sync_sample_py(pose_blk)
    assert pose_blk.reference_choices == [0, 2] and pose_blk.rendered_feat is None and pose_blk.attn2.cache_context_kv
    pose_blk.rendered_feat = torch.zeros(1)
    A._sync_sample_py(pose_blk)  # unchanged choices: the cached render survives
    assert pose_blk.rendered_feat is not None
    SP.choices = [1, 3]          # the driver picked other views: the cached render is for the old ones
    A._sync_sample_py(pose_blk)
    assert pose_blk.reference_choices == [1, 3] and pose_blk.rendered_feat is None
    A._sync_sample_py(plain_blk)
    assert plain_blk.reference_choices is None
    SP.choices = None
    with pytest.raises(RuntimeError):
        A._sync_sample_py(pose_blk)
    # any other function assigned to `forward` is installed (and observed: strict module route)
    other = A.BasicTransformerBlock(64, 1, 64, context_dim=32, checkpoint=False, attn_mode="softmax-xformers")
    other.forward = (lambda self, x, **kw: x).__get__(other, type(other))
    assert "forward" in other.__dict__ and A._watched(other)
    # ... and so are sample.py's own, when the recognition is switched off
    with routes.override(strict_sample_py=True):
        blk = A.BasicTransformerBlock(64, 1, 64, context_dim=32, checkpoint=False, attn_mode="softmax-xformers")
        SP.register(blk, [0])
        assert "forward" in blk.__dict__ and "_sample_py" not in blk.__dict__


def test_sample_py_fingerprints_are_the_reference_functions():
    """cd360/sample_py_patch.py's built-in fingerprints = the recorded ones (tests/golden/sample_py_fingerprints.json, written by
    make_golden.py from the reference's sample.py) and, where the reference tree is present (this container, not the GPU box), = the
    hash of sample.py's own two functions read in place -- through the same public entry point a user's rebinding goes through."""
    import ast
    import json
    from cd360 import sample_py_patch as SPP
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_py_fingerprints.json")))
    assert rec["_customforward"] in SPP.KNOWN["block"] and rec["customforward"] in SPP.KNOWN["st"]
    ref = "/root/reference/sample.py"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present")
    tree = ast.parse(open(ref).read())
    got = {n.name: SPP.fingerprint_node(n) for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in rec}
    assert got == rec
    # formatting, comments and line numbers do not move it; an edit of the body does
    src = ast.unparse([n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_customforward"][0])
    again = ast.parse("# a comment\n\n" + src.replace("\n", "\n\n", 3)).body[0]
    assert SPP.fingerprint_node(again) == rec["_customforward"]
    edited = ast.parse(src.replace("return", "x = x + 0\n    return", 1) if "\n    return" in src else src + "\n    pass").body[0]
    assert SPP.fingerprint_node(edited) != rec["_customforward"]


def test_edited_customforward_runs_its_own_body(tmp_path):
    """A function NAMED `_customforward` that touches every attribute sample.py's touches but has another body (a user's edited copy of
    sample.py) is not sample.py's: it is installed on the instance, runs ITS body, sends the block to the strict route, and draws one
    warning.  (Until round 6 the recognition was by name + co_names and silently ran the built-in sampling mode instead.)"""
    import importlib.util
    import warnings
    import sample_py_stub as SP
    from sgm.modules import attention as A
    mod_src = (
        "calls = []\n"
        "choices = None\n"
        "def _customforward(self, x, context=None, context_ref=None, pose=None, mask_ref=None, prev_weights=None, additional_tokens=None,\n"
        "                   n_times_crossframe_attn_in_self=0):\n"
        "    _ = (self.references, choices, self.attn1, self.norm1, self.attn2, self.norm2, self.rendered_feat, self.pose_emb_layers,\n"
        "         self.reference_attn, self.ff, self.norm3)\n"
        "    calls.append('edited')\n"
        "    return x * 2\n")
    path = tmp_path / "edited_sample.py"
    path.write_text(mod_src)
    spec = importlib.util.spec_from_file_location("edited_sample", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    blk = A.BasicTransformerBlock(64, 1, 64, context_dim=32, checkpoint=False, attn_mode="softmax-xformers")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        blk.forward = mod._customforward.__get__(blk, type(blk))
    assert "forward" in blk.__dict__ and "_sample_py" not in blk.__dict__ and A._watched(blk)
    assert any("differs from sample.py" in str(x.message) for x in w)
    for name in ("references", "rendered_feat", "pose_emb_layers", "reference_attn"):  # (a plain block lacks the pose attributes)
        if not hasattr(blk, name):
            object.__setattr__(blk, name, None)
    x = torch.ones(1, 4, 64)
    assert torch.equal(blk(x), x * 2) and mod.calls == ["edited"]
    # the stand-ins of the test suite are recognised only because register() declares them (trust); undeclared they are installed
    blk2 = A.BasicTransformerBlock(64, 1, 64, context_dim=32, checkpoint=False, attn_mode="softmax-xformers")
    from cd360 import sample_py_patch as SPP
    saved = {k: set(v) for k, v in SPP.KNOWN.items()}
    try:
        for k in SPP.KNOWN:
            SPP.KNOWN[k] = {fp for fp in saved[k] if fp in json_fingerprints()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            SP.register(blk2, [0], declare=False)
        assert "forward" in blk2.__dict__ and "_sample_py" not in blk2.__dict__
    finally:
        SPP.KNOWN.update(saved)


def json_fingerprints():
    import json
    return set(json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_py_fingerprints.json"))).values())


def test_submodule_swap_after_first_forward_is_seen_by_the_observer_check():
    """`_watched` caches the flat submodule list of a block; registering a module anywhere (a wrapper swapped in for attn / ff after the
    first forward) must invalidate it, or a hook on the new submodule would be missed and the fused route would keep reading the old weights."""
    from sgm.modules import attention as A
    blk = A.BasicTransformerBlock(64, 1, 64, context_dim=32, checkpoint=False, attn_mode="softmax-xformers")
    assert not A._watched(blk)
    wrapper = torch.nn.Sequential(blk.ff)
    wrapper.register_forward_hook(lambda m, i, o: None)
    blk.ff = wrapper
    assert A._watched(blk)


def test_cpu_forward_fails_loudly():
    """There is no CPU / PyTorch fallback behind the operators."""
    from cd360 import ops
    from cd360._lib import Cd360Error
    from sgm.modules.attention import MemoryEfficientCrossAttention
    att = MemoryEfficientCrossAttention(64, heads=1, dim_head=64)
    with pytest.raises(Cd360Error):
        att(torch.randn(1, 8, 64))
    with pytest.raises(Cd360Error):
        ops.memory_efficient_attention(torch.randn(2, 8, 64), torch.randn(2, 8, 64), torch.randn(2, 8, 64))
    # the operators added in round 4: inverse-CDF depth sampling (f4), the view-logit column, the render loss terms, the derived FeatureNeRF weights
    from sgm.modules.nerfsd_pytorch3d import Raymarcher
    bf = lambda *s: torch.randn(*s, dtype=torch.bfloat16)
    calls = [lambda: ops.sample_pdf(torch.linspace(0, 2, 5)[None], torch.ones(1, 4), torch.rand(1, 4)),
             lambda: Raymarcher(num_samples=4, stratified=False, training=False).importance_sampling(torch.rand(1, 16, 4, 1), 16, 4, "cpu"),
             lambda: ops.rowdot1(bf(4, 64), torch.randn(64)),
             lambda: ops.render_loss(torch.rand(2, 16, 1), torch.rand(2, 16, 4, 1), None, torch.rand(2, 16), torch.rand(2, 16), None, None, None),
             lambda: ops.nerf_pack_weights(bf(64, 262), bf(64), bf(64), bf(262), bf(1), bf(4, 64), torch.zeros(112, dtype=torch.int32))]
    for call in calls:
        with pytest.raises(Cd360Error):
            call()


def test_cpu_autograd_path_fails_loudly_too():
    """The differentiable route (cd360/grad.py) has no CPU / PyTorch fallback either: a CPU tensor that requires grad reaches the same
    C-ABI front end and raises, for every operator that records itself on the autograd tape."""
    from cd360 import ops
    from cd360._lib import Cd360Error
    r = lambda *s: torch.randn(*s, dtype=torch.bfloat16, requires_grad=True)
    calls = [lambda: ops.attention(r(1, 8, 64), r(1, 8, 64), r(1, 8, 64), 1),
             lambda: ops.self_attention_qkv(r(1, 8, 192), 1),
             lambda: ops.geglu(r(4, 128)),
             lambda: ops.add_layernorm(r(4, 64), r(4, 64), torch.ones(64, dtype=torch.bfloat16), torch.zeros(64, dtype=torch.bfloat16), 1e-5),
             lambda: ops.gn_silu(r(1, 16, 64), torch.ones(64), torch.zeros(64), 32, 1e-5, True),
             lambda: ops.volrender(torch.randn(1, 4, 3, 8, requires_grad=True), torch.randn(1, 4, 3), torch.rand(3)),
             lambda: ops.rowdot4(r(4, 64), torch.randn(4, 64)),
             lambda: ops.conv_igemm(r(1, 16, 64), torch.zeros(64, 576, dtype=torch.bfloat16), None, 1, 4, 4, 9, w_dgrad=lambda: None)]
    for call in calls:
        with pytest.raises(Cd360Error):
            call()


# ------------------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from cd360 import _lib
    header = open(os.path.join(ROOT, "include", "cd360_hip.h")).read()
    declared = set(re.findall(r"\b(cd360_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load(check_symbols=True)  # raises if the .so is missing or a symbol is absent
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)
    assert lib.cd360_nerf_k_padded() == 112
    assert lib.cd360_conv_stats_slabs(320) == 4 and lib.cd360_conv_stats_slabs(640) == 2 and lib.cd360_conv_k_order(1280, 9) == 5


def test_gemm_dispatch_contract_of_the_sdxl_shapes():
    """The host side sizes two buffers from what the GEMM dispatcher will choose (no GPU needed to ask): the row-statistics partials of a
    residual-stream GEMM (one per `cd360_gemm_tile_n` columns, consumed by the next LayerNorm fold) and the 64- or 32-row channel
    statistics of `proj_out` (consumed by the next GroupNorm).  Pin both for the shapes of the SDXL UNet at cfg-A / cfg-B batch sizes, and
    the small-batch tilings (64 x 128 tiles up to 128 tiles of 128 x 128; `gemm_small` = 0 switches the rule off)."""
    from cd360 import _lib
    lib = _lib.load()
    for b in (1, 2, 3):
        for tokens, C in ((4096, 640), (1024, 1280), (1024 // 4, 1280), (4096 // 4, 640)):  # 1024^2 and 512^2 images
            M = b * tokens
            assert lib.cd360_gemm_tile_n(M, C) in (128, 256), (M, C)           # C -> C, FF2: narrow outputs
            assert lib.cd360_gemm_tile_n(M, 3 * C) in (128, 192, 256), (M, C)   # q|k|v
            slab = lib.cd360_gemm_cstats_rows(M, C)                              # proj_out: 64-row slabs, 32 on the 64 x 128 tiling
            assert slab == (32 if -(-M // 128) * (C // 128) <= 128 else 64), (M, C, slab)
    assert lib.cd360_gemm_cstats_rows(3072, 1280) == 64 and lib.cd360_gemm_cstats_rows(12288, 640) == 64
    assert lib.cd360_gemm_tile_n(3072, 1280) == 128 and lib.cd360_gemm_tile_n(3072, 3840) == 192
    assert lib.cd360_gemm_tile_n(1024, 3840) == 128 and lib.cd360_gemm_tile_n(1024, 5120) == 128 and lib.cd360_gemm_tile_n(1024, 10240) in (192, 256)
    with _lib.tuning(gemm_small=0):  # the A/B partner: the rules before the small-batch tilings
        assert lib.cd360_gemm_cstats_rows(1024, 1280) == 64 and lib.cd360_gemm_tile_n(1024, 3840) in (192, 256)
    assert lib.cd360_gemm_cstats_rows(1024, 1280) == 32


@pytest.mark.parametrize("cin", [64, 320, 640, 960, 1280, 1920, 2560])
def test_packed_conv_weight_follows_the_kernels_k_tile_walk(cin):
    """ops.pack_conv_weight against the K-tile walk of the implicit-im2col convolution (gemm8p.hip `conv_next`: K-tile after K-tile, chunk
    in group fastest, then tap, then group, with cd360_conv_k_order(Cin, 9) chunks per group): K-tile kt of the packed matrix must be the
    64-channel chunk and the tap the kernel shifts the input pixel by for that tile."""
    from cd360 import _lib, ops
    kg = _lib.load().cd360_conv_k_order(cin, 9)
    assert (cin // 64) % kg == 0 and 1 <= kg <= 5
    cout = 16
    ci, tap = torch.arange(cin).view(1, cin, 1, 1), torch.arange(9).view(1, 1, 3, 3)
    w = (ci * 9 + tap).expand(cout, cin, 3, 3).float() + torch.arange(cout).view(cout, 1, 1, 1) * 0.0  # value = code of (channel, tap)
    wp = ops.pack_conv_weight(w.to(torch.float32)).float()  # bf16-rounded codes: compare against the same rounding
    want = w.to(torch.bfloat16).float()
    j = t = cg = 0
    for kt in range(9 * cin // 64):
        chunk = cg * kg + j
        assert torch.equal(wp[:, kt * 64:(kt + 1) * 64], want[:, chunk * 64:(chunk + 1) * 64, t // 3, t % 3]), (kt, chunk, t)
        j += 1
        if j == kg:
            j = 0
            t += 1
            if t == 9:
                t = 0
                cg += 1
    assert cg == cin // 64 // kg and j == 0 and t == 0  # the walk ends exactly at the last group


# ------------------------------------------------------------------------------------------------ fused-render algebra
def _emulate_fused_kernel(fw, cams, xref, S, far):
    """Test-only torch restatement of csrc/nerf_fused.hip + cd360/nerf.py (fp32): validates the weight re-layouts and the algebra."""
    from cd360 import nerf
    b, n, hw, C = xref.shape
    r = int(hw ** 0.5)
    xs = nerf.patch_positions(r, "cpu")
    t, dists = nerf.depth_samples(S, far, 0.0, "cpu", hw)
    rays = O.patch_rays(cams, xs, xs)
    pts = O.ray_points(rays, t[None, None])  # [b,hw,S,3]
    Y, lv = nerf.reference_tables(fw, xref)
    Y, lv = Y.float().reshape(b, n, hw, C), lv.reshape(b, n, hw, 1)
    # plucker table (kernel cd360_plucker_features) via the oracle's pieces
    tgt = rays[:, 0]
    cam_o = O.world_to_view(cams[:, 1:, None, :], tgt[:, None, :, :3])
    cam_d = O.rotate_to_view(cams[:, 1:, None, :], tgt[:, None, :, 3:])
    pl = O.plucker(torch.cat([cam_o, cam_d], -1))
    pf = torch.cat([O.positional_encoding(pl, 8), cam_d, torch.zeros(b, n, hw, 5)], -1)
    zP = (pf.reshape(-1, 104) @ fw.Wp_t + fw.b1).reshape(b, n, hw, C)
    cview = nerf.view_constants(fw, cams)
    grid = O.sample_grid(cams, pts)
    q = O.world_to_view(cams[:, 1:, None, None, :], pts[:, None])  # [b,n,hw,S,3]
    feats = torch.zeros(b, n, hw, S, 112)
    for k, col in enumerate(nerf.xyz_k_columns(C)):
        ks, h, j = k // 16, (k // 8) % 2, k % 8
        w = ks * 4 + (j >> 1)
        if w < 24:
            rev = q[..., w % 3] * (2.0 if h else 1.0) * 2.0 ** (2 * (w // 3) - 9)
            feats[..., k] = torch.cos(2 * np.pi * rev) if j & 1 else torch.sin(2 * np.pi * rev)
        elif col >= 0:
            feats[..., k] = q[..., col - (C + 96)]
    z = feats @ fw.Wk.float().t() + zP[:, :, :, None] + O.gather_bilinear(Y, grid)
    logit = O.gather_bilinear(lv, grid)[..., 0] + cview[:, :, None, None]
    a = torch.softmax(logit, 1)[..., None]
    g = (a * torch.nn.functional.silu(z)).sum(1)
    h = g @ fw.W2_t.float() + fw.b2.float()
    return h, h @ fw.Wd.t(), a


def test_fused_render_algebra_matches_oracle():
    from cd360 import nerf, synth
    from cd360.cameras import pack_cameras
    C, r, n, S, b = 64, 8, 3, 4, 2
    w = {k[len("model."):]: v for k, v in W.synth_state_dict({
        "model.plane_coefs.0.weight": (C, C + 198), "model.plane_coefs.0.bias": (C,), "model.plane_coefs.2.weight": (C, C),
        "model.plane_coefs.2.bias": (C,), "model.nviews.weight": (1, C + 198), "model.nviews.bias": (1,), "model.decoder.weight": (4, C)}, 7).items()}
    cams = pack_cameras(synth.pose_batch(b, n, seed=13))
    xref = W.tensor("xref", (b, n, r * r, C), seed=7)
    fw = nerf.FusedNerfWeights(w["plane_coefs.0.weight"], w["plane_coefs.0.bias"], w["plane_coefs.2.weight"], w["plane_coefs.2.bias"],
                               w["nviews.weight"], w["nviews.bias"], w["decoder.weight"], dtype=torch.float32)
    fw.Wk = fw.Wk_f32  # keep the emulation in fp32: this test is about layout/algebra, not bf16 rounding
    h, dec, a = _emulate_fused_kernel(fw, cams, xref, S, 2.0)
    feats, sigma, _, attn, rgb, _ = O.nerf_module(w, cams, xref, S, 2.0)
    assert torch.allclose(a, attn, atol=1e-5)
    assert torch.allclose(h, feats, atol=2e-4, rtol=1e-3), (h - feats).abs().max()
    assert torch.allclose(dec[..., 3:], sigma, atol=2e-4) and torch.allclose(dec[..., :3], rgb, atol=2e-4)


def test_xyz_k_columns_is_a_bijection_onto_the_99_inputs():
    from cd360.nerf import xyz_k_columns
    cols = xyz_k_columns(640)
    used = [c for c in cols if c >= 0]
    assert len(cols) == 112 and sorted(used) == list(range(640, 640 + 99))


def test_reference_sampling_context_assembly():
    """_references_as_context == sample.py:89-96 (null image for the unconditional third)."""
    from sgm.modules.attention import BasicTransformerBlock
    blk = BasicTransformerBlock(64, 1, 64, context_dim=32, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, num_samples=4,
                                rgb_predict=True, mode="feature-nerf")
    refs = torch.arange(5 * 4 * 64, dtype=torch.float32).reshape(5, 4, 64)
    blk.register_buffer("references", refs)
    blk.reference_choices = [0, 2]
    c3 = blk._references_as_context(3)
    assert c3.shape == (3, 2, 4, 64)
    assert torch.equal(c3[0, 0], refs[4]) and torch.equal(c3[0, 1], refs[4])
    assert torch.equal(c3[1, 0], refs[0]) and torch.equal(c3[1, 1], refs[2]) and torch.equal(c3[2], c3[1])
    c2 = blk._references_as_context(2)
    assert torch.equal(c2[0, 1], refs[4]) and torch.equal(c2[1, 1], refs[2])


def test_pack_cameras_roundtrip_and_cache():
    from cd360 import synth
    from cd360.cameras import pack_cameras, unpack_cameras
    pose = synth.pose_batch(2, 3, seed=5)
    packed = pack_cameras(pose)
    assert packed.shape == (2, 4, 16)
    again = pack_cameras(unpack_cameras(packed))
    assert torch.equal(packed, again)
    R = packed[..., :9].reshape(2, 4, 3, 3)
    assert torch.allclose(R @ R.transpose(-1, -2), torch.eye(3).expand(2, 4, 3, 3), atol=1e-5)


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under custom-diffusion360_amd/ may import or execute it."""
    pkg = os.path.join(ROOT, "custom-diffusion360_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "oracle." in src.replace("oracle/pose_path.py", ""):
                    offenders.append(os.path.join(d, f))
    assert not offenders, offenders


def test_layernorm_fold_and_geglu_packing_algebra():
    """Host side of cd360_gemm_bf16's fused epilogues (CPU math only): ops.pack_ln_linear gives (W', rowsum(W'), c) with
    LN(x) W^T + b == rstd (x W'^T - mu rowsum(W')) + c, and ops.geglu_row_order interleaves value / gate rows per 32 output columns."""
    import torch.nn.functional as F
    from cd360 import ops
    g = torch.Generator().manual_seed(0)
    K, N, M = 64, 128, 40
    x = torch.randn(M, K, generator=g) * 1.7 + 0.3
    w, b = torch.randn(N, K, generator=g) / 8, torch.randn(N, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    wp, wsum, cb = ops.pack_ln_linear(w, b, gamma, beta)
    assert wp.dtype == torch.bfloat16 and wsum.dtype == torch.float32 and cb.dtype == torch.float32
    mu, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + 1e-5)
    folded = rstd * (x @ wp.float().t() - mu * wsum[None]) + cb[None]
    want = F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-5), w, b)
    assert (folded - want).abs().max() < 2e-2 * want.abs().max()  # only the bf16 rounding of gamma o W separates them
    exact = rstd * (x @ (w * gamma[None]).t() - mu * (w * gamma[None]).sum(1)[None]) + cb[None]
    assert torch.allclose(exact, want, atol=1e-4, rtol=1e-4)       # the identity itself, without that rounding
    # row statistics as the kernel's epilogue forms them: sum and sum of squares per row
    assert torch.allclose(mu[:, 0], x.sum(1) / K) and torch.allclose(var[:, 0], (x * x).sum(1) / K - mu[:, 0] ** 2, atol=1e-5)
    inner = 96
    perm = ops.geglu_row_order(inner)
    assert sorted(perm.tolist()) == list(range(2 * inner))
    for grp in range(inner // 32):
        blk = perm[grp * 64:(grp + 1) * 64]
        assert blk[:32].tolist() == list(range(grp * 32, grp * 32 + 32)) and blk[32:].tolist() == list(range(inner + grp * 32, inner + grp * 32 + 32))


def test_per_stream_tuning_table_semantics_without_a_gpu():
    """cd360_set_stream_tuning / cd360_get_stream_tuning / cd360_query_stream are host-side bookkeeping: exercised here on made-up stream
    handles (no launch).  An override belongs to its stream only; the default is untouched; removing restores the default; the table holds
    16 streams and says so; the shape queries of the calling thread follow cd360_query_stream, other threads do not see that context."""
    import ctypes
    import threading
    from cd360 import _lib
    lib = _lib.load()
    base = _lib.get_tuning()
    h1, h2 = 0x1000, 0x2000
    try:
        _lib.set_stream_tuning(h1, gemm_cfg=3)
        assert _lib.get_stream_tuning(h1)["gemm_cfg"] == 3 and _lib.get_stream_tuning(h2) == base and _lib.get_tuning() == base
        _lib.set_stream_tuning(h1, conv_cfg=2)  # a second field on the same stream: the first is kept
        assert _lib.get_stream_tuning(h1)["gemm_cfg"] == 3 and _lib.get_stream_tuning(h1)["conv_cfg"] == 2
        # shape queries: M = 3072, N = 1280 picks 128-wide tiles by default, 256-wide under gemm_cfg = 3
        default_tile = lib.cd360_gemm_tile_n(3072, 1280)
        lib.cd360_query_stream(ctypes.c_void_p(h1))
        forced_tile = lib.cd360_gemm_tile_n(3072, 1280)
        seen = {}
        t = threading.Thread(target=lambda: seen.update(other=lib.cd360_gemm_tile_n(3072, 1280)))
        t.start(); t.join()
        lib.cd360_query_stream(ctypes.c_void_p(h2))  # a stream without override: the default again
        assert (default_tile, forced_tile, seen["other"], lib.cd360_gemm_tile_n(3072, 1280)) == (128, 256, 128, 128)
        lib.cd360_query_stream(None)
        # capacity: 16 streams, the 17th is refused, a known stream can still be updated
        extra = [0x10000 + 16 * i for i in range(15)]
        for h in extra:
            _lib.set_stream_tuning(h, gemm_cfg=2)
        t17 = _lib.Tuning()
        lib.cd360_get_tuning(ctypes.byref(t17))
        assert lib.cd360_set_stream_tuning(ctypes.c_void_p(0x99999), ctypes.byref(t17)) == -2
        _lib.set_stream_tuning(h1, gemm_cfg=1)
        assert _lib.get_stream_tuning(h1)["gemm_cfg"] == 1
        bad = _lib.Tuning()
        bad.size = 4
        assert lib.cd360_set_stream_tuning(ctypes.c_void_p(h1), ctypes.byref(bad)) == -1  # ABI size check
        # conv_kgroup is the K order weights are PACKED in: a pack-time, process-wide choice that a stream cannot override (the packer
        # runs on no stream in particular) -- refused per stream, and the shape query answers from the process default whatever the context
        kg = _lib.Tuning()
        lib.cd360_get_tuning(ctypes.byref(kg))
        kg.conv_kgroup = 2
        assert lib.cd360_set_stream_tuning(ctypes.c_void_p(h1), ctypes.byref(kg)) == -1
        lib.cd360_query_stream(ctypes.c_void_p(h1))
        assert lib.cd360_conv_k_order(1280, 9) == 5
        lib.cd360_query_stream(None)
    finally:
        for h in [h1, h2] + [0x10000 + 16 * i for i in range(15)]:
            _lib.clear_stream_tuning(h)
    assert _lib.get_stream_tuning(h1) == base and _lib.get_tuning() == base and not _lib._stream_overrides
    assert lib.cd360_gemm_tile_n(3072, 1280) == 128


def test_host_glue_builds_and_binds_to_the_loaded_library():
    """csrc_host/cd360_host.cpp (the C++ autograd node of the fine-tuning Linears) is host glue above the C ABI: it builds with g++ against
    torch's headers (no device code), resolves its launches in the libcd360_hip.so cd360._lib has loaded, and exposes `linear`.  No compute
    call here (no GPU)."""
    import __graft_entry__ as G
    G._build_host_glue()
    from cd360 import _host
    mod = _host.get()
    assert mod is not None and callable(mod.linear) and callable(mod.init)
    with pytest.raises(Exception):
        mod.init("/nonexistent/libcd360_hip.so")  # never a second copy of the library, never a fallback


def test_generated_gemm_loop_is_what_its_generator_writes():
    """csrc/gemm4w_loop.inc (the instruction stream of the four-wave 256 x 256 GEMM arrangement) is generated text: the committed file must be
    exactly what tools/gen_gemm4w_loop.py renders, and the stream must hold what its schedule promises -- per K-tile 64 MFMAs, 32 fragment
    reads and 16 LDS-DMA pieces, five tiles per trip of the ring."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_gemm4w_loop", os.path.join(ROOT, "tools", "gen_gemm4w_loop.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(os.path.join(ROOT, "custom-diffusion360_amd", "csrc", "gemm4w_loop.inc")) as f:
        assert f.read() == gen.render()
    body = gen.body(0)
    loop = body[body.index("1:"):body.index("9:")]
    assert sum(1 for x in loop if x.startswith("v_mfma_f32_32x32x16_bf16")) == 5 * 64
    assert sum(1 for x in loop if x.startswith("ds_read_b128")) == 5 * 32
    assert sum(1 for x in loop if re.match(r"buffer_load_dwordx4 .* lds$", x)) == 5 * 16
    assert sum(1 for x in loop if x == "s_barrier") == 5
    for u in range(5):  # every ring slot is written by exactly the tiles that own it: operand tile n lives in slot n % 5
        t = gen.tile(u, 0)
        slots = sorted({int(m.group(1), 16) // gen.SLOT for x in t for m in [re.match(r"s_add_u32 m0, %18, (0x[0-9a-f]+)", x)] if m})
        assert slots == sorted({(2 * u + 3) % 5, (2 * u + 4) % 5, (2 * u) % 5})


def test_no_kernel_keeps_a_stack_object_in_scratch_memory():
    """build() records hipcc's per-kernel resource remarks next to every object (lib/obj/*.resources.json).  A kernel whose scratch size is
    not explained by register spills holds a STACK OBJECT in memory -- e.g. loop counters captured by a lambda that the optimiser failed to
    promote: one scratch load + store + s_waitcnt vmcnt(0) per iteration, invisible to every parity test (round 6: every tap-shifted
    convolution ran at half speed behind such a counter).  None may; and VGPR spills stay confined to the kernels listed below."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "custom-diffusion360_amd", "lib", "obj", "*.resources.json")))
    if not files:
        pytest.skip("no resource records: the library was not built by __graft_entry__.build() in this tree")
    known_spills = ("ILi6ELi2ELi5ELi1ELi2ELi1ELi4ELi5E", "ILi4ELi2ELi2ELi2ELi2ELi1ELi4ELi4E", "ILi4ELi2ELi2ELi2ELi2ELi1ELi4ELi7E",  # gemm_movers = 4 forced on tilings that do not take movers by default
                    "nerf_fused_rec_kernelILi2E")  # one 8-byte value at the 256-register cap, stored before and reloaded behind the view loop
    seen = 0
    for f in files:
        for name, r in json.load(open(f)).items():
            seen += 1
            spills = r.get("vgpr_spill", 0) + r.get("sgpr_spill", 0)
            assert r.get("scratch", 0) == 0 or spills > 0, f"{os.path.basename(f)}: {name} keeps {r['scratch']} bytes per lane in scratch without a register spill: a stack object in memory"
            assert r.get("vgpr_spill", 0) == 0 or any(k in name for k in known_spills), f"{os.path.basename(f)}: {name} spills {r['vgpr_spill']} VGPRs"
    assert seen > 100
