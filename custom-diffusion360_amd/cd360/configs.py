"""The reference's network_config for the UNet (configs/train_co3d_concept.yaml:27-54), values verbatim: what sample.py / main.py hand to
`instantiate_from_config` for `model.params.network_config`.  bench.py, the tools and the tests build the SDXL UNet from it."""

SDXL_NETWORK_CONFIG = {
    "target": "sgm.modules.diffusionmodules.openaimodel.UNetModel",
    "params": dict(adm_in_channels=2816, num_classes="sequential", use_checkpoint=False, in_channels=4, out_channels=4, model_channels=320,
                   attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=64,
                   use_linear_in_transformer=True, transformer_depth=[1, 2, 10], context_dim=2048,
                   spatial_transformer_attn_type="softmax-xformers", image_cross_blocks=[0, 2, 4, 6, 8, 10], rgb=True, far=2,
                   num_samples=24, not_add_context_in_triplane=False, rgb_predict=True, add_lora=False, average=False,
                   use_prev_weights_imp_sample=True, stratified=True, imp_sampling_percent=0.9),
}
