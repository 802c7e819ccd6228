"""Tiny model configurations used for golden vectors (factory-kwargs overrides of the shipped
ViT-B/32 + text transformer + FDT config, example/clip_fdt/config_cc3m.yaml:1-21)."""

CFG = {
    # head_dim is 64 everywhere (as in ViT-B/32, ViT-L/14 and both text towers)
    "a": dict(width=128, heads=2, layers=2, res=64, patch=32, embed_dim=32,
              t_width=128, t_heads=2, t_layers=2, ctx=77, sd_num=128, sd_dim=32, batch=4),
    # non power-of-two everything: 3 heads, 9 patches, 24-token context, 320 codes
    "b": dict(width=192, heads=3, layers=1, res=96, patch=32, embed_dim=64,
              t_width=64, t_heads=1, t_layers=2, ctx=24, sd_num=320, sd_dim=64, batch=3),
    # ViT-L/14 geometry in small: 14x14 patches (588-long rows, not a multiple of 8), 8x8+1 = 65 tokens, wider text tower
    "c": dict(width=128, heads=2, layers=1, res=112, patch=14, embed_dim=64,
              t_width=192, t_heads=3, t_layers=1, ctx=16, sd_num=256, sd_dim=64, batch=3),
}

FDT_VARIANTS = [
    # (att_func, pool, temperature, logit_scale or None for ln(1/0.07))
    ("sparsemax", "max", 1000.0, None),     # shipped config
    ("sparsemax", "max", 1.0, None),
    ("sparsemax", "mean", 1.0, None),
    ("sparsemax", "sum", 1000.0, None),
    ("softmax", "max", 1.0, None),
    ("softmax", "max", 1000.0, None),
    ("softmax", "mean", 1.0, None),
    ("softmax", "sum", 1.0, None),
    ("sparsemax", "max", 1000.0, 5.0),      # exp(5) > 100: exercises the clamp of logit_scale.exp().data
    ("sigmoid", "max", 1000.0, None),       # round 3: att_func_type 'sigmoid' (weighted sum divided by the weights' row sum)
    ("sigmoid", "mean", 1.0, None),
]


def variant_key(v):
    return "%s_%s_T%g_ls%s" % (v[0], v[1], v[2], "d" if v[3] is None else ("%g" % v[3]))


def model_kwargs(c, fdt=None, bpe_path=None):
    """kwargs for model_entry(type=clip_fdt_vitb32 | clip_vitb32)."""
    kw = dict(
        image_encode=dict(embed_dim=c["embed_dim"], layers=c["layers"], width=c["width"], heads=c["heads"],
                          input_resolution=c["res"], patch_size=c["patch"]),
        text_encode=dict(bpe_path=bpe_path, text_encode_type="Transformer",
                         text_model_utils=dict(random=False, freeze=False), embed_dim=c["embed_dim"],
                         context_length=c["ctx"], transformer_width=c["t_width"],
                         transformer_heads=c["t_heads"], transformer_layers=c["t_layers"]),
    )
    if fdt is not None:
        att_func, pool, temp, _ = fdt
        kw["fdt"] = dict(sd_temperature=temp, att_func_type=att_func, pool_type=pool, use_allgather=True,
                         sd_num=c["sd_num"], sd_dim=c["sd_dim"], raw_img_ft_dim=c["width"],
                         raw_txt_ft_dim=c["t_width"])
    else:
        kw["clip"] = dict(use_allgather=True)
    return kw


def oracle_cfg(c, fdt=None):
    d = dict(v_heads=c["heads"], t_heads=c["t_heads"])
    if fdt is not None:
        d.update(att_func=fdt[0], pool=fdt[1], temperature=fdt[2])
    return d


def state_shapes(c, fdt=True, vocab=49409):
    """Ordered {parameter name: shape} of the reference model built from `c`
    (same order as nn.Module.named_parameters() in the reference; checked against
    g7_param_groups.json for the real ViT-B/32 sizes)."""
    s = {}
    W, D, Wt = c["width"], c["embed_dim"], c["t_width"]
    grid = c["res"] // c["patch"]
    if fdt:
        s["space_dict"] = (c["sd_num"], c["sd_dim"])
        s["logit_scale"] = (1,)
        s["logit_scale_sd"] = (1,)
    else:
        s["logit_scale"] = (1,)

    def blocks(pre, w, n):
        for i in range(n):
            b = "%stransformer.resblocks.%d." % (pre, i)
            s[b + "attn.in_proj_weight"] = (3 * w, w)
            s[b + "attn.in_proj_bias"] = (3 * w,)
            s[b + "attn.out_proj.weight"] = (w, w)
            s[b + "attn.out_proj.bias"] = (w,)
            s[b + "ln_1.weight"] = (w,)
            s[b + "ln_1.bias"] = (w,)
            s[b + "mlp.c_fc.weight"] = (4 * w, w)
            s[b + "mlp.c_fc.bias"] = (4 * w,)
            s[b + "mlp.c_proj.weight"] = (w, 4 * w)
            s[b + "mlp.c_proj.bias"] = (w,)
            s[b + "ln_2.weight"] = (w,)
            s[b + "ln_2.bias"] = (w,)

    s["visual.class_embedding"] = (W,)
    s["visual.positional_embedding"] = (grid * grid + 1, W)
    s["visual.proj"] = (W, D)
    s["visual.conv1.weight"] = (W, 3, c["patch"], c["patch"])
    s["visual.ln_pre.weight"] = (W,)
    s["visual.ln_pre.bias"] = (W,)
    blocks("visual.", W, c["layers"])
    s["visual.ln_post.weight"] = (W,)
    s["visual.ln_post.bias"] = (W,)
    s["encode_text.positional_embedding"] = (c["ctx"], Wt)
    blocks("encode_text.", Wt, c["t_layers"])
    s["encode_text.token_embedding.weight"] = (vocab, Wt)
    s["encode_text.ln_final.weight"] = (Wt,)
    s["encode_text.ln_final.bias"] = (Wt,)
    s["encode_text.text_projection.weight"] = (D, Wt)
    s["encode_text.text_projection.bias"] = (D,)
    if fdt:
        for side, ft in (("img_query_model.", W), ("txt_query_model.", Wt)):
            s[side + "q_map.0.weight"] = (ft,)
            s[side + "q_map.0.bias"] = (ft,)
            s[side + "q_map.1.weight"] = (c["sd_dim"], ft)
            s[side + "q_map.1.bias"] = (c["sd_dim"],)
            s[side + "q_map.3.weight"] = (c["sd_dim"],)
            s[side + "q_map.3.bias"] = (c["sd_dim"],)
            s[side + "q_map.4.weight"] = (c["sd_dim"], c["sd_dim"])
            s[side + "q_map.4.bias"] = (c["sd_dim"],)
    return s


VITB32 = dict(width=768, heads=12, layers=12, res=224, patch=32, embed_dim=512,
              t_width=512, t_heads=8, t_layers=12, ctx=77, sd_num=4096, sd_dim=512, batch=256)
