"""Model configurations of the DRAM hot path, as plain dicts for `models.DC3D(**cfg)` /
`models.DC3DATGeneric(**cfg)`.

`ST_DRAM_REF_MODEL` / `ST_DRAM_REF_ATT_MODEL` restate the `MODEL` dicts of the reference's shipped
settings (dram/exp_settings/st_dram_ref.py:54-71, st_dram_ref_att.py:56-82) without the
`method` key that JobRunner.init pops (dram/job_runner.py:362-363).  `SLIM` / `SLIM_ATT` are the same
topology and flags with channels / 8: the sizes the parity fixtures under tests/golden/ were generated at.
The oracle keeps its own copies (oracle/ is test infrastructure and is not imported from here);
tests/test_host_cpu.py checks that the two agree.
"""

ST_DRAM_REF_MODEL = {
    "n_layers": 3,
    "in_ch_list": [1, 64, 128, 256, 768, 384, 192],
    "base_ch_list": [32, 64, 128, 256, 256, 128, 64],
    "end_ch_list": [64, 128, 256, 512, 256, 128, 64],
    "kernel_sizes": [(3, 3)] * 7,
    "stacking": 3,
    "padding_list": [(1, 1)] * 7,
    "checkpoint_layers": [0, 1, 0, 1, 0, 1, 0],
    "dropout": 0.0,
    "upsample_ksize": (3, 3, 3),
    "upsample_sf": (2, 2, 2),
    "out_ch": 1,
}

ST_DRAM_REF_ATT_MODEL = dict(ST_DRAM_REF_MODEL, **{
    "at_spatial_size": (64, 64, 64), "at_f_dim": 8, "at_g_dim": 8, "at_g_iter": 1, "at_k_size": 3,
    "at_merge_type": "scaled_dot_product_relu", "at_self_loop": False, "at_layers": [-1, 0, 1],
    "at_p_enc_dim": 0, "at_geo_f_dim": 0,
})

SLIM = {
    "n_layers": 3,
    "in_ch_list": [1, 8, 16, 32, 96, 48, 24],
    "base_ch_list": [4, 8, 16, 32, 32, 16, 8],
    "end_ch_list": [8, 16, 32, 64, 32, 16, 8],
    "kernel_sizes": [(3, 3)] * 7,
    "stacking": 3,
    "padding_list": [(1, 1)] * 7,
    "checkpoint_layers": [0, 1, 0, 1, 0, 1, 0],
    "dropout": 0.0,
    "upsample_ksize": (3, 3, 3),
    "upsample_sf": (2, 2, 2),
    "out_ch": 1,
}

SLIM_ATT = dict(SLIM, at_spatial_size=(6, 5, 7), at_f_dim=4, at_g_dim=3, at_g_iter=1, at_k_size=3,
                at_merge_type="scaled_dot_product_relu", at_self_loop=False, at_layers=[-1, 0, 1],
                at_p_enc_dim=0, at_geo_f_dim=0)
