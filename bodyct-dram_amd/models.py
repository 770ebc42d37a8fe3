"""Drop-in for the DC3D part of the reference's flat module `dram/models.py` on MI355X.

`models.DC3D`, `models.HeNorm`, `models.Initializer` and
`models.pooling_dense_features` keep the reference's signatures, attribute
names and state-dict keys (86 entries for the st_dram_ref config), so
`get_callable_by_name("models.DC3D")(**MODEL)` in the reference's JobRunner.init
(dram/job_runner.py:362-366), `model.init(HeNorm(...))`, the loss's
`model(images, lobes)` call (dram/metrics.py:362) and checkpoints keep working.
All compute runs on libdram_hip.so.

`models.PCM` / `models.DC3DATGeneric` (what process_pipeline.py loads) are
provided with the reference's constructor signatures and state-dict keys; the DGL neighbour graph of
the reference (networkx + dgl, absent here) is replaced by the equivalent voxel-grid stencil kernels
(csrc/pcm.hip).  Parity of the attention itself is unpinned (no DGL to run the reference against): see
oracle/dram_oracle.py.

Reference citations are to /root/reference/dram/models.py.
"""
from parts import *  # noqa: F401,F403  (the reference relies on this star import, models.py:5)
from parts import ConvBlock5d, ConvPoolBlock5d, Identity, UpsampleConvBlock5d, checkpoint, functools, nn, np, torch

from dram_amd import engine as _engine
from dram_amd import functional as HF
from dram_amd.modules import HipBatchNorm3d, HipConv3d, HipReLU, HipUpsample, run_conv_stack


class Initializer:

    def initialize(self, module):
        raise NotImplementedError("need subclassing to implement.")


class HeNorm(Initializer):
    """He-normal conv weights, conv bias 0.01, norm weight 1 / bias 0 (models.py:17-35)."""

    def __init__(self, **kwargs):
        self.mode = kwargs.get('mode', 'fan_in')

    def initialize(self, module):
        convs = (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)

        def init_weights(m):
            if isinstance(m, convs):
                torch.nn.init.kaiming_normal_(m.weight, mode=self.mode)
                if m.bias is not None:
                    m.bias.data.fill_(0.01)
            elif isinstance(m, (nn.BatchNorm3d, nn.GroupNorm)):
                if m.weight is not None:   # affine=False variants ("bntna", "lnna") have no parameters
                    m.weight.data.fill_(1)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.reset_parameters()

        module.apply(init_weights)


def pooling_dense_features(dense_outs, lungs, pooling_method='avg'):
    """models.py:37-49.  The default branch (lobe-masked mean) is a HIP reduction; the two
    global poolings reduce a [B,C,*] tensor to [B,C] (mean = masked mean with an all-ones mask)."""
    B, C = dense_outs.shape[0], dense_outs.shape[1]
    if pooling_method == 'global_avg':
        ones = torch.ones((B, 1) + tuple(dense_outs.shape[2:]), dtype=dense_outs.dtype, device=dense_outs.device)
        return HF.masked_mean(dense_outs, ones).view(B, C)
    if pooling_method == 'global_max':
        return HF.global_max(dense_outs)
    # lungs.expand_as(dense_outs) (models.py:45): a [B,1,...] mask broadcasts over the channels inside the
    # kernel; a [B,C,...] mask (one mask per channel) is reduced channel by channel
    if lungs.shape[1] == 1:
        mask = lungs.expand(B, 1, *dense_outs.shape[2:]).to(dense_outs.dtype)
        return HF.masked_mean(dense_outs, mask)
    if tuple(lungs.shape) != tuple(dense_outs.shape):
        raise ValueError(f"pooling_dense_features: lungs {tuple(lungs.shape)} does not expand to {tuple(dense_outs.shape)}")
    cols = [HF.masked_mean(dense_outs[:, c:c + 1].contiguous(), lungs[:, c:c + 1].to(dense_outs.dtype).contiguous())
            for c in range(C)]
    return torch.cat(cols, dim=1)


class DC3D(nn.Module):
    """3-D U-Net regression network (models.py:52-147): n_layers ConvPoolBlock5d down, one
    ConvBlock5d bottleneck, up to n_layers UpsampleConvBlock5d up (`stacking` of them are used),
    a 1x1x1 head and a final trilinear resize to the input size."""

    def __init__(self, n_layers, in_ch_list, base_ch_list,
                 end_ch_list, out_ch, padding_list,
                 checkpoint_layers, dropout,
                 upsample_ksize=3, upsample_sf=2, kernel_sizes=None, stacking=0,
                 norm_method="bn", act_method='relu', pooling_method='avg', out_cls_ch=6):
        super(DC3D, self).__init__()
        self.dropout = dropout
        self.n_layers = n_layers
        self.padding_list = padding_list
        self.in_ch_list = in_ch_list
        self.base_ch_list = base_ch_list
        self.kernel_sizes = [3] * (n_layers * 2 + 1) if kernel_sizes is None else kernel_sizes
        self.end_ch_list = end_ch_list
        self.upsample_ksize = upsample_ksize
        self.upsample_sf = upsample_sf
        self.checkpoint_layers = checkpoint_layers
        self.norm_method = norm_method
        assert (len(end_ch_list) == len(base_ch_list) == len(in_ch_list) == len(padding_list))
        self.out_ch = out_ch
        self.stacking = stacking
        self.out_cls_ch = out_cls_ch
        self.pooling_method = pooling_method
        conv_bias = self.norm_method is None   # models.py:78

        def level(k):
            return ([in_ch_list[k], base_ch_list[k]], [base_ch_list[k], end_ch_list[k]],
                    checkpoint_layers[k], self.kernel_sizes[k], padding_list[k])

        down = []
        for n in range(n_layers):
            cin, cout, ck, ks, pad = level(n)
            down.append(ConvPoolBlock5d(cin, cout, ck, ks, conv_bias, pad, 2, 2, 0,
                                        norm_method=norm_method, act_method=act_method, dropout=dropout))
        self.ds_modules = nn.ModuleList(down)
        cin, cout, ck, ks, pad = level(n_layers)
        self.bg = ConvBlock5d(cin, cout, ck, ks, conv_bias, pad, dropout,
                              norm_method=norm_method, act_method=act_method)
        if (n_layers + 1) < len(in_ch_list):
            up = []
            for n in range(n_layers):
                cin, cout, ck, ks, pad = level(n_layers + 1 + n)
                up.append(UpsampleConvBlock5d(cin, cout, ck, self.upsample_sf, ks, conv_bias, pad,
                                              norm_method=norm_method, act_method=act_method, dropout=dropout))
            self.us_modules = nn.ModuleList(up)
        else:
            self.us_modules = None
        self.top_layer = HipConv3d(end_ch_list[n_layers + stacking], out_ch, kernel_size=1, padding=0)
        self.dummy = torch.ones(1, requires_grad=True)   # plain attribute: stays on the CPU (models.py:111)
        self.trace_path = None                           # written by the loss every step (metrics.py:202)

    def init(self, initializer):
        initializer.initialize(self)

    def pooling_dense_features(self, dense_outs, lungs, pooling_method='avg'):
        return pooling_dense_features(dense_outs, lungs, pooling_method)

    # How `checkpoint_layers` flags are honoured (models.py:122-143 wraps flagged blocks in
    # torch.utils.checkpoint):
    #   "stats"     (default) no recomputation -- with 288 GB of HBM the saved activations of a
    #               16 x 128^3 micro-batch fit -- but the one observable side effect of the
    #               reference's reentrant checkpoint is reproduced: a flagged block's BatchNorm
    #               running statistics are updated twice per training step, because its
    #               forward runs again during backward on the same batch.
    #   "recompute" torch.utils.checkpoint(use_reentrant=True) exactly like the reference
    #               (saves ~2x activation memory, costs +0.67 forward passes of convolutions).
    checkpoint_mode = "stats"

    def _run(self, flag, block, *tensors):
        if flag <= 0:
            return block(*tensors)
        if self.checkpoint_mode == "recompute":
            return checkpoint(block, *tensors, use_reentrant=True)
        # the reference re-runs the block in backward only if autograd records it
        twice = self.training and torch.is_grad_enabled() and any(
            isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)
        if not twice:
            return block(*tensors)
        norms = [m for m in block.modules() if isinstance(m, nn.BatchNorm3d)]
        for m in norms:
            m.stat_updates = 2
        try:
            return block(*tensors)
        finally:
            for m in norms:
                m.stat_updates = 1

    # `fused` (default): the forward/backward of a standard network (every stage conv3x3x3 -> BatchNorm/GroupNorm ->
    # ReLU) runs through dram_amd/engine.py: norm statistics in the conv epilogue, normalise + ReLU applied on load by
    # every consumer, one autograd node for the whole network.  Other networks, checkpoint_mode="recompute" and
    # fused=False take the per-op path below (one autograd Function per op of the reference).  Same values either way.
    fused = True

    def forward(self, x, lungs=None):
        if self.fused and self.checkpoint_mode != "recompute" and _engine.supports(self):
            dense_outs = _engine.run(self, x)
            return dense_outs, dense_outs
        L = self.n_layers
        skips = []
        cur = x
        for idx, ds in enumerate(self.ds_modules):
            if self.checkpoint_layers[idx] > 0 and idx == 0:
                feat, cur = self._run(1, ds, cur, self.dummy)
            else:
                feat, cur = self._run(self.checkpoint_layers[idx], ds, cur)
            skips.append(feat)
        cur = self._run(self.checkpoint_layers[L], self.bg, cur)
        if self.us_modules is not None:
            for idx, (us, skip) in enumerate(zip(self.us_modules, reversed(skips))):
                if self.stacking == idx:
                    break
                # NB: the flag index is n_layers + idx, not n_layers + 1 + idx (models.py:140)
                cur = self._run(self.checkpoint_layers[L + idx], us, cur, skip)
        dense_outs = self.top_layer(cur)
        if tuple(dense_outs.shape[-3:]) != tuple(x.shape[-3:]):
            dense_outs = HipUpsample(size=tuple(x.shape[-3:]), mode='trilinear', align_corners=True)(dense_outs)
        # (equal sizes: align_corners resampling is the identity -- scale (in-1)/(out-1) = 1 -- so the
        #  pass is skipped; models.py:146 always calls nn.Upsample)
        return dense_outs, dense_outs


def _resize(x, size):
    """F.interpolate(x, size=size, mode='trilinear', align_corners=True); the exact identity is skipped."""
    size = tuple(int(v) for v in size)
    if tuple(x.shape[-3:]) == size:
        return x
    return HF.upsample_trilinear_ac(x, size=size)


def _channel_linear(lin, x):
    """nn.Linear applied to the channel axis of [B,C,D,H,W] (the reference flattens to [nodes*batch, C]
    rows, models.py:381-393) = a 1x1x1 convolution with the same weight."""
    if not isinstance(lin, nn.Linear):
        return x            # parts.Identity (f_dim / g_dim <= 0, models.py:169-183)
    return HF.conv3d_k1(x, lin.weight.view(lin.out_features, lin.in_features, 1, 1, 1), lin.bias)


class PCM(nn.Module):
    """Local (k_size^3 neighbourhood) attention that refines a class-activation map `cam` with feature
    affinities (models.py:150-411).  Same constructor, attributes and parameter names (theta, phi, G, r,
    geo_theta, geo_phi) as the reference.

    The reference materialises the neighbourhood as a graph (init_graph, models.py:221-258) and lets
    DGL gather mailboxes; on a voxel grid that graph is a stencil of E offsets -- the
    `connectivity` structuring element zoomed to k_size, minus the centre when self_loop is False,
    clipped at the grid border (side nodes have fewer edges) -- and the forward is two kernels:
    attention weights per (node, offset) and their weighted aggregation (csrc/pcm.hip).

    Implemented merge types: the dot-product family `dram_amd.functional.PCM_MERGE_MODES`
    (incl. the shipped 'scaled_dot_product_relu'), the geo family `PCM_GEO_MERGES` (an appearance term
    plus a term over sin/cos positional encodings, `build_geo_feature`) and the sum-normalised
    `PCM_SUM_MERGES` (cosine, heu1, heu2) and 'l2' for f_dim == 1 (the one width for which the reference's broadcast
    of [.., 1, f_dim] against [.., f_dim, edges] and the reshape at models.py:396 are defined; other widths raise
    ValueError).  Unknown names raise NotImplementedError at call time, like the reference.

    Deliberate numerical deviation ('l2' only): the reference evaluates exp(-5 (theta - phi_e)^2) / sum_e exp(...)
    (models.py:262-264) literally, which is 0 / 0 = NaN for a node all of whose edges underflow (|theta - phi| > ~4.2 in fp32).
    Here the same ratio is computed as a softmax over the edges of 10 theta phi_e - 5 phi_e^2 (the common -5 theta^2 cancels,
    the maximum is subtracted): algebraically identical, finite for saturated inputs
    (tests/test_gpu_pcm.py::test_pcm_l2_merge_saturated_inputs_stay_finite).  Like all of PCM: parity unpinned (DGL absent)."""

    def __init__(self, pool_size, in_ch, g_ch, f_dim, geo_f_dim, g_dim, non_local_iter, k_size,
                 merge_type='l2', self_loop=True, connectivity=2, residual=False, p_enc_dim=32):
        super(PCM, self).__init__()
        self.in_ch = in_ch
        self.g_ch = g_ch
        self.f_dim = f_dim
        self.g_dim = g_dim
        self.pool_size = pool_size
        self.merge_type = merge_type
        self.self_loop = self_loop
        self.non_local_iter = non_local_iter
        self.k_size = k_size
        self.connectivity = connectivity
        self.residual = residual
        self.p_enc_dim = p_enc_dim
        self.geo_f_dim = geo_f_dim
        if self.g_dim > 0:
            self.G = nn.Linear(g_ch, g_dim)
            self.r = nn.Linear(g_dim, g_ch)
        else:
            self.G = Identity()
            self.r = Identity()
            self.g_dim = g_ch
        if f_dim > 0:
            self.theta = nn.Linear(in_ch, f_dim)
            self.phi = nn.Linear(in_ch, f_dim)
        else:
            self.theta = Identity()
            self.phi = Identity()
            self.f_dim = in_ch
        if self.p_enc_dim > 0:      # parameters kept for state-dict parity; only the geo merge types read them
            if geo_f_dim > 0:
                self.geo_theta = nn.Linear(p_enc_dim, geo_f_dim)
                self.geo_phi = nn.Linear(p_enc_dim, geo_f_dim)
            else:
                self.geo_theta = Identity()
                self.geo_phi = Identity()
                self.geo_f_dim = p_enc_dim
        self.graph = None           # the reference caches its DGLGraph here; we cache the offset list
        self._geo_cache = {}

    def build_geo_feature(self, x):
        """models.py:194-219: sin/cos positional encoding of the voxel coordinates, p_enc_dim/3 channels per axis
        (frequencies 1e-4^(2i/d)), the same for every sample; a constant of the grid, built once per shape with the
        reference's own torch expressions on the host and kept on the device."""
        spatial = tuple(int(v) for v in x.shape[-3:])
        key = (spatial, x.device)
        if key not in self._geo_cache:
            if self.p_enc_dim % (2 * len(spatial)) != 0:
                raise ValueError("Cannot use sin/cos positional encoding with "
                                 "odd dimension (got dim={:d})".format(self.p_enc_dim))
            p = torch.ones(spatial).nonzero().float().view(*spatial, len(spatial))
            pe = torch.zeros(self.p_enc_dim, *spatial)
            d_model = int(self.p_enc_dim / len(spatial))
            div = torch.pow(1e-4, torch.arange(0., d_model, 2) / d_model)
            for d in range(len(spatial)):
                start, end = d * d_model, (d + 1) * d_model
                arg = p[..., d].expand(len(div), *spatial) * div.view(len(div), *([1] * len(spatial)))
                pe[start:end:2] = torch.sin(arg)
                pe[(start + 1):end:2] = torch.cos(arg)
            self._geo_cache[key] = pe.unsqueeze(0).to(x.device)
        return self._geo_cache[key].expand(x.shape[0], -1, *spatial)

    def init_graph(self, spatial_size=None, k_size=None):
        """Neighbour offsets (dz,dy,dx) of models.py:230-232 (+ remove_self_loop, 257-258)."""
        k = self.k_size if k_size is None else k_size
        base = np.zeros((3, 3, 3), dtype=bool)      # ndimage.generate_binary_structure(3, connectivity)
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    base[dz + 1, dy + 1, dx + 1] = abs(dz) + abs(dy) + abs(dx) <= max(1, self.connectivity)
        if k != 3:
            from scipy import ndimage              # nearest-neighbour zoom exactly as the reference calls it
            base = ndimage.zoom(base, k / 3.0, order=0)
        off = np.asarray(np.where(base > 0)).T - np.asarray([k // 2] * 3)
        if not self.self_loop:
            off = off[np.any(off != 0, axis=1)]
        return tuple(tuple(int(v) for v in o) for o in off)

    def forward(self, cam, f, args=None):
        if self.graph is None:
            self.graph = self.init_graph(self.pool_size, self.k_size)
        offsets = self.graph
        theta = _channel_linear(self.theta, f)
        phi = _channel_linear(self.phi, f)
        geo_theta = geo_phi = None
        if self.merge_type in HF.PCM_GEO_MERGES:
            if self.p_enc_dim <= 0:
                raise ValueError(f"PCM merge_type {self.merge_type!r} needs p_enc_dim > 0")
            geo = self.build_geo_feature(f).contiguous()        # detached by construction (models.py:343)
            geo_theta = _channel_linear(self.geo_theta, geo)
            geo_phi = _channel_linear(self.geo_phi, geo)
        attn = HF.pcm_attention(theta, phi, offsets, self.merge_type, geo_theta, geo_phi)   # f is fixed over the iterations
        for _ in range(self.non_local_iter):
            y = HF.pcm_aggregate(attn, _channel_linear(self.G, cam), offsets)
            refined_cam = _channel_linear(self.r, y)
            cam = refined_cam + cam if self.residual else refined_cam
        return cam


class DC3DATGeneric(nn.Module):
    """DC3D + PCM refinement of the dense output (models.py:413-597): `forward` returns
    (dense_outs, refined_dense_outs).  Same constructor, attribute and state-dict names as the
    reference (ds_modules / bg / us_modules / top_layer / reshape / attention_module)."""

    checkpoint_mode = "stats"       # see DC3D

    def __init__(self, n_layers, in_ch_list, base_ch_list,
                 end_ch_list, out_ch, padding_list,
                 checkpoint_layers, dropout, at_spatial_size, at_f_dim, at_g_dim, at_p_enc_dim, at_geo_f_dim,
                 at_g_iter, at_k_size, at_merge_type, at_self_loop, at_layers,
                 upsample_ksize=3, upsample_sf=2, kernel_sizes=None, stacking=3,
                 norm_method="bn", act_method='relu', pooling_method='avg', out_cls_ch=6):
        super(DC3DATGeneric, self).__init__()
        self.dropout = dropout
        self.n_layers = n_layers
        self.padding_list = padding_list
        self.in_ch_list = in_ch_list
        self.base_ch_list = base_ch_list
        self.at_spatial_size = at_spatial_size
        self.out_cls_ch = out_cls_ch
        self.kernel_sizes = [3] * (n_layers * 2 + 1) if kernel_sizes is None else kernel_sizes
        self.end_ch_list = end_ch_list
        self.upsample_ksize = upsample_ksize
        self.upsample_sf = upsample_sf
        self.checkpoint_layers = checkpoint_layers
        self.norm_method = norm_method
        assert (len(end_ch_list) == len(base_ch_list) == len(in_ch_list) == len(padding_list))
        self.out_ch = out_ch
        self.stacking = stacking
        self.pooling_method = pooling_method
        self.at_f_dim = at_f_dim
        self.at_g_dim = at_g_dim
        self.at_g_iter = at_g_iter
        self.at_k_size = at_k_size
        self.at_p_enc_dim = at_p_enc_dim
        self.at_geo_f_dim = at_geo_f_dim
        self.at_merge_type = at_merge_type
        self.at_self_loop = at_self_loop
        self.at_layers = at_layers
        conv_bias = self.norm_method is None

        def level(k):
            return ([in_ch_list[k], base_ch_list[k]], [base_ch_list[k], end_ch_list[k]],
                    checkpoint_layers[k], self.kernel_sizes[k], padding_list[k])

        down = []
        for n in range(n_layers):
            cin, cout, ck, ks, pad = level(n)
            down.append(ConvPoolBlock5d(cin, cout, ck, ks, conv_bias, pad, 2, 2, 0,
                                        norm_method=norm_method, act_method=act_method, dropout=dropout))
        self.ds_modules = nn.ModuleList(down)
        cin, cout, ck, ks, pad = level(n_layers)
        self.bg = ConvBlock5d(cin, cout, ck, ks, conv_bias, pad, dropout, norm_method=norm_method, act_method=act_method)
        up = []
        for n in range(n_layers):
            cin, cout, ck, ks, pad = level(n_layers + 1 + n)
            up.append(UpsampleConvBlock5d(cin, cout, ck, self.upsample_sf, ks, conv_bias, pad,
                                          norm_method=norm_method, act_method=act_method, dropout=dropout))
        self.us_modules = nn.ModuleList(up)
        self.top_layer = HipConv3d(end_ch_list[n_layers + stacking], out_ch, kernel_size=1, padding=0)
        n_at_in_ch = at_f_dim * (len(at_layers) - 1) + 1 if -1 in at_layers else at_f_dim * len(at_layers)
        self.reshape = nn.ModuleList([
            nn.Sequential(HipConv3d(end_ch_list[l_id], at_f_dim, kernel_size=1, padding=0, stride=1),
                          HipBatchNorm3d(at_f_dim), HipReLU(inplace=True))
            for l_id in at_layers if l_id != -1])
        self.attention_module = PCM(at_spatial_size, n_at_in_ch, out_ch, at_f_dim, at_geo_f_dim, at_g_dim, at_g_iter,
                                    at_k_size, at_merge_type, at_self_loop, p_enc_dim=at_p_enc_dim)
        self.dummy = torch.ones(1, requires_grad=True)
        self.trace_path = None      # accepted and ignored: the per-scan heat-map dumps (models.py:508-539) are not built
        self.n_pcm_layer = 0

    def init(self, initializer):
        initializer.initialize(self)

    def pooling_dense_features(self, dense_outs, lungs, pooling_method='avg'):
        return pooling_dense_features(dense_outs, lungs, pooling_method)

    _run = DC3D._run
    us_flag_offset = 1              # checkpoint flag of up-block i: n_layers + 1 + i (models.py:573; DC3D: n_layers + i)
    # `fused` (default): the U-Net proper runs through dram_amd/engine.py like DC3D (one autograd node, norm statistics in the
    # conv epilogue, normalise + ReLU on load), the engine writes out the feature maps the attention module taps, and the
    # reshape convs, resizes and PCM run on their own kernels behind it.
    fused = True

    def apply_attention(self, x, lungs, dense_out, attention_features):
        """models.py:498-506: resize the dense map to the attention grid, refine, resize back."""
        refined = self.attention_module(_resize(dense_out, self.at_spatial_size), attention_features)
        return _resize(refined, dense_out.shape[2:])

    def forward(self, x, lungs=None):
        L = self.n_layers
        feats = [x] if -1 in self.at_layers else []
        nc = 0

        def tap(t):     # models.py:556,566,578: 1x1x1 conv + BN + ReLU on the *detached* feature map
            nonlocal nc
            feats.append(run_conv_stack([self.reshape[nc]], t.detach()))
            nc += 1

        if self.fused and self.checkpoint_mode != "recompute" and _engine.supports(self):
            n_up = self.stacking if 0 <= self.stacking < len(self.us_modules) else len(self.us_modules)   # (the loop below breaks at `stacking`)
            ids = [l for l in range(L + 1 + n_up) if l in self.at_layers]
            res = _engine.run(self, x, taps=ids)
            top, tapped = res if ids else (res, {})
            for l in ids:           # ascending layer id = the order in which the reference's forward meets them
                tap(tapped[l])
            dense_outs = top
            feats = [_resize(t, self.at_spatial_size) for t in feats]
            attention_features = functools.reduce(HF.crop_concat, feats)
            refined_dense_outs = self.apply_attention(x, lungs, dense_outs, attention_features)
            return dense_outs, refined_dense_outs
        skips = []
        cur = x
        for idx, ds in enumerate(self.ds_modules):
            if self.checkpoint_layers[idx] > 0 and idx == 0:
                feat, cur = self._run(1, ds, cur, self.dummy)
            else:
                feat, cur = self._run(self.checkpoint_layers[idx], ds, cur)
            skips.append(feat)
            if idx in self.at_layers:
                tap(feat)
        cur = self._run(self.checkpoint_layers[L], self.bg, cur)
        if L in self.at_layers:
            tap(cur)
        for idx, (us, skip) in enumerate(zip(self.us_modules, reversed(skips))):
            if self.stacking == idx:
                break
            cur = self._run(self.checkpoint_layers[L + 1 + idx], us, cur, skip)     # models.py:573 (+1, unlike DC3D)
            if L + idx + 1 in self.at_layers:
                tap(cur)
        dense_outs = _resize(self.top_layer(cur), x.shape[-3:])
        feats = [_resize(t, self.at_spatial_size) for t in feats]
        attention_features = functools.reduce(HF.crop_concat, feats)      # torch.cat(dim=1): equal sizes, crop offset 0
        refined_dense_outs = self.apply_attention(x, lungs, dense_outs, attention_features)
        return dense_outs, refined_dense_outs
