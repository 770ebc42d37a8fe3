"""Drop-in for the DC3D part of the reference's flat module `dram/models.py` on MI355X.

`models.DC3D`, `models.HeNorm`, `models.Initializer` and
`models.pooling_dense_features` keep the reference's signatures, attribute
names and state-dict keys (86 entries for the st_dram_ref config), so
`get_callable_by_name("models.DC3D")(**MODEL)` in the reference's JobRunner.init
(dram/job_runner.py:362-366), `model.init(HeNorm(...))`, the loss's
`model(images, lobes)` call (dram/metrics.py:362) and checkpoints keep working.
All compute runs on libdram_hip.so.

Out of scope of this build: `PCM` / `DC3DATGeneric`
(DGL graph attention) are not provided.

Reference citations are to /root/reference/dram/models.py.
"""
from parts import *  # noqa: F401,F403  (the reference relies on this star import, models.py:5)
from parts import ConvBlock5d, ConvPoolBlock5d, UpsampleConvBlock5d, checkpoint, nn, torch

from dram_amd import functional as HF
from dram_amd.modules import HipConv3d, HipUpsample


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
        raise NotImplementedError("pooling_method='global_max' is not implemented on the HIP path")
    lungs_expand = lungs.expand(B, 1, *dense_outs.shape[2:]) if lungs.shape[1] == 1 else None
    if lungs_expand is None:
        raise ValueError("pooling_dense_features: lungs must be a [B,1,D,H,W] mask")
    return HF.masked_mean(dense_outs, lungs_expand.to(dense_outs.dtype))


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

    def forward(self, x, lungs=None):
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
