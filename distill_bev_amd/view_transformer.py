"""LSS view transformers -- registry mirror of
``mmdet3d/models/necks/view_transformer_mine.py`` (ViewTransformerLiftSplatShoot :59-264,
ViewTransformerLSSBEVDepth :283-344) and of the bev_pool-based twins in
``view_transformer.py`` (OfficialViewTransformer*), all backed by the fused gfx950 lift-splat.

State-dict keys (frustum, dx, bx, nx, depthnet.*, featnet.*, extra_depthnet.*, dcn.*, se.*) and
constructor kwargs are the reference's.  The three reference splat variants (argsort+cumsum,
unique+scatter_sum, bev_pool extension) are the same segmented sum; all route to one HIP op.
"""
import torch
import torch.nn as nn

from . import dcn  # noqa: F401  (registers conv type 'DCNv2')
from . import lss as LSS
from .depth_head import depth_head
from .lift_splat import lift_splat, lift_splat_prepare, lift_splat_prepare_cam, voxel_pooling as _voxel_pooling
from .nets import SELikeModule
from .registry import MODELS, build_backbone, build_conv_layer


@MODELS.register_module()
class ViewTransformerLiftSplatShoot(nn.Module):
    def __init__(self, grid_config=None, data_config=None, numC_input=512, numC_Trans=64, downsample=16,
                 accelerate=True, max_drop_point_rate=0.0, use_bev_pool=True, **kwargs):
        super().__init__()
        if grid_config is None:
            grid_config = {"xbound": [-51.2, 51.2, 0.8], "ybound": [-51.2, 51.2, 0.8],
                           "zbound": [-10.0, 10.0, 20.0], "dbound": [1.0, 60.0, 1.0]}
        self.grid_config = grid_config
        dx, bx, nx = LSS.gen_dx_bx(grid_config["xbound"], grid_config["ybound"], grid_config["zbound"])
        self.dx = nn.Parameter(dx, requires_grad=False)
        self.bx = nn.Parameter(bx, requires_grad=False)
        self.nx = nn.Parameter(nx, requires_grad=False)
        # host copies: kernel launch parameters must not cost a device->host sync per step
        self._dx_host = [float(v) for v in dx]
        self._bx_host = [float(v) for v in bx]
        self._nx_host = [int(v) for v in nx.long()]
        if data_config is None:
            data_config = {"input_size": (256, 704)}
        self.data_config = data_config
        self.downsample = downsample
        self.frustum = nn.Parameter(LSS.create_frustum(data_config["input_size"], downsample,
                                                       tuple(grid_config["dbound"])), requires_grad=False)
        self.D = self.frustum.shape[0]
        self.numC_input = numC_input
        self.numC_Trans = numC_Trans
        self.depthnet = nn.Conv2d(numC_input, self.D + numC_Trans, kernel_size=1, padding=0)
        self.accelerate = accelerate
        self.max_drop_point_rate = max_drop_point_rate

    def get_depth_dist(self, x):
        return x.softmax(dim=1)

    def get_geometry(self, rots, trans, intrins, post_rots, post_trans):
        """vt_mine.py:111-139 (torch ops, same sequence)."""
        return LSS.get_geometry(self.frustum, rots, trans, intrins, post_rots, post_trans)

    def prepare(self, geom):
        """voxel index + cell->points CSR of a geometry tensor (shared by fwd and bwd)."""
        return lift_splat_prepare(geom, self._dx_host, self._bx_host, self._nx_host)

    def voxel_pooling(self, geom_feats, x):
        """vt_mine.py:141-181 call surface: geom f32[B,N,D,H,W,3], x f32[B,N,D,H,W,C] volume."""
        return _voxel_pooling(x, self.prepare(geom_feats))

    voxel_pooling_accelerated = voxel_pooling

    def lift_splat(self, geom, depth_prob, img_feat):
        """fused: == voxel_pooling(geom, depth_prob[:,None] * img_feat[:,:,None] permuted)."""
        return lift_splat(depth_prob, img_feat, self.prepare(geom))

    def lift_splat_cameras(self, rots, trans, intrins, post_rots, post_trans, depth_prob, img_feat):
        """get_geometry + lift + splat in the library: the ego-frame geometry is evaluated inside the voxel
        index kernel (same arithmetic as get_geometry, bit-identical indices), never materialised."""
        prep = lift_splat_prepare_cam(self.frustum, rots, trans, intrins, post_rots, post_trans,
                                      self._dx_host, self._bx_host, self._nx_host)
        return lift_splat(depth_prob, img_feat, prep)

    def forward(self, input):
        x, rots, trans, intrins, post_rots, post_trans = input[:6]
        B, N, C, H, W = x.shape
        x = self.depthnet(x.view(B * N, C, H, W))
        depth = self.get_depth_dist(x[:, :self.D])
        img_feat = x[:, self.D:(self.D + self.numC_Trans)]
        return self.lift_splat_cameras(rots, trans, intrins, post_rots, post_trans, depth, img_feat)


@MODELS.register_module()
class ViewTransformerLSSBEVDepth(ViewTransformerLiftSplatShoot):
    def __init__(self, extra_depth_net, loss_depth_weight, se_config=dict(), dcn_config=dict(bias=True), **kwargs):
        super().__init__(**kwargs)
        self.loss_depth_weight = loss_depth_weight
        self.extra_depthnet = build_backbone(extra_depth_net)
        c = extra_depth_net["num_channels"][0]
        self.featnet = nn.Conv2d(self.numC_input, self.numC_Trans, kernel_size=1, padding=0)
        self.depthnet = nn.Conv2d(c, self.D, kernel_size=1, padding=0)
        self.dcn = nn.Sequential(build_conv_layer(dict(type="DCNv2", deform_groups=1), c, c, kernel_size=3,
                                                  stride=1, padding=1, dilation=1, **dcn_config),
                                 nn.BatchNorm2d(c))
        self.se = SELikeModule(self.numC_input, feat_channel=c, **se_config)

    def depth_feat_and_prob(self, x, rots, trans, intrins, post_rots, post_trans):
        """vt_mine.py:311-327 / bevdet_distill_more.py:396-411 -> (img_feat, depth_digit, depth_prob); the tail of the depth branch
        -- self.dcn's BatchNorm2d, self.depthnet, get_depth_dist -- is one kernel after the norm's statistics (depth_head.py)."""
        BN = x.shape[0]
        img_feat = self.featnet(x)
        cam_params = torch.cat([intrins.reshape(BN, -1), post_rots.reshape(BN, -1), post_trans.reshape(BN, -1),
                                rots.reshape(BN, -1), trans.reshape(BN, -1)], dim=1)
        depth_feat = self.se(x, cam_params)
        depth_feat = self.extra_depthnet(depth_feat)[0]
        if type(self).get_depth_dist is ViewTransformerLiftSplatShoot.get_depth_dist and len(self.dcn) == 2:
            depth_digit, depth_prob = depth_head(self.dcn[0](depth_feat), self.dcn[1], self.depthnet)
        else:                                       # a subclass with its own depth distribution / another dcn stack
            depth_digit = self.depthnet(self.dcn(depth_feat))
            depth_prob = self.get_depth_dist(depth_digit)
        return img_feat, depth_digit, depth_prob

    def depth_and_feat(self, x, rots, trans, intrins, post_rots, post_trans):
        """vt_mine.py:311-323 / bevdet_distill_more.py:396-410 -> (img_feat, depth_digit)."""
        return self.depth_feat_and_prob(x, rots, trans, intrins, post_rots, post_trans)[:2]

    def forward(self, input):
        x, rots, trans, intrins, post_rots, post_trans = input[:6]
        B, N, C, H, W = x.shape
        img_feat, depth_digit, depth_prob = self.depth_feat_and_prob(x.view(B * N, C, H, W), rots, trans, intrins, post_rots,
                                                                     post_trans)
        return self.lift_splat_cameras(rots, trans, intrins, post_rots, post_trans, depth_prob, img_feat), depth_digit


# bev_pool-based twins of view_transformer.py resolve to the same implementation
MODELS.register_module(name="OfficialViewTransformerLiftSplatShoot", module=ViewTransformerLiftSplatShoot)
MODELS.register_module(name="OfficialViewTransformerLSSBEVDepth", module=ViewTransformerLSSBEVDepth)
