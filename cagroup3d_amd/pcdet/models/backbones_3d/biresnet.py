"""BiResNet: dual-resolution sparse ResNet + DAPPM (mirror of
pcdet/models/backbones_3d/biresnet.py:8-406) on the gfx950 sparse engine (`cagroup3d_amd.me`).

Module / parameter names follow the reference so that its checkpoints' keys line up
(conv1, layer1..5, layer3_..5_, compression3/4, down3/4, spp, out; `.kernel`, `.bn.weight` ...).
Layer table: SURVEY.md Appendix A."""
import torch.nn as nn

from .... import me as ME

BN_MOM = 0.1


def _conv_bn(cin, cout, k, stride=1, relu=False):
    layers = [ME.MinkowskiConvolution(cin, cout, kernel_size=k, stride=stride, bias=False, dimension=3),
              ME.MinkowskiBatchNorm(cout, momentum=BN_MOM)]
    if relu:
        layers.append(ME.MinkowskiReLU(inplace=True))
    return layers


class BasicBlock(nn.Module):
    """k3 conv-BN-ReLU, k3 conv-BN, (+ downsampled) residual, optional ReLU (biresnet.py:8-50)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, no_relu=False, bn_momentum=0.1,
                 dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                             dimension=dimension)
        self.norm1 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                             dimension=dimension)
        self.norm2 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.downsample, self.no_relu = downsample, no_relu

    def forward(self, x):
        out = self.conv2(self.norm1(self.conv1(x), act=ME.ACT_RELU))
        res = x if self.downsample is None else self.downsample(x)
        return self.norm2(out, act=ME.ACT_NONE if self.no_relu else ME.ACT_RELU, residual=res)   # BN + add (+ReLU) fused


class Bottleneck(nn.Module):
    """k1 - k3(stride) - k1 with expansion 2 (biresnet.py:52-103)."""
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, no_relu=True, bn_momentum=0.1,
                 dimension=-1):
        super().__init__()
        assert dimension > 0
        self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=1, stride=1, bias=False, dimension=dimension)
        self.norm1 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, bias=False,
                                             dilation=dilation, dimension=dimension)
        self.norm2 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = ME.MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, stride=1, bias=False,
                                             dimension=dimension)
        self.norm3 = ME.MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.downsample, self.stride, self.no_relu = downsample, stride, no_relu

    def forward(self, x):
        out = self.norm1(self.conv1(x), act=ME.ACT_RELU)
        out = self.norm2(self.conv2(out), act=ME.ACT_RELU)
        res = x if self.downsample is None else self.downsample(x)
        return self.norm3(self.conv3(out), act=ME.ACT_NONE if self.no_relu else ME.ACT_RELU, residual=res)


def _pre_act(cin, cout, k, pool=None):
    """(pool) - BN - ReLU - conv: the pre-activation branches of DAPPM (biresnet.py:109-174)."""
    layers = [] if pool is None else [ME.MinkowskiAvgPooling(kernel_size=pool[0], stride=pool[1], dimension=3)]
    layers += [ME.MinkowskiBatchNorm(cin, momentum=BN_MOM), ME.MinkowskiReLU(inplace=True),
               ME.MinkowskiConvolution(cin, cout, kernel_size=k, bias=False, dimension=3)]
    return ME.Sequential(*layers)


class DAPPM(nn.Module):
    """Deep aggregation pyramid pooling on the coarsest map (biresnet.py:105-203)."""

    def __init__(self, inplanes, branch_planes, outplanes, dimension=-1):
        assert dimension > 0
        super().__init__()
        self.scale1 = _pre_act(inplanes, branch_planes, 1, pool=(5, 2))
        self.scale2 = _pre_act(inplanes, branch_planes, 1, pool=(9, 4))
        self.scale3 = _pre_act(inplanes, branch_planes, 1, pool=(17, 8))
        self.scale4 = _pre_act(inplanes, branch_planes, 1, pool=(33, 16))
        self.scale0 = _pre_act(inplanes, branch_planes, 1)
        self.process1 = _pre_act(branch_planes, branch_planes, 3)
        self.process2 = _pre_act(branch_planes, branch_planes, 3)
        self.process3 = _pre_act(branch_planes, branch_planes, 3)
        self.process4 = _pre_act(branch_planes, branch_planes, 3)
        self.compression = _pre_act(branch_planes * 5, outplanes, 1)
        self.shortcut = _pre_act(inplanes, outplanes, 1)

    def forward(self, x):
        xc = x.C.float()
        feats = [self.scale0(x)]
        for scale, process in ((self.scale1, self.process1), (self.scale2, self.process2),
                               (self.scale3, self.process3), (self.scale4, self.process4)):
            up = x._like(scale(x).features_at_coordinates(xc))      # pooled branch, interpolated back
            feats.append(process(up + feats[-1]))
        return self.compression(ME.cat(*feats)) + self.shortcut(x)


class BiResNet(nn.Module):
    def __init__(self, model_cfg, block=BasicBlock, **kwargs):
        super().__init__()
        cin = model_cfg.get("IN_CHANNELS", 3)
        cout = model_cfg.get("OUT_CHANNELS", 64)
        layers = model_cfg.get("LAYERS", [2, 2, 2, 2])
        planes = model_cfg.get("PLANES", 64)
        spp_planes = model_cfg.get("SPP_PLANES", 128)
        dim = model_cfg.get("DIMENSION", 3)
        hi = planes * 2
        assert not model_cfg.get("AUGMENT", False), "seghead_extra (AUGMENT) is never built by CAGroup3D.yaml"

        def plain(ci, co, k, stride=1):   # conv default bias=False in ME
            return ME.MinkowskiConvolution(ci, co, kernel_size=k, stride=stride, dimension=dim)

        self.conv1 = ME.Sequential(plain(cin, planes, 3), ME.MinkowskiBatchNorm(planes, momentum=BN_MOM),
                                   ME.MinkowskiReLU(inplace=True),
                                   plain(planes, planes, 3), ME.MinkowskiBatchNorm(planes, momentum=BN_MOM),
                                   ME.MinkowskiReLU(inplace=True))
        self.relu = ME.MinkowskiReLU(inplace=False)
        self.layer1 = self._make_layer(block, planes, planes, layers[0], stride=2, dimension=dim)
        self.layer2 = self._make_layer(block, planes, planes * 2, layers[1], stride=2, dimension=dim)
        self.layer3 = self._make_layer(block, planes * 2, planes * 4, layers[2], stride=2, dimension=dim)
        self.layer4 = self._make_layer(block, planes * 4, planes * 8, layers[3], stride=2, dimension=dim)
        self.compression3 = ME.Sequential(*_conv_bn(planes * 4, hi, 1))
        self.compression4 = ME.Sequential(*_conv_bn(planes * 8, hi, 1))
        self.down3 = ME.Sequential(*_conv_bn(hi, planes * 4, 3, stride=2))
        self.down4 = ME.Sequential(*(_conv_bn(hi, planes * 4, 3, stride=2, relu=True)
                                     + _conv_bn(planes * 4, planes * 8, 3, stride=2)))
        self.layer3_ = self._make_layer(block, planes * 2, hi, 2, dimension=dim)
        self.layer4_ = self._make_layer(block, hi, hi, 2, dimension=dim)
        self.layer5_ = self._make_layer(Bottleneck, hi, hi, 1, dimension=dim)
        self.layer5 = self._make_layer(Bottleneck, planes * 8, planes * 8, 1, stride=2, dimension=dim)
        self.spp = DAPPM(planes * 16, spp_planes, planes * 4, dimension=dim)
        self.out = ME.Sequential(
            ME.MinkowskiConvolutionTranspose(planes * 4, planes * 4, kernel_size=2, stride=2, dimension=dim),
            ME.MinkowskiBatchNorm(planes * 4, momentum=BN_MOM), ME.MinkowskiReLU(inplace=True),
            ME.MinkowskiConvolution(planes * 4, cout, kernel_size=1, bias=False, dimension=dim),
            ME.MinkowskiBatchNorm(cout, momentum=BN_MOM), ME.MinkowskiReLU(inplace=True))
        self.num_point_features = cout
        self.init_weights()

    def init_weights(self):
        """Kaiming-normal (fan_out) on MinkowskiConvolution kernels only -- the transposed conv keeps its
        default init, as in the reference (biresnet.py:326-333)."""
        for m in self.modules():
            if type(m) is ME.MinkowskiConvolution:
                ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, block, inplanes, planes, blocks, stride=1, dimension=-1):
        down = None
        if stride != 1 or inplanes != planes * block.expansion:
            down = ME.Sequential(ME.MinkowskiConvolution(inplanes, planes * block.expansion, kernel_size=1,
                                                         stride=stride, dimension=dimension),
                                 ME.MinkowskiBatchNorm(planes * block.expansion, momentum=BN_MOM))
        mods = [block(inplanes, planes, stride=stride, downsample=down, dimension=dimension)]
        inplanes = planes * block.expansion
        for i in range(1, blocks):
            mods.append(block(inplanes, planes, stride=1, no_relu=(i == blocks - 1), dimension=dimension))
        return ME.Sequential(*mods)

    def forward(self, input_dict):
        from .... import engine
        if engine.applicable(self):
            # the whole pass as one launch program (engine.py): one foreign call forward, one backward
            gs = getattr(self, "grad_sync", None)
            hooks = None
            if gs is not None and gs.mid:
                gs.begin_mid()
                hooks = {"mid": (lambda: gs._on_stem_output_grad(None), {id(p) for p in gs.mid})}
            try:
                return {"sp_tensor": engine.run_backbone(self, input_dict["sp_tensor"], input_dict.get("engine_program"), hooks)}
            except engine.NotReady:
                pass                                                  # (first steps: weights not in the step's arena yet)
        x = self.conv1(input_dict["sp_tensor"])                       # ts 1
        l1 = self.layer1(x)                                           # ts 2
        l2 = self.layer2(self.relu(l1))                               # ts 4
        if self.training and getattr(self, "grad_sync", None) is not None and not ME.coords_only():
            self.grad_sync.attach_mid(l2.F)          # every deeper layer's gradient is complete when this one is
        r2 = self.relu(l2)                                            # (one ReLU for both consumers)
        l3 = self.layer3(r2)                                          # ts 8
        hi = self.layer3_(r2)                                         # ts 4 (high-resolution branch)

        # the joins  lo = l3 + down3(relu(hi)),  hi = hi + compression3(relu(l3))(hi.C)  are only ever read through a ReLU:
        # ME.add_relu forms relu(a + b) in one pass (biresnet.py:378-394 in the reference: add, then relu at the consumer)
        r3, rh = self.relu(l3), self.relu(hi)
        lo_r = ME.add_relu(l3, self.down3(rh))
        hi_r = ME.add_relu(hi, hi._like(self.compression3(r3).features_at_coordinates(hi.C.float())))

        l4 = self.layer4(lo_r)                                        # ts 16
        hi = self.layer4_(hi_r)
        r4, rh = self.relu(l4), self.relu(hi)
        lo_r = ME.add_relu(l4, self.down4(rh))
        hi_r = ME.add_relu(hi, hi._like(self.compression4(r4).features_at_coordinates(hi.C.float())))

        hi = self.layer5_(hi_r)
        hi = hi._like(hi.F + self.spp(self.layer5(lo_r)).features_at_coordinates(hi.C.float()))
        return {"sp_tensor": self.out(hi)}                            # ts 2, OUT_CHANNELS
