"""Image / label encoders on the gfx950 kernels.

Drop-in for ``deephumor.models.encoders`` (reference encoders.py:7-144): same class names,
constructor arguments, ``forward`` signatures and state-dict keys.  The modules below are
parameter containers -- the arithmetic runs in ``libdeephumor_hip.so`` (``deephumor_amd.hip``),
batched over images.  Inference only: like the reference's ``generate`` callers
(deephumor_demo.ipynb:1134-1141) the model must be in ``eval()`` mode.
"""
import os

import torch
from torch import nn

from .. import hip
from ._f32x_guard import f32x_guarded
from .. import f32xp

RESNET50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


def _require_eval(module, p):
    if module.training and p > 0:
        raise RuntimeError("deephumor_amd implements the inference path (eval mode); "
                           "call model.eval() -- train-mode dropout is out of scope")


class _Bottleneck(nn.Module):
    """Parameter holder with torchvision's Bottleneck attribute names (conv1..bn3, downsample)."""

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None
        self.stride = stride


def _resnet50_trunk():
    """Children 0..7 of the reference's ``self.resnet`` (encoders.py:37-38): conv1, bn1, relu,
    maxpool, layer1..layer4 -- the 318 trunk tensors of the state dict."""
    mods = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1)]
    inplanes = 64
    for planes, blocks, stride in RESNET50_STAGES:
        stage = [_Bottleneck(inplanes, planes, stride, downsample=True)]
        inplanes = planes * 4
        stage += [_Bottleneck(inplanes, planes, 1, downsample=False) for _ in range(1, blocks)]
        mods.append(nn.Sequential(*stage))
    return nn.Sequential(*mods)


def _bn_affine(bn):
    """Eval-mode BatchNorm as y = x*scale + shift (applied in the kernel epilogue, weights untouched)."""
    scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)).contiguous()
    shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
    return scale, shift


class _Planned:
    """Lazily derived device-side constants, rebuilt when any parameter/buffer changes (storage pointer or in-place
    version counter).  The tensor list is cached -- walking a ResNet-50's modules costs more host time per call than
    the key itself -- and dropped whenever the module is converted or (re)loaded (``_apply``: ``.to`` / ``.cuda`` /
    ``.bfloat16``; ``load_state_dict``), the two ways standard PyTorch code replaces parameter or buffer objects."""

    def _plan_tensors_list(self, refresh=False):
        ts = self.__dict__.get("_plan_tensors")
        if ts is None or refresh:
            ts = list(self.parameters()) + list(self.buffers())
            object.__setattr__(self, "_plan_tensors", ts)
        return ts

    def _plan_key(self, refresh=False):
        ts = self._plan_tensors_list(refresh)
        return tuple(t.data_ptr() for t in ts), tuple(t._version for t in ts), hip.options_epoch

    def _drop_plan(self):
        for m in self.modules():
            if isinstance(m, _Planned):
                m.__dict__.pop("_plan_tensors", None)
                m.__dict__.pop("_plan_cache", None)

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._drop_plan()
        return out

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self._drop_plan()
        return out

    def _get_plan(self):
        key = self._plan_key()
        if self.__dict__.get("_plan_cache") is None or self._plan_cache[0] != key:
            key = self._plan_key(refresh=True)            # something changed: re-walk the modules, then rebuild
            object.__setattr__(self, "_plan_cache", (key, self._build_plan()))
        return self._plan_cache[1]


class ImageEncoder(_Planned, nn.Module):
    """ResNet-50 image encoder (reference encoders.py:7-70).

    ``forward(images[N,3,H,W])`` returns ``emb [N, emb_dim]`` or, with ``spatial_features``,
    ``(emb, spatial_emb [N, k*k, emb_dim])``; the Linear is shared, BatchNorm1d only on the global
    branch (encoders.py:61 vs :67).  The reference's ``pretrained=True`` download (encoders.py:34)
    is not reproduced: weights come from ``load_state_dict`` / ``from_pretrained``.
    """

    def __init__(self, emb_dim=256, dropout=0.2, spatial_features=False):
        super().__init__()
        self.spatial_features = spatial_features
        self.resnet = _resnet50_trunk()
        for p in self.resnet.parameters():
            p.requires_grad = False                       # encoders.py:35-36
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.linear = nn.Linear(2048, emb_dim)
        self.bn = nn.BatchNorm1d(emb_dim)
        self.dropout = nn.Dropout(dropout)

    def _build_plan(self):
        """fp32 weights -> NCHW vector-ALU convolutions (the parity path); bf16 weights -> channels-last
        activations with every bottleneck conv on the bf16 matrix cores (weights repacked once to
        [Cout, kh, kw, Cin]; the stem's 3 input channels are zero-padded to 8)."""
        wdt = self.linear.weight.dtype
        bf16 = wdt in hip.HALF_DTYPES            # the 16-bit paths (bf16 / fp16 storage and MFMA operands)
        # fp32 weights with option "f32_split": channels-last fp32 activations, every convolution / linear layer as three fp16 MFMAs
        # on split operands (csrc/gemm_f32x.hip) -- fp32-class results on the matrix cores instead of the vector ALUs
        split = (wdt == torch.float32 and self.linear.weight.is_cuda and bool(hip.option("f32_split"))
                 and all(hip.f32_split_ok(p) for p in self.parameters() if p.dim() > 1))

        def conv(c, bn, relu, residual=False, stem=False):
            s, b = _bn_affine(bn)
            w = c.weight.detach()
            if split:
                wl = w.permute(0, 2, 3, 1)                                   # [Cout, kh, kw, Cin]: ci fastest
                if wl.shape[3] % 4:                                          # the stem: 3 input channels zero-padded to 4
                    w4 = torch.zeros(tuple(wl.shape[:3]) + ((wl.shape[3] + 3) // 4 * 4,), dtype=w.dtype, device=w.device)
                    w4[..., :wl.shape[3]] = wl
                    wl = w4
                ent = dict(w=w.contiguous(), wx=hip.split_f32x(wl.reshape(wl.shape[0], -1).contiguous()), ks=w.shape[2], cin=wl.shape[3],
                           scale=s, shift=b, stride=c.stride[0], pad=c.padding[0], relu=relu, residual=residual)
                if ent["ks"] == 1 and ent["stride"] == 1 and hip.option("f32_planes"):
                    # the wide 1 x 1 layers (conv3, layer1's downsample) as a persistent streaming kernel (csrc/conv1x1_f32x.hip)
                    ent["wxs"] = f32xp.pack_conv1x1(ent["wx"])
                return ent
            if bf16 and stem:
                # 3 input channels zero-padded to 8: the stem becomes a Cin=8 channels-last conv on the matrix cores
                w8 = torch.zeros((w.shape[0], w.shape[2], w.shape[3], 8), dtype=w.dtype, device=w.device)
                w8[..., :w.shape[1]] = w.permute(0, 2, 3, 1)
                w = w8.contiguous()
            elif bf16:
                w = w.permute(0, 2, 3, 1).contiguous()
            else:
                w = w.contiguous()
            return dict(w=w, scale=s, shift=b, stride=c.stride[0], pad=c.padding[0], relu=relu, residual=residual)

        blocks = []
        stem = conv(self.resnet[0], self.resnet[1], True, stem=True)
        c1 = self.resnet[0]
        if bf16 and tuple(c1.weight.shape) == (64, 3, 7, 7) and c1.stride[0] == 2 and c1.padding[0] == 3:
            stem["wpk"] = hip.pack_stem_weight(c1.weight, wdt)      # the direct-convolution stem kernel's weight layout
        for stage in list(self.resnet)[4:]:
            for blk in stage:
                ent = dict(
                    down=conv(blk.downsample[0], blk.downsample[1], False) if blk.downsample is not None else None,
                    c1=conv(blk.conv1, blk.bn1, True), c2=conv(blk.conv2, blk.bn2, True),
                    c3=conv(blk.conv3, blk.bn3, True, residual=True), dual=None)
                c1w = ent["c1"]["w"]
                if (bf16 and c1w.is_cuda and tuple(c1w.shape[1:3]) == (1, 1) and ent["c1"]["stride"] == 1 and c1w.shape[3] in (256, 512, 1024)
                        and c1w.shape[0] % 128 == 0):
                    ent["c1"]["wpk1"] = hip.pack_mfma_fragments(c1w.reshape(c1w.shape[0], c1w.shape[3]).contiguous())
                if bf16 and ent["down"] is not None:
                    # first block of a stage on the bf16 path: relu(bn3(conv3(y)) + bn_d(downsample(x))) as ONE GEMM over
                    # [y | x at the strided pixels] with the BatchNorm scales folded into the weights -- the identity
                    # tensor (411 MB in stage 1 at 256 images) is neither written nor read back
                    c3, dn = ent["c3"], ent["down"]
                    cout = c3["w"].shape[0]
                    w_cat = torch.cat([c3["w"].float().view(cout, -1) * c3["scale"][:, None],
                                       dn["w"].float().view(cout, -1) * dn["scale"][:, None]], 1)
                    if c3["w"].shape[-1] % 64 == 0 and dn["w"].shape[-1] % 64 == 0:
                        ent["dual"] = dict(w=w_cat.to(wdt).contiguous(), shift=(c3["shift"] + dn["shift"]).contiguous(),
                                           stride=dn["stride"])
                        if w_cat.is_cuda and w_cat.shape[1] in (128, 384, 768) and cout % 256 == 0:
                            ent["dual"]["wpk"] = hip.pack_mfma_fragments(ent["dual"]["w"])
                if (bf16 and ent["down"] is None and ent["c2"]["stride"] == 1 and ent["c2"]["w"].shape[1] == 3
                        and ent["c3"]["w"].shape[0] == 4 * ent["c2"]["w"].shape[0] and ent["c2"]["w"].shape[0] == ent["c2"]["w"].shape[3]
                        and hip.bottleneck_tail_s3_supported(14, 14, ent["c2"]["w"].shape[0])):
                    # stage-3 blocks: the fused tail streams the weights from L2 into registers in MFMA fragment order
                    ent["w2p"] = hip.pack_mfma_fragments(ent["c2"]["w"])
                    ent["w3p"] = hip.pack_mfma_fragments(ent["c3"]["w"].reshape(ent["c3"]["w"].shape[0], -1).contiguous())
                elif (bf16 and ent["c2"]["w"].is_cuda and ent["down"] is None and ent["c2"]["stride"] == 1 and ent["c2"]["w"].shape[1] == 3
                        and ent["c3"]["w"].shape[0] == 4 * ent["c2"]["w"].shape[0] and ent["c2"]["w"].shape[0] == ent["c2"]["w"].shape[3]
                        and hip.bottleneck_tail_s1_supported(56, 56, ent["c2"]["w"].shape[0])):
                    # stage-1 blocks: 4-row strips (conv_s1.hip); this block's conv1 in fragment order too, for the previous tail to run it
                    ent["w2p1"] = hip.pack_mfma_fragments(ent["c2"]["w"])
                    ent["w3p1"] = hip.pack_mfma_fragments(ent["c3"]["w"].reshape(ent["c3"]["w"].shape[0], -1).contiguous())
                    if (tuple(c1w.shape[1:3]) == (1, 1) and ent["c1"]["stride"] == 1 and ent["c1"]["relu"]
                            and hip.bottleneck_tail_s1_supported(56, 56, ent["c2"]["w"].shape[0], c1w.shape[0]) and c1w.shape[3] == 4 * ent["c2"]["w"].shape[0]):
                        ent["c1"]["wpkf"] = hip.pack_mfma_fragments(c1w.reshape(c1w.shape[0], c1w.shape[3]).contiguous())
                elif (bf16 and ent["c2"]["w"].is_cuda and ent["down"] is None and ent["c2"]["stride"] == 1 and ent["c2"]["w"].shape[1] == 3
                        and ent["c3"]["w"].shape[0] == 4 * ent["c2"]["w"].shape[0] and ent["c2"]["w"].shape[0] == ent["c2"]["w"].shape[3]
                        and hip.bottleneck_tail_s2_supported(28, 28, ent["c2"]["w"].shape[0])):
                    # stage-2 blocks: the same structure on 4-row strips, three workgroups per CU (conv_s2.hip)
                    ent["w2p2"] = hip.pack_mfma_fragments(ent["c2"]["w"])
                    ent["w3p2"] = hip.pack_mfma_fragments(ent["c3"]["w"].reshape(ent["c3"]["w"].shape[0], -1).contiguous())
                    if (tuple(c1w.shape[1:3]) == (1, 1) and ent["c1"]["stride"] == 1 and ent["c1"]["relu"] and tuple(c1w.shape[::3]) == (128, 512)):
                        ent["c1"]["wpkf"] = hip.pack_mfma_fragments(c1w.reshape(128, 512).contiguous())     # for the previous tail to run it
                c3w = ent["c3"]["w"]
                if (bf16 and c3w.is_cuda and ent["down"] is None and tuple(c3w.shape[1:]) == (1, 1, 512) and c3w.shape[0] % 128 == 0
                        and c3w.shape[0] // 128 in (1, 2, 4, 8, 16)):
                    ent["c3"]["wpk1"] = hip.pack_mfma_fragments(c3w.reshape(c3w.shape[0], 512).contiguous())      # stage-4 conv3 + residual
                c2w = ent["c2"]["w"]
                if (bf16 and c2w.is_cuda and ent["c2"]["stride"] == 1 and ent["c2"]["pad"] == 1 and tuple(c2w.shape[1:3]) == (3, 3)
                        and hip.conv3x3_s4_supported(7, 7, c2w.shape[0]) and c2w.shape[3] == c2w.shape[0]):
                    # stage-4 conv2 (stride 1): two images x half of the channels per workgroup, weights register-streamed (conv_s4.hip)
                    ent["c2"]["wpk4"] = hip.pack_mfma_fragments(c2w)
                blocks.append(ent)
        s, b = _bn_affine(self.bn)
        return dict(stem=stem, blocks=blocks, bn_scale=s, bn_shift=b, bf16=bf16, dtype=wdt, split=split,
                    lin_w=self.linear.weight.detach(), lin_b=self.linear.bias.detach().float().contiguous(),
                    lin_wx=hip.split_f32x(self.linear.weight.detach().contiguous()) if split else None)

    @staticmethod
    def _conv(x, c, residual=None, nhwc=False):
        if (nhwc and "wpk1" in c and (residual is None or x.shape[3] == 512)
                and hip.conv1x1_wreg_supported(x.shape[0] * x.shape[1] * x.shape[2], x.shape[3], c["w"].shape[0])
                and hip.option("encoder_generic") < 1):
            # conv1 of the K >= 256 bottlenecks, conv3 + residual of stage 4: weights stationary in registers, pixels streamed
            # (csrc/conv1x1_wreg.hip; bit-identical)
            return hip.conv1x1_wreg_nhwc(x, c["wpk1"], c["w"].shape[0], c["scale"], c["shift"], relu=c["relu"], residual=residual)
        if (nhwc and residual is None and "wpk4" in c and c["relu"] and hip.conv3x3_s4_supported(x.shape[1], x.shape[2], x.shape[3])
                and hip.option("encoder_generic") < 1):
            return hip.conv3x3_s4_nhwc(x, c["wpk4"], c["scale"], c["shift"])
        if (nhwc and residual is None and c["relu"] and c["stride"] == 1 and c["pad"] == 1 and c["w"].shape[1] == 3
                and hip.conv3x3_direct_supported(x.shape[1], x.shape[2], x.shape[3], c["w"].shape[0])
                and hip.option("encoder_generic") < 2):
            return hip.conv3x3_direct_nhwc(x, c["w"], c["scale"], c["shift"])      # patch-resident direct convolution (stages 1-2)
        fn = hip.conv2d_nhwc_bn_act if nhwc else hip.conv2d_bn_act
        return fn(x, c["w"], c["scale"], c["shift"], residual=residual, relu=c["relu"], stride=c["stride"], pad=c["pad"])

    TRUNK_MAX_IMAGES = 384

    def _features_split(self, images, plan):
        """The trunk on channels-last fp32 tensors with split-operand MFMAs (plan["split"]) -> [N, H/32, W/32, 2048] fp32."""
        def cv(x, c, residual=None):
            if c.get("wxs") is not None and f32xp.conv1x1_stream_supported(x.shape[0] * x.shape[1] * x.shape[2], x.shape[3], c["wx"].shape[1]):
                return f32xp.conv1x1_stream(x, c["wxs"], c["scale"], c["shift"], residual=residual, relu=c["relu"])
            return hip.conv2d_nhwc_f32x(x, c["wx"], c["ks"], c["scale"], c["shift"], residual=residual, relu=c["relu"], stride=c["stride"],
                                        pad=c["pad"])
        def c12(x, c1, c2):
            # conv1 -> conv2 of stages 2 - 4 (3 x 3 at >= 128 channels): conv1 stores its output as fp16 planes, conv2 reads them through the
            # planes kernel (csrc/gemm_f32xp.hip: both operands by LDS-DMA, two wave groups a phase apart; 13 - 27 % faster, bit-identical)
            if hip.option("f32_planes") and c2["ks"] == 3 and c2["cin"] % 32 == 0 and c2["wx"].shape[1] >= 128 and not c1["residual"]:
                yp = f32xp.conv2d_nhwc_planes_out(x, c1["wx"], c1["ks"], c1["scale"], c1["shift"], relu=c1["relu"], stride=c1["stride"], pad=c1["pad"])
                return f32xp.conv2d_nhwc(yp, c2["wx"], 3, c2["scale"], c2["shift"], relu=c2["relu"], stride=c2["stride"], pad=c2["pad"])
            return cv(cv(x, c1), c2)
        st = plan["stem"]
        x = hip.nchw_to_nhwc_f32(images.float().contiguous(), cp=st["cin"])
        x = hip.maxpool3x3s2_nhwc_f32(cv(x, st))
        for blk in plan["blocks"]:
            y = c12(x, blk["c1"], blk["c2"])
            idt = x if blk["down"] is None else cv(x, blk["down"])
            x = cv(y, blk["c3"], residual=idt)
        return x

    def features(self, images):
        """Trunk output (encoders.py:56): ``[N, 2048, H/32, W/32]`` fp32 on the parity path,
        channels-last ``[N, H/32, W/32, 2048]`` bf16 on the bf16 path."""
        plan = self._get_plan()
        nhwc = plan["bf16"]
        st = plan["stem"]
        # what torch's conv2d would refuse (the kernels take the layout on trust: a 1- or 4-channel tensor read as 3 channels is foreign
        # memory or garbage, not an error -- round 5 probe)
        if images.dim() != 4:
            raise RuntimeError(f"Expected 4D (batched) input to conv2d, but got input of size: {list(images.shape)}")
        packed_in = nhwc and images.shape[-1] == 8 and images.dtype == plan["dtype"]      # experiments.inference.preprocess_images' layout
        if not packed_in:
            if images.shape[1] != 3:
                raise RuntimeError(f"Given groups=1, weight of size [64, 3, 7, 7], expected input{list(images.shape)} to have 3 channels, "
                                   f"but got {images.shape[1]} channels instead")
            if images.dtype not in (torch.float32, torch.bfloat16, torch.float16):
                raise TypeError(f"unsupported image dtype {images.dtype}: float32 (or the model's 16-bit type) NCHW images, or the packed "
                                "tensor of experiments.inference.preprocess_images")
        if plan["split"]:
            return self._features_split(images, plan)
        if nhwc and images.shape[0] > self.TRUNK_MAX_IMAGES:
            #  the streaming / register-streamed 16-bit kernels index pixels with 32-bit magic divisions and are tuned for
            # <= 256 images per launch (the stage-1 dual kernel stops applying near 436 images): larger batches go through the trunk
            # in chunks of 256 -- every image's arithmetic is independent of its batch, so the features are the same bit for bit
            return torch.cat([self.features(images[i:i + 256]) for i in range(0, images.shape[0], 256)], 0)
        if nhwc:
            prepacked = images.dim() == 4 and images.shape[-1] == 8 and images.dtype == plan["dtype"]
            h_in, w_in = (images.shape[1], images.shape[2]) if prepacked else (images.shape[2], images.shape[3])
            direct = ("wpk" in st and h_in >= 2 and w_in >= 2 and ((h_in - 1) // 2 + 1) % 2 == 0 and ((w_in - 1) // 2 + 1) % 2 == 0
                      and (prepacked or images.shape[1] == 3) and hip.option("encoder_generic") < 3)
            if direct:
                # conv1 + bn1 + relu + maxpool as ONE direct-convolution launch that reads the caller's tensor as it is (fp32 NCHW
                # or the preprocessing kernels' packed 16-bit layout): no packing launch, no im2col traffic, no un-pooled activation
                src = images.contiguous() if prepacked else images.float().contiguous()
                x = hip.stem_conv7_bn_relu_maxpool(src, st["wpk"], st["scale"], st["shift"])
                return self._trunk(x, plan)
            if prepacked:
                packed = images.contiguous()              # already normalised + packed (experiments.inference.preprocess_images)
            else:
                packed = hip.pack_nchw_to_nhwc8(images.float().contiguous(), out_dtype=plan["dtype"])
            ks, cout = st["w"].shape[1], st["w"].shape[0]
            ho = (packed.shape[1] + 2 * st["pad"] - ks) // st["stride"] + 1
            wo = (packed.shape[2] + 2 * st["pad"] - ks) // st["stride"] + 1
            if ho % 2 == 0 and wo % 2 == 0 and cout <= 64 and hip.option("encoder_generic") < 3:
                # conv1 + bn1 + relu + maxpool in one launch: the un-pooled 112 x 112 x 64 activation never exists
                x = hip.conv2d_nhwc_bn_relu_maxpool(packed, st["w"], st["scale"], st["shift"], st["stride"], st["pad"])
            else:
                x = hip.maxpool3x3s2_nhwc(self._conv(packed, st, nhwc=True))
        else:
            x = hip.maxpool3x3s2(self._conv(images.contiguous(), st))
        return self._trunk(x, plan)

    def _trunk(self, x, plan):
        """layer1..layer4 on the pooled stem output: the kernel of every bottleneck is chosen ONCE per (plan, input shape) -- a table of
        steps kept with the plan (``_trunk_table``) -- and a forward is a walk over that table."""
        key = (tuple(x.shape), hip.options_epoch)
        table = plan.setdefault("_trunk_tables", {}).get(key)
        if table is None:
            table = plan["_trunk_tables"][key] = self._trunk_table(plan, tuple(x.shape))
        ready = None                                      # the NEXT block's conv1 output when the previous tail launch computed it
        for step in table:
            x, ready = step(x, ready)
        return x

    def _trunk_table(self, plan, shape):
        """[step(x, y1_ready) -> (x, y1_for_next_block)] for an input of ``shape`` ([N, H, W, C] on the 16-bit paths, NCHW in fp32).
        Option ``encoder_generic``: 0 = every specialised kernel; >= 1 without round 4's (streaming 1x1, stage-1 / 2 / 4 tails, conv1 fusions);
        >= 2 also without the patch-resident 3x3 / fused tails of rounds 2-3: every convolution through the implicit-GEMM tile kernel (all
        of them bit-identical to it: tests/test_bf16_gpu.py)."""
        nhwc, blocks, lvl = plan["bf16"], plan["blocks"], hip.option("encoder_generic")
        conv = self._conv
        table = []
        h, w = (shape[1], shape[2]) if nhwc else (shape[2], shape[3])
        for bi, blk in enumerate(blocks):
            c1, c2, c3 = blk["c1"], blk["c2"], blk["c3"]
            nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
            cmid = c2["w"].shape[0]
            first = lambda x, ready, c1=c1: ready if ready is not None else conv(x, c1, nhwc=True)
            if (nhwc and lvl < 1 and "w2p1" in blk and nxt is not None and "wpkf" in nxt["c1"] and "w2p1" in nxt
                    and hip.bottleneck_tail_s1_supported(h, w, cmid, nxt["c1"]["w"].shape[0])):
                # stage 1 (56 x 56 x 64), next block's conv1 256 -> 64: 4-row strips, weights register-streamed, and that conv1 in the
                # same launch on the output tile while it is in LDS -- the 411 MB tensor is not read back for it (conv_s1.hip).
                # (Without the fusion the ring kernel below is the faster tail in the encoder: 195 against 214 us.)
                n1 = nxt["c1"]
                table.append(lambda x, ready, blk=blk, c2=c2, c3=c3, n1=n1, first=first: hip.bottleneck_tail_s1_nhwc(
                    first(x, ready), blk["w2p1"], c2["scale"], c2["shift"], blk["w3p1"], c3["scale"], c3["shift"], x,
                    n1["wpkf"], n1["scale"], n1["shift"], n1["w"].shape[0]))
            elif nhwc and lvl < 2 and "w2p" in blk and hip.bottleneck_tail_s3_supported(h, w, cmid):
                # stage 3 (14 x 14 x 256): one image per workgroup, patch-resident 3x3 + 1x1 expansion, weights register-streamed
                table.append(lambda x, ready, blk=blk, c2=c2, c3=c3, first=first: (hip.bottleneck_tail_s3_nhwc(
                    first(x, ready), blk["w2p"], c2["scale"], c2["shift"], blk["w3p"], c3["scale"], c3["shift"], x), None))
            elif nhwc and lvl < 1 and "w2p2" in blk and hip.bottleneck_tail_s2_supported(h, w, cmid):
                if nxt is not None and "wpkf" in nxt["c1"] and "w2p2" in nxt:
                    # + the next block's conv1 (512 -> 128) on the output chunks in LDS: 170 us against 135 + 55 us
                    n1 = nxt["c1"]
                    table.append(lambda x, ready, blk=blk, c2=c2, c3=c3, n1=n1, first=first: hip.bottleneck_tail_s2_nhwc(
                        first(x, ready), blk["w2p2"], c2["scale"], c2["shift"], blk["w3p2"], c3["scale"], c3["shift"], x,
                        n1["wpkf"], n1["scale"], n1["shift"], 128))
                else:
                    table.append(lambda x, ready, blk=blk, c2=c2, c3=c3, first=first: (hip.bottleneck_tail_s2_nhwc(
                        first(x, ready), blk["w2p2"], c2["scale"], c2["shift"], blk["w3p2"], c3["scale"], c3["shift"], x), None))
            elif (nhwc and lvl < 2 and blk["dual"] is None and blk["down"] is None and c2["stride"] == 1 and c2["w"].shape[1] == 3
                    and hip.conv3x3_direct_supported(h, w, c2["w"].shape[3], cmid) and c3["w"].shape[0] == 4 * cmid):
                # conv2 + bn2 + relu + conv3 + bn3 + residual + relu in one launch: the conv2 output tile stays in LDS
                table.append(lambda x, ready, c2=c2, c3=c3, first=first: (hip.bottleneck_tail_nhwc(
                    first(x, ready), c2["w"], c2["scale"], c2["shift"], c3["w"], c3["scale"], c3["shift"], x), None))
            else:
                table.append(self._generic_block(blk, nxt, nhwc, lvl))
            if blk["down"] is not None:
                st = blk["down"]["stride"]
                h, w = (h - 1) // st + 1, (w - 1) // st + 1
        return table

    def _generic_block(self, blk, nxt, nhwc, lvl):
        """conv1 -> conv2 -> (conv3 + downsample as ONE GEMM | downsample, conv3 + residual): the first block of every stage, and every
        block when the specialised tails do not apply."""
        conv = self._conv

        def step(x, ready):
            y1 = ready if ready is not None else conv(x, blk["c1"], nhwc=nhwc)
            y = conv(y1, blk["c2"], nhwc=nhwc)
            d = blk["dual"]
            if d is not None:
                if "wpk" in d and lvl < 1 and hip.conv1x1_dual_wreg_supported(y.shape, x.shape, d["w"].shape[0]):
                    if nxt is not None and "wpkf" in nxt["c1"] and d["w"].shape == (256, 128) and nxt["c1"]["w"].shape[0] == 64:
                        # layer1.0's ending + layer1.1's conv1 in one launch (the block's 256 output channels are in the workgroup)
                        n1 = nxt["c1"]
                        return hip.conv1x1_dual_wreg_nhwc(y, x, d["wpk"], 256, d["shift"], d["stride"], relu=True, w1p=n1["wpkf"],
                                                          scale1=n1["scale"], shift1=n1["shift"], n1=64)
                    return hip.conv1x1_dual_wreg_nhwc(y, x, d["wpk"], d["w"].shape[0], d["shift"], d["stride"], relu=True), None
                return hip.conv1x1_dual_nhwc(y, x, d["w"], d["shift"], d["stride"], relu=True), None
            idt = x if blk["down"] is None else conv(x, blk["down"], nhwc=nhwc)
            return conv(y, blk["c3"], residual=idt, nhwc=nhwc), None
        return step

    @f32x_guarded
    def forward(self, images):
        _require_eval(self, self.dropout.p)
        plan = self._get_plan()
        feats = self.features(images)
        n = feats.shape[0]
        w, b = plan["lin_w"], plan["lin_b"]
        if plan["bf16"]:
            pooled, rows = hip.avgpool_nhwc(feats), feats.view(n, -1, feats.shape[-1])     # already [N, k*k, 2048]
        elif plan["split"]:
            pooled, rows = hip.avgpool_nhwc_f32(feats), feats.view(n, -1, feats.shape[-1])
            emb = hip.linear_f32x(pooled, plan["lin_wx"], b, scale=plan["bn_scale"], shift=plan["bn_shift"])
            if not self.spatial_features:
                return emb
            return emb, hip.linear_f32x(rows.reshape(-1, rows.shape[-1]), plan["lin_wx"], b).view(n, rows.shape[1], -1)
        else:
            pooled = hip.avgpool_rows(feats)
        emb = hip.linear(pooled, w, b, scale=plan["bn_scale"], shift=plan["bn_shift"])
        if not self.spatial_features:
            return emb
        if not plan["bf16"]:
            rows = hip.nchw_to_rows(feats)                                    # [N, k*k, 2048]
        rows2 = rows.reshape(-1, rows.shape[-1])
        if rows2.dtype == torch.float16:
            # fp16 underflows below 3e-8 where fp32 / bf16 do not, and TransformerDecoder reads a row with any exactly-zero feature as
            # padding (transformers.py:480): fp32 out of the GEMM, then a rounding that keeps non-zero values non-zero
            spatial = hip.round16_keep_nonzero(hip.linear(rows2, w, b, out_dtype=torch.float32), torch.float16)
        else:
            spatial = hip.linear(rows2, w, b)
        return emb, spatial.view(n, rows.shape[1], -1)


class LabelEncoder(nn.Module):
    """Mean label embedding (reference encoders.py:73-106)."""

    def __init__(self, num_tokens, emb_dim=256, dropout=0.2):
        super().__init__()
        self.embedding = nn.Embedding(num_tokens, emb_dim)
        self.dropout = nn.Dropout(dropout)

    def check(self, labels):
        """``nn.Embedding``'s IndexError for a label outside the table: one ``aminmax`` + a host read -- a blocking sync on the current
        stream, so the image + label encoders make it BEFORE the image trunk is queued (``ImageLabelEncoder.forward``), not behind it."""
        from .beam import check_ids
        check_ids(labels, self.embedding.num_embeddings)

    def forward(self, labels, out=None, checked=False):
        _require_eval(self, self.dropout.p)
        if not checked:
            self.check(labels)
        w = self.embedding.weight.detach()
        if out is None:
            out = torch.empty((labels.shape[0], w.shape[1]), dtype=w.dtype, device=w.device)
        return hip.label_mean(w, labels.contiguous(), out)


class ImageLabelEncoder(nn.Module):
    """Image + label encoder (reference encoders.py:109-144): Linear(cat[image_emb, label_emb])."""

    def __init__(self, num_tokens, emb_dim=256, dropout=0.2):
        super().__init__()
        self.image_encoder = ImageEncoder(emb_dim, dropout)
        self.label_encoder = LabelEncoder(num_tokens, emb_dim, dropout)
        self.linear = nn.Linear(2 * emb_dim, emb_dim)
        self.dropout = nn.Dropout(dropout)

    def _combine(self, image_emb, labels):
        n, e = image_emb.shape
        both = torch.empty((n, 2 * e), dtype=image_emb.dtype, device=image_emb.device)
        both[:, :e].copy_(image_emb)
        self.label_encoder(labels, out=both[:, e:], checked=True)
        return hip.linear(both, self.linear.weight.detach(), self.linear.bias.detach().float())

    @f32x_guarded
    def forward(self, images, labels):
        _require_eval(self, self.dropout.p)
        self.label_encoder.check(labels)                    # host sync in front of the trunk's launches, not behind them
        return self._combine(self.image_encoder(images), labels)


class SpatialImageLabelEncoder(ImageLabelEncoder):
    """BASELINE config 5 composition (no reference class): the image encoder
    keeps its spatial branch, the start embedding is Linear(cat[global_emb, label_emb])."""

    def __init__(self, num_tokens, emb_dim=512, dropout=0.2):
        super().__init__(num_tokens, emb_dim, dropout)
        self.image_encoder.spatial_features = True

    @f32x_guarded
    def forward(self, images, labels):
        _require_eval(self, self.dropout.p)
        self.label_encoder.check(labels)
        emb, spatial = self.image_encoder(images)
        return self._combine(emb, labels), spatial
