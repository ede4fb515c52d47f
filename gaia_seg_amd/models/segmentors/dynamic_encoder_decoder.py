"""DynamicEncoderDecoder — the search-space root.

Host-side mirror of gaiaseg/models/segmentors/dynamic_encoder_decoder.py:8-42: routes
``arch['backbone']`` to the backbone; decode_head / neck / auxiliary_head manipulation are no-ops
exactly as in the reference (:35-42)."""
from ...core.dynamic import DynamicMixin
from ..builder import SEGMENTORS
from .encoder_decoder import EncoderDecoder


@SEGMENTORS.register_module()
class DynamicEncoderDecoder(EncoderDecoder, DynamicMixin):
    search_space = {"backbone", "decode_head", "neck", "auxiliary_head"}

    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None):
        super().__init__(backbone=backbone, decode_head=decode_head, neck=neck,
                         auxiliary_head=auxiliary_head, train_cfg=train_cfg, test_cfg=test_cfg,
                         pretrained=pretrained)

    def manipulate_backbone(self, arch_meta):
        self.backbone.manipulate_arch(arch_meta)

    def manipulate_decode_head(self, arch_meta):
        pass

    def manipulate_neck(self, arch_meta):
        pass

    def manipulate_auxiliary_head(self, arch_meta):
        pass

    def active_parameters(self):
        """Parameters that take part in the current subnet: everything except the blocks skipped
        by the depth state (they are DDP 'unused parameters' in the reference,
        gaiaseg/apis/train.py:88-96, and receive no optimizer update: SURVEY.md Appendix A13)."""
        seen, out = set(), []
        mods = list(self.backbone.active_modules())
        for name, m in self.named_children():
            if name != "backbone":
                mods.append(m)
        for m in mods:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out
