"""The plugin surface of the reference: the same registry objects and
``build_*`` helpers as /root/reference/mmdet/models/builder.py:7-59,
mmdet/core/bbox/builder.py:4-21, mmdet/core/bbox/match_costs/builder.py:4-9 and
mmdet/models/utils/builder.py:5-11 (plus the ext-mmcv transformer registries), so that the
reference's config files resolve ``type='DeformableDETR_il'`` etc. to our classes."""
import warnings

from .registry import Registry, build_from_cfg

MODELS = Registry("models")
BACKBONES = NECKS = HEADS = LOSSES = DETECTORS = MODELS   # mmdet aliases one registry (builder.py:9-15)
TRANSFORMER = Registry("Transformer")
LINEAR_LAYERS = Registry("linear layers")
ATTENTION = Registry("attention")
FEEDFORWARD_NETWORK = Registry("feed-forward Network")
TRANSFORMER_LAYER = Registry("transformerLayer")
TRANSFORMER_LAYER_SEQUENCE = Registry("transformer-layers sequence")
POSITIONAL_ENCODING = Registry("position encoding")
BBOX_ASSIGNERS = Registry("bbox_assigner")
BBOX_SAMPLERS = Registry("bbox_sampler")
BBOX_CODERS = Registry("bbox_coder")
MATCH_COST = Registry("Match Cost")


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py:47-59."""
    if train_cfg is not None or test_cfg is not None:
        warnings.warn("train_cfg and test_cfg is deprecated, please specify them in model", UserWarning)
    assert cfg.get("train_cfg") is None or train_cfg is None, \
        "train_cfg specified in both outer field and model field "
    assert cfg.get("test_cfg") is None or test_cfg is None, \
        "test_cfg specified in both outer field and model field "
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


def build_attention(cfg, default_args=None):
    return build_from_cfg(cfg, ATTENTION, default_args)


def build_feedforward_network(cfg, default_args=None):
    return build_from_cfg(cfg, FEEDFORWARD_NETWORK, default_args)


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


def build_positional_encoding(cfg, default_args=None):
    return build_from_cfg(cfg, POSITIONAL_ENCODING, default_args)


def build_assigner(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_ASSIGNERS, default_args)


def build_sampler(cfg, **default_args):
    return build_from_cfg(cfg, BBOX_SAMPLERS, default_args)


def build_match_cost(cfg, default_args=None):
    return build_from_cfg(cfg, MATCH_COST, default_args)
