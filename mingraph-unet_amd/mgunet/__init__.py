"""mgunet: MI355X-native MinGraph-UNet segmentation hot path (host-side mirror of the reference's
model/ API over libmgunet.so).  Importing works without a GPU; running anything needs one."""
from .config import build_from_config, get_config_recursively, load_config  # noqa: F401
from .detection import DetectionHead  # noqa: F401
from .engine import E2ETrainer, FlatAdam, MinGraphUNet, MinGraphUNetE2E, StepLR, Trainer, adam_state_dict, allreduce_mean_, argmax_classes, gat_forward_csr, segment_batch, shard_batch  # noqa: F401
from .gat import GATNetwork, GraphAttentionLayer, MultiHeadGATLayer, seed_dropout  # noqa: F401
from .losses import EllipticalShapeLoss, FeatureConsistencyLoss, TVLoss, dice_loss  # noqa: F401
from .preprocess import EdgeDetector, HistogramEqualizer, ImagePreprocessor, patch_features_u8, postprocess_segmentation  # noqa: F401
from .mincut import MinCutRefinement, PatchSegmentPredictor  # noqa: F401
from .patch_graph import PatchGraphConstructor  # noqa: F401
from .region import FeatureFusion, region_edge_index, region_fuse, region_mean_pool, region_stage  # noqa: F401
from .unet import ConvBlock, DecoderBlock, UNet, UNetDecoder, UNetEncoder  # noqa: F401
from ._lib import build, lib  # noqa: F401

__all__ = ["TVLoss", "dice_loss", "FeatureConsistencyLoss", "EllipticalShapeLoss", "ImagePreprocessor", "EdgeDetector", "HistogramEqualizer",
           "patch_features_u8", "postprocess_segmentation", "DetectionHead", "FeatureFusion", "region_stage", "region_mean_pool", "region_fuse", "region_edge_index", "MinCutRefinement", "PatchSegmentPredictor", "UNet", "UNetEncoder", "UNetDecoder", "ConvBlock", "DecoderBlock", "GATNetwork", "MultiHeadGATLayer",
           "GraphAttentionLayer", "PatchGraphConstructor", "MinGraphUNet", "MinGraphUNetE2E", "segment_batch", "argmax_classes",
           "gat_forward_csr", "shard_batch", "Trainer", "E2ETrainer", "FlatAdam", "StepLR", "adam_state_dict", "allreduce_mean_", "load_config", "get_config_recursively", "build_from_config", "build", "lib"]
