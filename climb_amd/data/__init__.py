from .image_pipeline import DeviceImagePipeline, vilt_output_size, resample_coefficients
from .prefetch import PrefetchLoader
from .sharding import ShardedBatchSampler, ShardedDataLoader, dp_rank_world, replicated, shard_of

__all__ = ["DeviceImagePipeline", "PrefetchLoader", "vilt_output_size", "resample_coefficients", "ShardedBatchSampler", "ShardedDataLoader",
           "dp_rank_world", "replicated", "shard_of"]
