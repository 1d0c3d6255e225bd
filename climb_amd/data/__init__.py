from .image_pipeline import DeviceImagePipeline, vilt_output_size, resample_coefficients
from .prefetch import PrefetchLoader

__all__ = ["DeviceImagePipeline", "PrefetchLoader", "vilt_output_size", "resample_coefficients"]
