from .image_pipeline import DeviceImagePipeline, vilt_output_size, resample_coefficients

__all__ = ["DeviceImagePipeline", "vilt_output_size", "resample_coefficients"]
