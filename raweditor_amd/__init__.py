"""raweditor_amd -- MI355X-native develop path for RawEditor (Bayer demosaic + 10-slider colour
stack + histogram as fused HIP kernels behind the librawdev C ABI, include/rawdev.h).

Host mirror of the reference interface for this path:
    EditParams      <- state::edit::EditParams   (src/state/edit.rs)
    RenderPipeline  <- gpu::RenderPipeline       (src/gpu/pipeline.rs)
    BatchExporter   <- (new) frame-sharded batch export + global histogram (one process per GPU)
    NodeBatch       <- (new) the same over the GPUs of one node from one process (RCCL all-reduce inside librawdev)
"""
from ._lib import (FMT_RGBA_F16, FMT_RGBA_F32, FMT_RGBA_U8, FMT_RGB_U8, MATH_STRICT, MATH_CONTRACTED, MATRIX_REFERENCE,
                   MATRIX_ROW_MAJOR, BYTES_PER_PIXEL,
                   RawdevError, device_count)
from .edit import EditParams, FIELDS, UI_RANGES
from .pipeline import (RenderPipeline, PinnedBytes, measure_hbm, measure_valu, calculate_cam_to_srgb_matrix, is_identity_matrix, derived_dims, elided_steps,
                       IDENTITY_MATRIX)
from .batch import BatchExporter, NodeBatch, shard_frames
from .export import Exporter

__all__ = ["EditParams", "RenderPipeline", "PinnedBytes", "measure_hbm", "measure_valu", "BatchExporter", "NodeBatch", "Exporter", "RawdevError", "shard_frames",
           "FMT_RGBA_F32", "FMT_RGBA_F16", "FMT_RGBA_U8", "FMT_RGB_U8", "MATH_STRICT", "MATH_CONTRACTED", "MATRIX_REFERENCE", "MATRIX_ROW_MAJOR", "BYTES_PER_PIXEL", "FIELDS", "UI_RANGES",
           "calculate_cam_to_srgb_matrix", "is_identity_matrix", "derived_dims", "elided_steps", "device_count", "IDENTITY_MATRIX"]
