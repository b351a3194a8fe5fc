"""nsof -- MI355X-native neuromorphic optical-flow core (host-side Python mirror).

Mirrors the two interfaces the reference's scripts call on the hot path:

* ``calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations,
  poly_n, poly_sigma, flags)`` -- same positional order and keyword names as
  ``cv2.calcOpticalFlowFarneback`` at /root/reference/optical_flow_seg.py:203
  (``farneback_params`` dict at :73-81); ``install()`` assigns it onto ``cv2`` so the
  reference's seg/ob/prediction scripts run unmodified where cv2 exists.
* ``simulate(...)``, ``update_state(w, V)``, ``resistance_exp(w)`` -- the accumulator of
  /root/reference/eventsim/event_mem_sim.py:40-63,164-286.

All arithmetic runs in libnsof.so (HIP, gfx950).  No fallbacks.
"""
from .errors import NsofError, error  # noqa: F401
from .context import Context, default_context  # noqa: F401
from .farneback import (FarnebackParams, StreamPool, calcOpticalFlowFarneback, effective_levels, farneback_batch,  # noqa: F401
                        farneback_many, farneback_pairs, farneback_pairs_dev, farneback_roi_sequence_dev,
                        farneback_sequence, install, level_size,
                        pinned_empty, uninstall)
from .accumulator import (PARAMS, DT, THETA_EVENTS, REFRACTORY_US, Accumulator, bincount_2d,  # noqa: F401
                          generate_synthetic_events, load_events, resistance_exp, simulate, simulate_frames,
                          slice_indices, update_state)

from .gating import (GatingConfig, connectedComponentsWithStats, current_to_gray, dataset_config, frame_to_gray,  # noqa: F401,E402
                     gating_maps, load_gating_stack, opticalFlow3D, process_merged_region, process_separate_regions, update_transition_pic)
from .segment import (MORPH_CROSS, MORPH_ELLIPSE, MORPH_RECT, dilate, erode, getStructuringElement, motion_mask,  # noqa: F401,E402
                      motion_mask_dev, process_flow_region, task_results)
from .predict import (BORDER_CONSTANT, BORDER_REPLICATE, INTER_LINEAR, calculateIntegralError, gray_u8_dev,  # noqa: F401,E402
                      predict_region, predict_region_dev, remap, structural_similarity)
from .frames import compress_image, crop_image, im2double, imresize_lanczos3, process_images  # noqa: F401,E402
from .flowviz import flow_to_image, flow_uv_to_colors, make_colorwheel, viz  # noqa: F401,E402

__all__ = ["calcOpticalFlowFarneback", "install", "uninstall", "FarnebackParams", "farneback_batch", "Context",
           "default_context", "simulate", "update_state", "resistance_exp", "Accumulator", "NsofError", "error"]
