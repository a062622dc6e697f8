"""Kernel times of the rasterizer instances the OPERATOR API reaches (GaussianRasterizer, the reference's own call shape,
diff_cur_rasterization/__init__.py:46-151), per BASELINE config, measured with the library's own HIP events (cgs_prof_*):

    python profiles/probes/general_instances.py [cfg3] [n_rep]

Cases (upstream gradients / what requires grad):
    reference_call    colours == 1 without grad, only dL/dcolour flowing in (gaussian_renderer/__init__.py:96-129) -> the gated
                      pair-major unit kernel
    training_general  the same call with the unit route switched off (OPT_GENERAL_BACKWARD in the settings) -> k_render_bwd3<0,0,0>
    colour_grad       arbitrary colours that require grad, dL/dcolour only                            -> k_render_bwd3<0,0,1>
    colour_allmap     ... + dL/dall_map                                                               -> k_render_bwd3<1,0,1>
    all_grad          ... + dL/dinvdepth + dL/dall_map                                                -> k_render_bwd3<1,1,1>
Prints one JSON object {case: {kernel: us}} (bench.time_instances; bench.py's `general_route` block reports the same)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402  (the measurement lives in bench.py: its `general_route` block reports the same numbers)

CASES, time_instances = bench.CASES, bench.time_instances


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    print(json.dumps(time_instances(cfg, n_rep)))
