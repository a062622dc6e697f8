"""Emits tests/golden/train_py_surface.json: every attribute / method name the reference's training driver touches on its
model object (``gaussians.<name>`` in /root/reference/train.py), with the call-site lines -- the surface a drop-in model class
has to offer.  Data only (names and line numbers); run HERE (needs /root/reference), the fixture travels.
usage: python tests/golden/make_train_surface.py"""
import json
import os
import re

REF = "/root/reference/train.py"
names = {}
for no, line in enumerate(open(REF), 1):
    for m in re.finditer(r"\bgaussians\.([A-Za-z_][A-Za-z_0-9]*)", line):
        names.setdefault(m.group(1), []).append(no)
# SURVEY.md section 2 marks these OUT OF SCOPE (CPU/numpy post-processing and visualisation of the trained curves)
# (merge_curves / fit_curve_to_line, SURVEY 2a row 6, were on this list until round 6: scene/topology.py has them now)
out_of_scope = {"draw_curve": "matplotlib visualisation (SURVEY 2a row 14)",
                "draw_ellipsoids": "matplotlib visualisation (SURVEY 2a row 14)"}
out = {"source": "train.py of zhirui-gao/Curve-Gaussian (reference snapshot 2025-09-05)",
       "names": {k: names[k] for k in sorted(names)}, "out_of_scope": out_of_scope}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_py_surface.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, len(names), "names")
