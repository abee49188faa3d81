import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_raytracer_amd import api
from oracle import pyoracle
args = ["scenes/cornell", "-w=5", "-s=1"] + sys.argv[1:]
hs = api.HostScene(args)
ref, ost = pyoracle.render(hs.desc, hs.camera, hs.params)
np.set_printoptions(precision=17, linewidth=250)
sc = api.DeviceScene(hs.desc, 0)
p = hs.params.copy(); p.pipeline = api.RT_PIPELINE_MEGAKERNEL
mega = sc.render(hs.camera, p)
os.environ["RT_WF_TRACE"] = "32"
p.pipeline = api.RT_PIPELINE_WAVEFRONT
sys.stderr.flush()
wf = sc.render(hs.camera, p)
print("mega == oracle:", np.array_equal(mega, ref, equal_nan=True))
bad = np.argwhere(~((wf == ref) | (np.isnan(wf) & np.isnan(ref))).all(axis=2))
print("wavefront bad pixels (y, x):", bad.tolist())
for y, x in bad:
    print("pixel", y, x, "sample index", y * hs.width + x, "wf", wf[y, x, :3], "ref", ref[y, x, :3])
    orgb, otr = pyoracle.trace_sample(hs.desc, hs.camera, hs.params, 0, int(x), int(y), 0, 0)
    print(" oracle per bounce: t, pos, material, kind, pdf, s_pdf | ray o, d")
    for r in otr:
        print("  ", r[0], r[1:4], int(r[4]), int(r[5]), r[6], r[7], "|", r[11:14], r[14:17])
