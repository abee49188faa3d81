import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_raytracer_amd import api

def run(args, envs):
    hs = api.HostScene(args)
    ds = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy(); p.pipeline = api.RT_PIPELINE_MEGAKERNEL; p.collect_stats = int(os.environ.get('DBG_STATS', '0'))
    mega = ds.render(hs.camera, p); sm = ds.stats()
    for env in envs:
        for k in ("RT_WF_POOL", "RT_WF_REFILL", "RT_WF_SPLIT", "RT_WF_CHECK", "RT_WF_INNER_MIN", "RT_LDS_TABLES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        p.pipeline = api.RT_PIPELINE_WAVEFRONT
        outs = []
        for rep in range(2):
            wf = ds.render(hs.camera, p); st = ds.stats()
            bad = int((~((wf == mega) | (np.isnan(wf) & np.isnan(mega)))).any(axis=2).sum())
            outs.append((bad, st.rays))
        print(args[0], env, "bad px/rays per run", outs, "mega rays", sm.rays, flush=True)

for k in ("RT_WF_POOL", "RT_WF_REFILL", "RT_WF_SPLIT", "RT_WF_CHECK", "RT_WF_INNER_MIN", "RT_LDS_TABLES"): os.environ.pop(k, None)
for d in (1, 2, 3, 4, 20):
    run(["scenes/cornell", "-w=64", "-s=16", "--seed=1", f"--max-depth={d}"], [{"RT_LDS_TABLES": "0"}])
# which pixels / how different at depth 2
hs = api.HostScene(["scenes/cornell", "-w=64", "-s=16", "--seed=1", "--max-depth=2"])
ds = api.DeviceScene(hs.desc, 0)
p = hs.params.copy(); p.pipeline = api.RT_PIPELINE_MEGAKERNEL
mega = ds.render(hs.camera, p)
os.environ["RT_LDS_TABLES"] = "0"; p.pipeline = api.RT_PIPELINE_WAVEFRONT
wf = ds.render(hs.camera, p)
bad = np.argwhere((wf != mega).any(axis=2))
print("bad", len(bad), "bbox", bad.min(axis=0), bad.max(axis=0))
for (y, x) in bad[:5]: print((y, x), "wf", wf[y, x, :3], "mega", mega[y, x, :3])
