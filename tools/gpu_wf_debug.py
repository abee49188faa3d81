import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_raytracer_amd import api

def run(args, envs):
    hs = api.HostScene(args)
    ds = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy(); p.pipeline = api.RT_PIPELINE_MEGAKERNEL; p.collect_stats = 1
    mega = ds.render(hs.camera, p); sm = ds.stats()
    for env in envs:
        for k in ("RT_WF_POOL", "RT_WF_REFILL", "RT_WF_SPLIT", "RT_WF_CHECK", "RT_WF_INNER_MIN"):
            os.environ.pop(k, None)
        os.environ.update(env)
        p.pipeline = api.RT_PIPELINE_WAVEFRONT
        outs = []
        for rep in range(2):
            wf = ds.render(hs.camera, p); st = ds.stats()
            bad = int((~((wf == mega) | (np.isnan(wf) & np.isnan(mega)))).any(axis=2).sum())
            outs.append((bad, st.rays))
        print(args[0], env, "bad px/rays per run", outs, "mega rays", sm.rays, flush=True)

envs = [{}, {"RT_WF_REFILL": "64"}, {"RT_WF_POOL": "4096"}, {"RT_WF_POOL": "4096", "RT_WF_REFILL": "64"}, {"RT_WF_CHECK": "1"}, {"RT_WF_SPLIT": "0"}]
run(["scenes/cornell", "-w=64", "-s=16", "--seed=1"], envs)
run(["-w=96", "-s=16", "--seed=7"], envs)
run(["scenes/light_test", "-w=96", "-s=32", "-t=2", "--seed=3"], envs)
