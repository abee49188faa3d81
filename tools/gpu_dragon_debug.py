import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rust_raytracer_amd import api
from oracle import pyoracle
np.set_printoptions(linewidth=250, precision=9, suppress=False)
import subprocess
if not os.path.exists("scenes/resource/dragon_high.obj"):
    subprocess.run(["./tools/gen_dragon", "scenes/resource/dragon_high.obj"], check=True)
hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=4", "--seed=12"])
scene = api.DeviceScene(hs.desc, 0)
q = hs.params.copy(); q.band_rows, q.n_parts, q.part = 1, 97, 5
ref, _ = pyoracle.render(hs.desc, hs.camera, q)
gpu = scene.render(hs.camera, q)
rows = api.owned_rows(hs.height, q)
a, b = gpu[..., :3], ref[..., :3]
rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
bad = np.argwhere((rel > 1e-9).any(axis=2))
print("bad pixels", len(bad), "of", a.shape[0] * a.shape[1])
dump = []
for (r, x) in bad[:12]:
    y = rows[r]
    for sy in range(2):
        for sx in range(2):
            orgb, otr = pyoracle.trace_sample(hs.desc, hs.camera, hs.params, 0, int(x), int(y), sx, sy)
            grgb, gtr = scene.trace_sample(hs.camera, hs.params, 0, int(x), int(y), sx, sy)
            if np.allclose(orgb, grgb, rtol=1e-9, equal_nan=True): continue
            n = min(len(otr), len(gtr))
            k = next((i for i in range(n) if not np.isclose(otr[i, 0], gtr[i, 0], rtol=1e-9)), None)
            print(f"px ({x},{y}) s({sx},{sy}) first differing bounce {k} of {len(otr)}/{len(gtr)}")
            if k is not None:
                print("   oracle t,pos,mat", otr[k, :5], "o", otr[k, 11:14], "d", otr[k, 14:17])
                print("   gpu    t,pos,mat,op,tri", gtr[k, :7], "o", gtr[k, 11:14], "d", gtr[k, 14:17])
                dump.append({"o": list(otr[k, 11:14]), "d": list(otr[k, 14:17]), "t_oracle": otr[k, 0], "t_gpu": gtr[k, 0], "tri_gpu": gtr[k, 6],
                             "o_gpu": list(gtr[k, 11:14]), "d_gpu": list(gtr[k, 14:17])})
json.dump(dump, open("gpurun_out/dragon_mismatch.json", "w"))
