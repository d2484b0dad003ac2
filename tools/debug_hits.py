import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
from conftest import SCENE_CSG, SCENE_BUMPY, SCENE_GLASS
rt = ge.load_package(); ob = ge.load_oracle()
scene = {"csg": SCENE_CSG, "bumpy": SCENE_BUMPY, "glass": SCENE_GLASS}.get(sys.argv[1], sys.argv[1])
opt = rt.FrontScene(scene, "test", True); raw = rt.FrontScene(scene, "test", False)
g = rt.GpuScene(opt.desc); o = ob.OracleScene(raw.desc)
rng = np.random.default_rng(7)
n = 30000
org = np.tile(np.array([0.0, 0.0, 9.0]), (n, 1)) + rng.normal(0, 0.02, (n, 3))
dr = np.concatenate([rng.uniform(-0.7, 0.7, (n, 2)), -np.ones((n, 1))], axis=1)
G = g.intersect(org, dr, max_hits=12); O = o.intersect(org, dr, max_hits=12)
same = (G[0] == O[0]) & (G[3] == O[3]).all(axis=1) & (G[4] == O[4]).all(axis=1) & (G[5] == O[5]).all(axis=1)
bad = np.where(~same)[0]
print("mismatch", len(bad), "of", n)
ids = {}
d = raw.desc.contents
for i in range(d.n_nodes):
    ids[d.nodes[i].id] = d.nodes[i].type
for k in bad[:12]:
    print("ray", k, "gpu cnt", G[0][k], "orc cnt", O[0][k])
    print("  gpu t   ", np.round(G[1][k][:max(G[0][k],O[0][k])], 9), "leaf", G[3][k][:G[0][k]], "mat", G[4][k][:G[0][k]])
    print("  orc t   ", np.round(O[1][k][:max(G[0][k],O[0][k])], 9), "leaf", O[3][k][:O[0][k]], "mat", O[4][k][:O[0][k]], "types", [ids.get(x) for x in O[3][k][:O[0][k]]])
