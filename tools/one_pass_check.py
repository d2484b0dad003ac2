"""On the GPU box: S0 at 4K / 64 spp rendered with the default pass size and as ONE pass (chunk_samples = 2^26): planes must be equal."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge
rt = ge.load_package()
scene = sys.argv[1] if len(sys.argv) > 1 else "S0"
fs = rt.FrontScene(scene, "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
a, sa = gs.render(3840, 2160, 64)
assert rt.rt_lib().rt_set_option(b"chunk_samples", 1 << 26) == 0
gs2 = rt.GpuScene(fs.desc)
b, sb = gs2.render(3840, 2160, 64)
print(scene, "rays", sa.rays(), sb.rays(), "equal planes:", bool(np.array_equal(a, b)), "max", float(np.abs(a - b).max()))
assert sa.rays() == sb.rays() and np.array_equal(a, b)
