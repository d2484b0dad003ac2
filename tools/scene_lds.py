"""Print what a scene's upload decides about LDS (RT_AMD_DEBUG line): tools/scene_lds.py S4"""
import os, sys
os.environ["RT_AMD_DEBUG"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
fs = rt.FrontScene(sys.argv[1] if len(sys.argv) > 1 else "S4", "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
