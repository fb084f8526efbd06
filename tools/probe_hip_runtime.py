import sys, importlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "libhsa" in l})
if order == "lib_first":
    e3d = importlib.import_module("dataset-pipeline_amd"); print("init", e3d.lib().e3d_init(0)); print(maps())
    import torch; print("torch avail", torch.cuda.is_available()); print(maps())
elif order == "import_torch_then_lib":
    import torch
    print(maps())
    e3d = importlib.import_module("dataset-pipeline_amd"); print("init", e3d.lib().e3d_init(0)); print(maps())
    print("torch avail", torch.cuda.is_available())
else:
    import torch; print("torch avail", torch.cuda.is_available(), torch.zeros(1, device="cuda")); print(maps())
    e3d = importlib.import_module("dataset-pipeline_amd"); print("init", e3d.lib().e3d_init(0)); print(maps())
