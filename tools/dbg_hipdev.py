import ctypes, os, sys
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "torch":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
from g1_locomotion_amd import _lib
lib = _lib.load()
if mode == "libfirst":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
maps = open("/proc/self/maps").read()
print(sorted(set(l.split()[-1] for l in maps.splitlines() if "amdhip64" in l or "hsa-runtime" in l)))
cfg = _lib.default_config()
h = ctypes.c_void_p()
rc = lib.srbdqp_create(ctypes.byref(cfg), ctypes.byref(h))
print("create rc", rc, lib.srbdqp_last_error(None))
